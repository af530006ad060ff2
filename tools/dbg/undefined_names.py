"""Names a function reads that no enclosing scope, the module or builtins define (the class of bug a test that
never runs the line cannot see): python tools/dbg/undefined_names.py FILE..."""
import builtins
import symtable
import sys


def check(path):
    src = open(path).read()
    top = symtable.symtable(src, path, 'exec')
    module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    star = 'import *' in src
    bad = []

    def walk(t):
        for s in t.get_symbols():
            n = s.get_name()
            if s.is_referenced() and s.is_global() and not s.is_assigned() and n not in module_names \
                    and not hasattr(builtins, n) and n not in ('__file__', '__name__', '__doc__'):
                bad.append((t.get_name(), t.get_lineno(), n))
        for c in t.get_children():
            walk(c)
    walk(top)
    return [] if star else bad


if __name__ == '__main__':
    rc = 0
    for f in sys.argv[1:]:
        for scope, line, name in check(f):
            print(f'{f}:{line}: {scope}() reads undefined name {name!r}')
            rc = 1
    sys.exit(rc)
