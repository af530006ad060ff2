#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- tests/golden/export_cpp.json: SHA-256 (and the first lines) of the
facedetectcnn-data.cpp text produced by the UNMODIFIED reference tool (tools/yunet2cpp.py,
class CppConvertor) for deterministic detector states (detect_oracle.make_state)."""
import hashlib
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detect_oracle as D   # noqa: E402
import ref_stub             # noqa: E402


def reference_tool():
    ref_stub.load_reference()
    exp = types.ModuleType('mmdet.core.export')
    exp.build_model_from_cfg = None          # only used by the tool's __main__
    sys.modules['mmdet.core.export'] = exp
    spec = importlib.util.spec_from_file_location(
        'ref_yunet2cpp', os.path.join(ref_stub.REF_ROOT, 'tools', 'yunet2cpp.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    tool = reference_tool()
    out = {}
    for kind in ('n', 's'):
        arch, sd = D.make_state(kind, 3, 160, calib_iters=5)
        model, _ = ref_stub.build_detector(f'yunet_{kind}.py')
        model.load_state_dict(sd, strict=True)
        text = tool.CppConvertor(model).data
        out[kind] = dict(sha256=hashlib.sha256(text.encode()).hexdigest(), length=len(text),
                         head=text[:200], tail=text[-300:])
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'export_cpp.json')
    json.dump(out, open(path, 'w'), indent=1)
    print({k: v['sha256'][:16] for k, v in out.items()})


if __name__ == '__main__':
    main()
