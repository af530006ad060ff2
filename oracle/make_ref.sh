#!/bin/bash
# TEST / MEASUREMENT INFRASTRUCTURE.  Puts the reference's hot-path files (the modules oracle/ref_stub.py imports,
# SURVEY.md 8(c), plus configs/yunet_{n,s}.py) UNMODIFIED under oracle/_ref/, which is git-ignored -- no reference
# source ever enters the history -- but travels to the GPU box with gpurun like the built .so files do, so that
# bench.py's cpu_baseline times THE REFERENCE ITSELF (its own Python files on torch CPU, under the arithmetic-free mmcv
# stub of oracle/ref_stub.py) on the MI355X host's cores: cpu_baseline.kind = "reference" (VERDICT r3
# next 5).  Runs where /root/reference exists (the build container; __graft_entry__.build() calls it).
#   oracle/make_ref.sh [REFERENCE_ROOT]
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=${1:-${YUNET_REFERENCE_ROOT:-/root/reference}}
DST=$HERE/_ref
[ -d "$SRC/mmdet" ] || { echo "no reference tree at $SRC"; exit 0; }
# build into a scratch directory and swap it in only when complete: a failure half-way leaves the previous copy alone
TMP=$HERE/_ref.tmp.$$
rm -rf "$TMP"
mkdir -p "$TMP/configs"
trap 'rm -rf "$TMP"' EXIT
python3 - "$HERE" "$SRC" "$TMP" <<'PY'
import os, shutil, sys
here, src, dst = sys.argv[1:4]
sys.path.insert(0, here)
os.environ['YUNET_REFERENCE_ROOT'] = src
import ref_stub
n = 0
for leaf in ref_stub._LEAVES:
    rel = os.path.join(*leaf.split('.')) + '.py'
    os.makedirs(os.path.dirname(os.path.join(dst, rel)), exist_ok=True)
    shutil.copyfile(os.path.join(src, rel), os.path.join(dst, rel))
    n += 1
for cfg in ('yunet_n.py', 'yunet_s.py'):
    shutil.copyfile(os.path.join(src, 'configs', cfg), os.path.join(dst, 'configs', cfg))
print(f'oracle/_ref: {n} reference modules + 2 configs from {src}')
PY
rm -rf "$DST"
mv "$TMP" "$DST"
trap - EXIT
