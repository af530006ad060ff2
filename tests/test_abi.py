"""CPU: libyunet_hip.so loads without a GPU and exports every symbol include/yunet_hip.h declares;
the ctypes mirrors of the C structs have the C layout.  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'yunet_hip.h')


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|size_t)\s+(yunet_\w+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    import yunet_amd._lib as L
    lib = L.load()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} is declared in include/yunet_hip.h but not exported'
    assert sorted(L.EXPORTED) == names, 'ctypes signature table and header disagree'
    assert lib.yunet_abi_version() == 11
    assert lib.yunet_conv_blocks() >= 256
    assert lib.yunet_loss_blocks(256, 2100) >= 1


def test_missing_library_fails_loudly(monkeypatch):
    import yunet_amd._lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', '/nonexistent/libyunet_hip.so')
    with pytest.raises(L.YunetHipError, match='no CPU fallback'):
        L.load()


def test_ctypes_structs_match_c_layout(tmp_path):
    import yunet_amd._lib as L
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "yunet_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(YunetOp),sizeof(YunetDP),sizeof(YunetBN),sizeof(YunetLevels),'
                   'sizeof(YunetLossCfg),offsetof(YunetOp,p),offsetof(YunetOp,bn),'
                   'offsetof(YunetOp,dp),offsetof(YunetOp,lv),offsetof(YunetDP,prof),sizeof(YunetComm),'
                   'offsetof(YunetComm,status));return 0;}')
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(L.YunetOp), C.sizeof(L.YunetDP), C.sizeof(L.YunetBN), C.sizeof(L.YunetLevels),
            C.sizeof(L.YunetLossCfg), L.YunetOp.p.offset, L.YunetOp.bn.offset, L.YunetOp.dp.offset,
            L.YunetOp.lv.offset, L.YunetDP.prof.offset, C.sizeof(L.YunetComm), L.YunetComm.status.offset]
    assert got == want


def test_op_list_constants_match_the_header():
    """The op-list conventions the Python plan builder relies on (lane / group slots of YunetOp.i, group size, lane count)
    are the header's."""
    import yunet_amd._lib as L
    txt = open(HEADER).read()
    defs = {k: int(v) for k, v in re.findall(r'#define\s+(YUNET_\w+)\s+(\d+)\s', txt)}
    assert defs['YUNET_OP_LANE'] == L.OP_LANE and defs['YUNET_MAX_LANES'] == L.MAX_LANES
    assert defs['YUNET_OP_GROUP'] == L.OP_GROUP and defs['YUNET_DP_GROUP_MAX'] == L.DP_GROUP_MAX
    assert L.OP_GROUP != L.OP_LANE and L.OP_GROUP < 11        # i[11] carries the activation storage type
    assert defs['YUNET_MAX_RANKS'] == L.MAX_RANKS and defs['YUNET_IPC_HANDLE_BYTES'] == L.IPC_HANDLE_BYTES
    assert defs['YUNET_COMM_HEADER_BYTES'] == L.COMM_HEADER_BYTES        # one-shot inbox: flags + counter before the slots
    lib = L.load()
    assert lib.yunet_comm_inbox_bytes(4, 1000) == L.COMM_HEADER_BYTES + 2 * 4 * 1024
