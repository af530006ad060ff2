"""not-gpu: weight export in libfacedetection's C++ data format (yunet_amd.export.to_cpp)
against the text produced by the unmodified reference tool (tools/yunet2cpp.py CppConvertor):
committed SHA-256 for deterministic states, and a live comparison when the reference tree exists."""
import hashlib
import json
import os

import pytest
import torch

import detect_oracle as D

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'export_cpp.json')


def build(kind):
    import yunet_amd
    from yunet_amd.export import to_cpp
    arch, sd = D.make_state(kind, 3, 160, calib_iters=5)
    cfg = yunet_amd.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                 'configs', f'yunet_{kind}.py'))
    model = yunet_amd.build_detector(cfg.model)
    model.load_state_dict(sd, strict=True)
    return to_cpp(model), sd


@pytest.mark.parametrize('kind', ['n', 's'])
def test_cpp_export_matches_reference_tool(kind):
    g = json.load(open(GOLD))[kind]
    text, _ = build(kind)
    assert len(text) == g['length']
    assert text[:200] == g['head'] and text[-300:] == g['tail']
    assert hashlib.sha256(text.encode()).hexdigest() == g['sha256']
    # structure: stem row padded to 32, depthwise transposed, one ConvInfoStruct row per array pair
    n_arrays = text.count('_weight[')
    assert f'ConvInfoStruct param_pConvInfo[{n_arrays}]' in text
    assert 'backbone__model0_pw_weight[16*32*1*1]' in text


def test_cpp_export_live_reference():
    import ref_stub
    if not ref_stub.available():
        pytest.skip('reference tree not present')
    import make_golden_export as M
    tool = M.reference_tool()
    text, sd = build('s')
    model, _ = ref_stub.build_detector('yunet_s.py')
    model.load_state_dict(sd, strict=True)
    assert tool.CppConvertor(model).data == text


def test_number_format():
    from yunet_amd.export import _num
    assert _num(1.0, '.3g') == '1.f' and _num(0.5, '.3g') == '0.5f'
    assert _num(1.23456e-7, '.3g') == '1.23e-07f' and _num(-12.0, '.3g') == '-12.f'


def test_cpp_export_tower_head_live_reference():
    """YuNet_Head's own defaults (per-level cls / reg towers, yunet_head.py:52-53, 115-147): the C++ array dump walks
    the module tree generically, so the tower units appear where the reference tool puts them -- byte-identical to the
    unmodified CppConvertor on the same weights."""
    import ref_stub
    if not ref_stub.available():
        pytest.skip('reference tree not present')
    import make_golden_export as M
    import yunet_amd
    import yunet_oracle as O
    from yunet_amd.export import to_cpp
    tool = M.reference_tool()
    head = dict(stacked_convs=2, shared_stacked_convs=2)
    ref_model, _ = ref_stub.build_detector('yunet_n.py', head=head)
    cfg = yunet_amd.Config.fromfile(os.path.join(ref_stub.REF_ROOT, 'configs', 'yunet_n.py'))
    cfg.model.bbox_head.update(head)
    mine = yunet_amd.build_detector(cfg.model)
    sd = O.init_state(mine.arch(), seed=5)
    g = torch.Generator().manual_seed(5)
    for k in sd:
        if k.endswith('running_mean'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.3
        if k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    mine.load_state_dict(sd, strict=True)
    ref_model.load_state_dict(sd, strict=True)
    mine.eval()
    text = to_cpp(mine)
    assert tool.CppConvertor(ref_model).data == text
    assert text.count('multi_level_cls_convs') > 0 and text.count('multi_level_reg_convs') > 0
