"""CPU: the ONNX exporter (libfacedetection.train_amd/onnx_export.py) -- structure against the
reference's shipped onnx/yunet_n_320_320.onnx (same node sequence, operator attributes, input and
12 output names / shapes, opset, IR version) and numerics against the oracle's eval-mode forward.
The reference's own file is executed by the same mini runtime with weights/yunet_n.pth as a check of
the runtime itself."""
import os

import pytest
import torch

import onnx_mini as OM
import yunet_oracle as O

REF_ONNX = '/root/reference/onnx/yunet_n_320_320.onnx'
REF_CKPT = '/root/reference/weights/yunet_n.pth'


def _expected(sd, arch, x):
    cls, box, obj, kps = O.conv_stack_forward(x, sd, arch, training=False)
    want = {}
    for name, maps, act in (('cls', cls, True), ('obj', obj, True), ('bbox', box, False), ('kps', kps, False)):
        for m, s in zip(maps, arch['strides']):
            t = m.permute(0, 2, 3, 1).reshape(m.shape[0], -1, m.shape[1])
            want[f'{name}_{s}'] = torch.sigmoid(t) if act else t
    return want


def _state(kind, seed, **arch_kw):
    arch = O.yunet_arch(kind, **arch_kw)
    sd = O.init_state(arch, seed)
    g = torch.Generator().manual_seed(seed)
    for k in sd:                                     # non-trivial running statistics / affine terms
        if k.endswith('running_mean') or k.endswith('.bn.bias') or k.endswith('bn1.bias'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.3
        if k.endswith('running_var') or k.endswith('.bn.weight') or k.endswith('bn1.weight'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    return arch, sd


@pytest.mark.parametrize('kind,hw,dynamic', [('n', (320, 320), False), ('s', (160, 224), False), ('n', (320, 320), True)])
def test_export_numerics_vs_oracle(kind, hw, dynamic, tmp_path):
    from yunet_amd.onnx_export import export_onnx
    arch, sd = _state(kind, 3)
    path = str(tmp_path / 'm.onnx')
    export_onnx(sd, arch, path, input_shape=hw, dynamic=dynamic)
    m = OM.load(path)
    shapes = [hw, (hw[0] + 64, hw[1] - 32)] if dynamic else [hw]
    for h, w in shapes:
        x = torch.rand(2 if dynamic else 1, 3, h, w, generator=torch.Generator().manual_seed(1)) * 255
        got, want = OM.run(m, x), _expected(sd, arch, x)
        assert list(got) == [f'{n}_{s}' for n in ('cls', 'obj', 'bbox') for s in (8, 16, 32)] + \
            [f'kps_{s}' for s in (8, 16, 32)]                         # tools/yunet2onnx.py:85-90
        for k in want:
            assert got[k].shape == want[k].shape, k
            assert float((got[k] - want[k]).abs().max()) <= 1e-4 * max(1.0, float(want[k].abs().max())), k


def test_export_tower_head_numerics_vs_oracle(tmp_path):
    """YuNet_Head's own defaults (shared_stacked_convs = 2, stacked_convs = 2: per-level cls / reg towers,
    yunet_head.py:115-147, 191-207): the exported graph takes cls from the cls tower and bbox / obj / kps from the reg
    tower -- numerics against the oracle's eval forward (pinned to the reference for this head in
    tests/test_oracle_vs_reference.py), and the node count grows by the 12 tower units + 3 more share units."""
    from yunet_amd.onnx_export import export_onnx
    arch, sd = _state('n', 5, stacked_convs=2, shared_stacked_convs=2)
    path = str(tmp_path / 't.onnx')
    export_onnx(sd, arch, path, input_shape=(160, 160))
    m = OM.load(path)
    x = torch.rand(1, 3, 160, 160, generator=torch.Generator().manual_seed(2)) * 255
    got, want = OM.run(m, x), _expected(sd, arch, x)
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert float((got[k] - want[k]).abs().max()) <= 1e-4 * max(1.0, float(want[k].abs().max())), k
    # a unit is Conv (1x1) + Conv (3x3 depthwise, BN folded) + Relu: 15 more units than the shipped YuNet_n graph
    arch0, sd0 = _state('n', 5)
    export_onnx(sd0, arch0, str(tmp_path / 'n.onnx'), input_shape=(160, 160))
    n_nodes = lambda mm: len(mm.nodes) if hasattr(mm, 'nodes') else len(mm['nodes'])          # noqa: E731
    assert n_nodes(m) == n_nodes(OM.load(str(tmp_path / 'n.onnx'))) + 15 * 3


@pytest.mark.skipif(not os.path.exists(REF_ONNX), reason='reference tree not present')
def test_structure_matches_reference_file(tmp_path):
    from yunet_amd.onnx_export import export_onnx
    arch, sd = _state('n', 0)
    ours = OM.structure(OM.load(export_onnx(sd, arch, None, (320, 320))))
    ref = OM.structure(OM.load(REF_ONNX))
    assert ours['opset'] == ref['opset'] == 11 and ours['ir_version'] == ref['ir_version']
    assert ours['inputs'] == ref['inputs'] and ours['outputs'] == ref['outputs']
    assert len(ours['ops']) == len(ref['ops']) == 115
    for i, (a, b) in enumerate(zip(ours['ops'], ref['ops'])):
        assert a == b, (i, a, b)
    # same initialiser shapes (BatchNorm folded: weight + bias per conv, plus the few constants)
    mo, mr = OM.load(export_onnx(sd, arch, None, (320, 320))), OM.load(REF_ONNX)
    so = sorted(tuple(v.shape) for v in mo['inits'].values() if v.ndim == 4)
    sr = sorted(tuple(v.shape) for v in mr['inits'].values() if v.ndim == 4)
    assert so == sr and len(so) == 59


@pytest.mark.skipif(not (os.path.exists(REF_ONNX) and os.path.exists(REF_CKPT)), reason='reference tree not present')
def test_reference_file_runs_in_the_mini_runtime_and_matches_its_checkpoint():
    """Validates the test runtime: the shipped ONNX file, executed here, reproduces the oracle's
    forward with the shipped trained weights."""
    ck = torch.load(REF_CKPT, map_location='cpu', weights_only=False)
    sd = {k: v.float() if v.is_floating_point() else v for k, v in ck['state_dict'].items()}
    arch = O.yunet_arch('n')
    x = torch.rand(1, 3, 320, 320, generator=torch.Generator().manual_seed(2)) * 255
    got, want = OM.run(OM.load(REF_ONNX), x), _expected(sd, arch, x)
    for k in want:
        assert float((got[k] - want[k]).abs().max()) <= 2e-3 * max(1.0, float(want[k].abs().max())), k
    # and our exporter, fed the same checkpoint, gives the same numbers as the shipped file
    from yunet_amd.onnx_export import export_onnx
    ours = OM.run(OM.load(export_onnx(sd, arch, None, (320, 320))), x)
    for k in want:
        assert float((ours[k] - got[k]).abs().max()) <= 2e-3 * max(1.0, float(got[k].abs().max())), k
