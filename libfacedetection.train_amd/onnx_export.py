"""ONNX export of the YuNet test-time graph: 12 outputs cls_/obj_/bbox_/kps_{8,16,32}.

Reference: tools/yunet2onnx.py:59-113 (`torch.onnx.export` of the detector with
return_loss=False, opset 11, constant folding, output names built at :85-90) and the
`torch.onnx.is_in_onnx_export()` branch of YuNet_Head.forward (yunet_head.py:227-245: each
level's map is permuted to NHWC, flattened to [N, H*W, C], sigmoid on cls and obj).  The
shipped result is /onnx/yunet_n_320_320.onnx: 59 Conv (BatchNorm folded into the depthwise
conv of every unit and into the stem), 18 Relu, 4 MaxPool, 2 Resize (nearest, x2), 2 Add,
12 Transpose + Reshape, 6 Sigmoid.

The `onnx` package is not available here, so the ModelProto is written directly in protobuf
wire format (the handful of message types below; field numbers from onnx.proto3, IR version 6).
The graph is emitted in the reference's node order with the same operator attributes;
`oracle/onnx_mini.py` (test infrastructure) parses and executes both files for the tests.
"""
import struct

import numpy as np
import torch

BN_EPS = 1e-5


# ------------------------------------------------------------------------- protobuf wire format
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode()
    return _key(field, 2) + _varint(len(b)) + b


def _f_float(field, v):
    return _key(field, 5) + struct.pack('<f', float(v))


FLOAT, INT64 = 1, 7      # TensorProto.DataType


def tensor_proto(name, arr):
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype('float32'): FLOAT, np.dtype('int64'): INT64}[arr.dtype]
    out = b''.join(_f_varint(1, d) for d in arr.shape)
    return out + _f_varint(2, dt) + _f_bytes(8, name) + _f_bytes(9, arr.tobytes())


def attr_proto(name, value):
    """AttributeProto: ints (7), int (2), float (1) or string (3)."""
    out = _f_bytes(1, name)
    if isinstance(value, (list, tuple)):
        return out + b''.join(_f_varint(8, v) for v in value) + _f_varint(20, 7)
    if isinstance(value, int):
        return out + _f_varint(3, value) + _f_varint(20, 2)
    if isinstance(value, float):
        return out + _f_float(2, value) + _f_varint(20, 1)
    return out + _f_bytes(4, value) + _f_varint(20, 3)


def node_proto(op, inputs, outputs, **attrs):
    out = b''.join(_f_bytes(1, i) for i in inputs) + b''.join(_f_bytes(2, o) for o in outputs)
    out += _f_bytes(3, f'{op}_{outputs[0]}') + _f_bytes(4, op)
    return out + b''.join(_f_bytes(5, attr_proto(k, v)) for k, v in attrs.items())


def value_info(name, shape, elem=FLOAT):
    dims = b''
    for d in shape:
        dims += _f_bytes(1, _f_bytes(2, d) if isinstance(d, str) else _f_varint(1, d))
    tensor = _f_varint(1, elem) + _f_bytes(2, dims)
    return _f_bytes(1, name) + _f_bytes(2, _f_bytes(1, tensor))


# ----------------------------------------------------------------------------------- the graph
def fold_bn(w, b, sd, prefix):
    """Conv followed by eval-mode BatchNorm -> one conv (what constant folding leaves in the
    reference file): w' = w * gamma / sqrt(var + eps), b' = (b - mean) * gamma / sqrt(var + eps) + beta."""
    scale = sd[prefix + '.weight'] / torch.sqrt(sd[prefix + '.running_var'] + BN_EPS)
    return w * scale.view(-1, 1, 1, 1), (b - sd[prefix + '.running_mean']) * scale + sd[prefix + '.bias']


class _Graph:
    def __init__(self):
        self.nodes, self.inits, self.n = [], [], 0

    def tmp(self):
        self.n += 1
        return f't{self.n}'

    def init(self, name, arr):
        self.inits.append(tensor_proto(name, arr))
        return name

    def conv(self, x, w_name, w, b_name, b, stride=1, group=1):
        k = int(w.shape[-1])
        self.init(w_name, w.detach().cpu().float().numpy())
        self.init(b_name, b.detach().cpu().float().numpy())
        y = self.tmp()
        p = k // 2
        self.nodes.append(node_proto('Conv', [x, w_name, b_name], [y], dilations=[1, 1], group=int(group),
                                     kernel_shape=[k, k], pads=[p, p, p, p], strides=[stride, stride]))
        return y

    def op(self, op, inputs, out=None, **attrs):
        y = out or self.tmp()
        self.nodes.append(node_proto(op, inputs, [y], **attrs))
        return y


def build_graph(sd, arch, input_shape=(320, 320), dynamic=False):
    """state_dict (reference key names, eval statistics) -> serialized GraphProto pieces."""
    g = _Graph()
    sd = {k: v.detach().cpu().float() if torch.is_tensor(v) and v.is_floating_point() else v
          for k, v in sd.items()}

    def unit(x, prefix, with_bn):
        w1, b1 = sd[prefix + '.conv1.weight'], sd[prefix + '.conv1.bias']
        x = g.conv(x, prefix + '.conv1.weight', w1, prefix + '.conv1.bias', b1)
        w2, b2 = sd[prefix + '.conv2.weight'], sd[prefix + '.conv2.bias']
        if with_bn:
            w2, b2 = fold_bn(w2, b2, sd, prefix + '.bn')
            x = g.conv(x, prefix + '.conv2.weight_bn', w2, prefix + '.conv2.bias_bn', b2, group=w2.shape[0])
            return g.op('Relu', [x])
        return g.conv(x, prefix + '.conv2.weight', w2, prefix + '.conv2.bias', b2, group=w2.shape[0])

    # backbone (yunet_backbone.py:33-41)
    w0, b0 = fold_bn(sd['backbone.model0.conv1.weight'], sd['backbone.model0.conv1.bias'], sd, 'backbone.model0.bn1')
    x = g.conv('input', 'backbone.model0.conv1.weight_bn', w0, 'backbone.model0.conv1.bias_bn', b0, stride=2)
    x = g.op('Relu', [x])
    x = unit(x, 'backbone.model0.conv2', True)
    feats = []
    st = arch['stage_channels']
    for i in range(len(st)):
        if i > 0:
            x = unit(x, f'backbone.model{i}.conv1', True)
            x = unit(x, f'backbone.model{i}.conv2', True)
        if i in arch['out_idx']:
            feats.append(x)
        if i in arch['downsample_idx']:
            x = g.op('MaxPool', [x], ceil_mode=0, kernel_shape=[2, 2], pads=[0, 0, 0, 0], strides=[2, 2])
    # TFPN (tfpn.py:33-45)
    g.init('resize_roi', np.zeros(0, np.float32))
    g.init('resize_scales', np.array([1, 1, 2, 2], np.float32))
    for i in range(len(feats) - 1, 0, -1):
        feats[i] = unit(feats[i], f'neck.lateral_convs.{i}', True)
        up = g.op('Resize', [feats[i], 'resize_roi', 'resize_scales'], coordinate_transformation_mode='asymmetric',
                  cubic_coeff_a=-0.75, mode='nearest', nearest_mode='floor')
        feats[i - 1] = g.op('Add', [feats[i - 1], up])
    feats[0] = unit(feats[0], 'neck.lateral_convs.0', True)
    outs = [feats[i] for i in arch['neck_out_idx']]
    # head (yunet_head.py:175-247)
    for l in range(len(outs)):
        for j in range(arch['shared_stacked_convs']):
            outs[l] = unit(outs[l], f'bbox_head.multi_level_share_convs.{l}.{j}', True)
    src = {name: outs for name in ('cls', 'bbox', 'obj', 'kps')}
    if arch.get('stacked_convs', 0) > 0:            # per-level towers (yunet_head.py:191-207)
        towers = {}
        for l in range(len(outs)):
            for t in ('cls', 'reg'):
                x = outs[l]
                for j in range(arch['stacked_convs']):
                    x = unit(x, f'bbox_head.multi_level_{t}_convs.{l}.{j}', True)
                towers[t, l] = x
        src = dict(cls=[towers['cls', l] for l in range(len(outs))])
        src.update({name: [towers['reg', l] for l in range(len(outs))] for name in ('bbox', 'obj', 'kps')})
    maps = {}
    for name in ('cls', 'bbox', 'obj', 'kps'):
        for l in range(len(outs)):
            maps[name, l] = unit(src[name][l], f'bbox_head.multi_level_{name}.{l}', False)
    out_infos = []
    chans = dict(cls=1, obj=1, bbox=4, kps=2 * arch['kps_num'])
    lead = 0 if dynamic else 1                       # Reshape: 0 copies the batch dimension
    for name in ('cls', 'obj', 'bbox', 'kps'):
        shp = g.init(f'shape_{name}', np.array([lead, -1, chans[name]], np.int64))
        for l, s in enumerate(arch['strides']):
            t = g.op('Transpose', [maps[name, l]], perm=[0, 2, 3, 1])
            final = f'{name}_{s}'
            if name in ('cls', 'obj'):
                r = g.op('Reshape', [t, shp])
                g.op('Sigmoid', [r], out=final)
            else:
                g.op('Reshape', [t, shp], out=final)
            n_pri = 'dim' if dynamic else (input_shape[0] // s) * (input_shape[1] // s)
            out_infos.append(value_info(final, ['batch' if dynamic else 1, n_pri, chans[name]]))
    in_info = value_info('input', ['batch', 3, 'height', 'width'] if dynamic
                         else [1, 3, int(input_shape[0]), int(input_shape[1])])
    return g, in_info, out_infos


def export_onnx(state_dict, arch, path=None, input_shape=(320, 320), dynamic=False, opset=11,
                producer='libfacedetection.train_amd'):
    """-> serialized ModelProto bytes (also written to `path`)."""
    if opset != 11:
        raise NotImplementedError('the graph uses the opset-11 form of Resize (scales input), like the reference')
    if input_shape[0] % 32 or input_shape[1] % 32:
        raise ValueError('input height / width must be multiples of 32')
    g, in_info, out_infos = build_graph(state_dict, arch, input_shape, dynamic)
    graph = b''.join(_f_bytes(1, n) for n in g.nodes) + _f_bytes(2, 'yunet') + \
        b''.join(_f_bytes(5, t) for t in g.inits) + _f_bytes(11, in_info) + \
        b''.join(_f_bytes(12, o) for o in out_infos)
    model = _f_varint(1, 6) + _f_bytes(2, producer) + _f_bytes(3, '0.2') + _f_bytes(7, graph) + \
        _f_bytes(8, _f_bytes(1, '') + _f_varint(2, opset))
    if path:
        with open(path, 'wb') as f:
            f.write(model)
    return model
