"""TEST INFRASTRUCTURE ONLY -- a minimal ONNX reader and executor (the `onnx` / `onnxruntime`
packages are not installed here).

`load(path)` decodes the protobuf wire format of a ModelProto into plain dicts (field numbers from
onnx.proto3); `run(model, x)` executes the dozen operator types a YuNet export contains with torch
CPU ops.  Used by tests/test_onnx_export.py to (a) compare the structure of the exporter's output
with the reference's shipped /onnx/yunet_n_320_320.onnx and (b) check both files numerically against
the oracle's eval-mode forward."""
import struct

import numpy as np
import torch
import torch.nn.functional as F


def _rv(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def fields(b):
    i, out = 0, []
    while i < len(b):
        k, i = _rv(b, i)
        f, w = k >> 3, k & 7
        if w == 0:
            v, i = _rv(b, i)
        elif w == 1:
            v, i = b[i:i + 8], i + 8
        elif w == 2:
            n, i = _rv(b, i)
            v, i = b[i:i + n], i + n
        elif w == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f'wire type {w}')
        out.append((f, w, v))
    return out


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _tensor(b):
    dims, dt, name, raw, f32, i64 = [], None, '', None, [], []
    for f, w, v in fields(b):
        if f == 1:
            if w == 0:
                dims.append(v)
            else:                                   # packed
                j = 0
                while j < len(v):
                    d, j = _rv(v, j)
                    dims.append(d)
        elif f == 2:
            dt = v
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
        elif f == 4:
            f32 += list(struct.unpack(f'<{len(v) // 4}f', v)) if w == 2 else [struct.unpack('<f', v)[0]]
        elif f == 7:
            if w == 0:
                i64.append(_signed(v))
            else:
                j = 0
                while j < len(v):
                    d, j = _rv(v, j)
                    i64.append(_signed(d))
    np_dt = {1: np.float32, 7: np.int64}[dt]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_dt)
    else:
        arr = np.array(f32 if dt == 1 else i64, dtype=np_dt)
    return name, arr.reshape(dims).copy()


def _attr(b):
    name, val, ints, floats = None, None, [], []
    for f, w, v in fields(b):
        if f == 1:
            name = v.decode()
        elif f == 2:
            val = struct.unpack('<f', v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = v.decode()
        elif f == 8:
            if w == 0:
                ints.append(_signed(v))
            else:
                j = 0
                while j < len(v):
                    d, j = _rv(v, j)
                    ints.append(_signed(d))
        elif f == 7:
            floats += list(struct.unpack(f'<{len(v) // 4}f', v)) if w == 2 else [struct.unpack('<f', v)[0]]
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def _value_info(b):
    name, shape = '', []
    for f, w, v in fields(b):
        if f == 1:
            name = v.decode()
        elif f == 2:
            for f2, _, t in fields(v):
                if f2 == 1:
                    for f3, _, s in fields(t):
                        if f3 == 2:
                            for _, _, d in fields(s):
                                df = fields(d)
                                shape.append(df[0][2] if df and df[0][1] == 0 else
                                             (df[0][2].decode() if df else '?'))
    return name, shape


def load(path_or_bytes):
    b = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
    m = dict(ir_version=None, producer=None, opset=None, nodes=[], inits={}, inputs=[], outputs=[])
    for f, w, v in fields(b):
        if f == 1:
            m['ir_version'] = v
        elif f == 2:
            m['producer'] = v.decode()
        elif f == 8:
            for f2, _, x in fields(v):
                if f2 == 2:
                    m['opset'] = x
        elif f == 7:
            for f2, _, x in fields(v):
                if f2 == 1:
                    n = dict(op=None, inputs=[], outputs=[], attrs={})
                    for f3, _, y in fields(x):
                        if f3 == 1:
                            n['inputs'].append(y.decode())
                        elif f3 == 2:
                            n['outputs'].append(y.decode())
                        elif f3 == 4:
                            n['op'] = y.decode()
                        elif f3 == 5:
                            k, val = _attr(y)
                            n['attrs'][k] = val
                    m['nodes'].append(n)
                elif f2 == 5:
                    name, arr = _tensor(x)
                    m['inits'][name] = arr
                elif f2 == 11:
                    m['inputs'].append(_value_info(x))
                elif f2 == 12:
                    m['outputs'].append(_value_info(x))
    return m


def run(m, x):
    """Execute the graph on a float32 tensor [N,3,H,W] -> {output name: tensor}."""
    env = {k: torch.from_numpy(v) for k, v in m['inits'].items()}
    env[[n for n, _ in m['inputs'] if n not in m['inits']][0]] = x
    for n in m['nodes']:
        i = [env[k] if k else None for k in n['inputs']]
        a = n['attrs']
        op = n['op']
        if op == 'Conv':
            assert a['dilations'] == [1, 1] and a['pads'][0] == a['pads'][2]
            y = F.conv2d(i[0], i[1], i[2] if len(i) > 2 else None, stride=a['strides'], padding=a['pads'][:2],
                         groups=a['group'])
        elif op == 'Relu':
            y = F.relu(i[0])
        elif op == 'Sigmoid':
            y = torch.sigmoid(i[0])
        elif op == 'MaxPool':
            assert not a.get('ceil_mode', 0)
            y = F.max_pool2d(i[0], a['kernel_shape'], a['strides'], a['pads'][:2])
        elif op == 'Resize':
            assert a['mode'] == 'nearest' and a['coordinate_transformation_mode'] == 'asymmetric' \
                and a['nearest_mode'] == 'floor'
            sc = i[2].tolist()
            assert sc[:2] == [1.0, 1.0]
            y = F.interpolate(i[0], scale_factor=(sc[2], sc[3]), mode='nearest')
        elif op == 'Add':
            y = i[0] + i[1]
        elif op == 'Transpose':
            y = i[0].permute(*a['perm'])
        elif op == 'Reshape':
            shp = [int(v) for v in i[1].tolist()]
            shp = [i[0].shape[k] if v == 0 else v for k, v in enumerate(shp)]
            y = i[0].reshape(shp)
        elif op == 'Shape':
            y = torch.tensor(i[0].shape, dtype=torch.int64)
        elif op == 'Gather':
            y = i[0][i[1]] if a.get('axis', 0) == 0 else None
        elif op == 'Unsqueeze':
            y = i[0].reshape([1] * len(a['axes']) + list(i[0].shape)) if i[0].dim() == 0 else i[0].unsqueeze(a['axes'][0])
        elif op == 'Concat':
            y = torch.cat([t.reshape(-1) if t.dim() == 0 else t for t in i], dim=a['axis'])
        elif op == 'Constant':
            raise NotImplementedError('Constant nodes')
        else:
            raise NotImplementedError(op)
        env[n['outputs'][0]] = y
    return {name: env[name] for name, _ in m['outputs']}


def structure(m):
    """Order-preserving structural summary: (op, sorted attrs) per node + output names / shapes."""
    return dict(ops=[(n['op'], tuple(sorted((k, tuple(v) if isinstance(v, list) else v)
                                            for k, v in n['attrs'].items()))) for n in m['nodes']],
                outputs=m['outputs'], inputs=[i for i in m['inputs'] if i[0] not in m['inits']],
                opset=m['opset'], ir_version=m['ir_version'])
