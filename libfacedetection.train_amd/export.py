"""Export of trained weights in the reference's deployment format.

`to_cpp(model)` writes the `facedetectcnn-data.cpp` text that ShiqiYu/libfacedetection compiles
in: every Conv_head / ConvDPUnit / Conv4layerBlock in module order, BatchNorm folded into the
preceding convolution (eval statistics), and the weight layouts the C++ inference engine expects
(reference tools/yunet2cpp.py:26-154):
    stem 3x3x3      [co][27] re-ordered to (ky,kx)-major / channel-minor and padded to 32 per row
    pointwise 1x1   [co][ci]
    depthwise 3x3   transposed to [9][c]
Numbers are printed with format(x, '.3g') + 'f' ('.f' for integers), exactly as the reference.
"""
import numpy as np
import torch

HEADER = ('// Auto generated data file\n'
          '// Copyright (c) 2018-2023, Shiqi Yu, all rights reserved.\n'
          '#include "facedetectcnn.h"\n\n')


def _num(x, precision):
    s = format(x, precision)
    return s + ('.f' if '.' not in s and 'e' not in s else 'f')


def fold_bn(conv, bn):
    """(weight, bias) of conv followed by eval-mode bn (tools/yunet2cpp.py:44-53)."""
    scales = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    bias = (conv.bias.detach() - bn.running_mean) * scales + bn.bias.detach()
    weight = conv.weight.detach() * scales.reshape(-1, 1, 1, 1)
    return weight, bias


def _entry(weight, bias, name, depthwise=False, with_bn=False, first3x3x3=False, precision='.3g'):
    co, ci, kw, kh = weight.shape
    w = weight.detach().cpu().numpy().astype(np.float32)
    if first3x3x3:
        src = w.reshape(co, 27)
        out = np.zeros((co, 32), dtype=np.float64)
        for off in range(27):                      # (ci, ky, kx) -> (ky, kx, ci)
            out[:, (off % 9) * 3 + off // 9] = src[:, off]
        flat, size, cin = out.reshape(-1), f'{co}*32*1*1', 32
    elif depthwise:
        flat, size, cin = w.reshape(-1, 9).transpose().reshape(-1), f'{co}*{ci}*{kw}*{kh}', co
    else:
        flat, size, cin = w.reshape(-1), f'{co}*{ci}*{kw}*{kh}', ci
    b = bias.detach().cpu().numpy().astype(np.float32).reshape(-1)
    return dict(wname=f'{name}_weight', wsize=size, w=','.join(_num(v, precision) for v in flat),
                bname=f'{name}_bias', bsize=str(co), b=','.join(_num(v, precision) for v in b),
                with_bn=with_bn, dw=depthwise, cin=cin, cout=co)


def _dp_unit(unit, name, out):
    out.append(_entry(unit.conv1.weight, unit.conv1.bias, name + '_pw'))
    if unit.withBNRelu:
        w, b = fold_bn(unit.conv2, unit.bn)
        out.append(_entry(w, b, name + '_dw', depthwise=True, with_bn=True))
    else:
        out.append(_entry(unit.conv2.weight, unit.conv2.bias, name + '_dw', depthwise=True))


def collect(model):
    """Depth-first over named_children(): the first supported block type on a path is exported
    as a whole (tools/yunet2cpp.py:117-124)."""
    entries = []

    def walk(mod, prefix):
        for name, child in mod.named_children():
            path = f'{prefix}__{name}' if prefix else name
            kind = type(child).__name__
            if kind == 'Conv_head':
                w, b = fold_bn(child.conv1, child.bn1)
                entries.append(_entry(w, b, path + '_pw', with_bn=True, first3x3x3=True))
                _dp_unit(child.conv2, path + '_dp', entries)
            elif kind == 'ConvDPUnit':
                _dp_unit(child, path, entries)
            elif kind == 'Conv4layerBlock':
                _dp_unit(child.conv1, path + '_dp1', entries)
                _dp_unit(child.conv2, path + '_dp2', entries)
            else:
                walk(child, path)
    walk(model, '')
    return entries


def to_cpp(model):
    """The complete facedetectcnn-data.cpp text for `model` (put in eval mode)."""
    model.eval()
    ents = collect(model)
    if not ents:
        raise ValueError('no Conv_head / ConvDPUnit / Conv4layerBlock modules found')
    cb = lambda v: 'true' if v else 'false'    # noqa: E731
    text = HEADER
    for d in ents:
        text += f"float {d['wname']}[{d['wsize']}] = {{{d['w']}}};\n"
        text += f"float {d['bname']}[{d['bsize']}] = {{{d['b']}}};\n"
    text += '\n//(in_channels, out_channels, is_depthwise, is_pointwise, with_bn, weight_ptr, bias_ptr)\n'
    text += f'ConvInfoStruct param_pConvInfo[{len(ents)}] = {{\n'
    rows = [f"\t{{{d['cin']}, {d['cout']}, {cb(d['dw'])}, {cb(not d['dw'])}, {cb(d['with_bn'])}, "
            f"{d['wname']}, {d['bname']}}}" for d in ents]
    text += ',\n'.join(rows) + '\n};'
    return text
