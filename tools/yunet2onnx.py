#!/usr/bin/env python
"""Same CLI as the reference's tools/yunet2onnx.py: checkpoint -> ONNX graph with the 12 outputs
cls_/obj_/bbox_/kps_{8,16,32} (BatchNorm folded, opset 11).

    python tools/yunet2onnx.py CONFIG CHECKPOINT [--output-file out.onnx] [--shape 640 640] [--dynamic-export]

Written without the `onnx` / torch.onnx exporters (protobuf wire format, yunet_amd/onnx_export.py);
--verify executes the written file with the repo's minimal runtime when oracle/ is present.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import yunet_amd  # noqa: E402
from yunet_amd.onnx_export import export_onnx  # noqa: E402


def main():
    p = argparse.ArgumentParser(description='Convert YuNet checkpoints to ONNX')
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--output-file', default=None)
    p.add_argument('--opset-version', type=int, default=11)
    p.add_argument('--shape', type=int, nargs='+', default=[640, 640])
    p.add_argument('--dynamic-export', action='store_true')
    a = p.parse_args()
    cfg = yunet_amd.Config.fromfile(a.config)
    model = yunet_amd.build_detector(cfg.model)
    ck = torch.load(a.checkpoint, map_location='cpu', weights_only=False)
    model.load_state_dict(ck['state_dict'] if 'state_dict' in ck else ck, strict=True)
    shape = (a.shape[0], a.shape[0]) if len(a.shape) == 1 else tuple(a.shape[:2])
    out = a.output_file
    if out is None:                                      # tools/yunet2onnx.py:226-232 naming
        stem = os.path.splitext(os.path.basename(a.checkpoint))[0]
        out = os.path.join('./onnx', f"{stem}_{'dynamic' if a.dynamic_export else f'{shape[0]}_{shape[1]}'}.onnx")
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    export_onnx(model.state_dict(), model.arch(), out, input_shape=shape, dynamic=a.dynamic_export,
                opset=a.opset_version)
    print(f'Successfully exported ONNX model: {out}')


if __name__ == '__main__':
    main()
