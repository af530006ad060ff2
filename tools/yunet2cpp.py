#!/usr/bin/env python
"""Same CLI as the reference's tools/yunet2cpp.py: checkpoint -> facedetectcnn-data.cpp
(BatchNorm-folded weights in libfacedetection's layouts).

    python tools/yunet2cpp.py CONFIG CHECKPOINT [--output-file ./work_dirs/facedetectcnn-data.cpp]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import yunet_amd  # noqa: E402
from yunet_amd.export import to_cpp  # noqa: E402


def main():
    p = argparse.ArgumentParser(description='Convert YuNet checkpoints to libfacedetection dnn data')
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--output-file', default='./work_dirs/facedetectcnn-data.cpp')
    p.add_argument('--no_summary', action='store_true')
    a = p.parse_args()
    cfg = yunet_amd.Config.fromfile(a.config)
    model = yunet_amd.build_detector(cfg.model)
    ck = torch.load(a.checkpoint, map_location='cpu', weights_only=False)
    model.load_state_dict(ck['state_dict'] if 'state_dict' in ck else ck, strict=True)
    if not a.no_summary:
        n = sum(p.numel() for p in model.parameters())
        print(f"{'=' * 30}\nParams: {n}\n{'=' * 30}")
    os.makedirs(os.path.dirname(os.path.abspath(a.output_file)), exist_ok=True)
    with open(a.output_file, 'w') as f:
        f.write(to_cpp(model))
    print(f'Convert successful!\nFrom {a.config} with {a.checkpoint}\nTo {a.output_file}')


if __name__ == '__main__':
    main()
