# YuNet_n on MI355X.  Same top-level names / model dict as the reference config surface
# (mmcv python-dict config); the data section points at the synthetic WIDER-Face-shaped
# generator because the cv2/WIDER pipeline is outside the accelerated path.
optimizer = dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0005)
optimizer_config = dict(grad_clip=None)
lr_mult = 8
lr_config = dict(policy='step', warmup='linear', warmup_iters=1500, warmup_ratio=0.001,
                 step=[50 * lr_mult, 68 * lr_mult])
runner = dict(type='EpochBasedRunner', max_epochs=80 * lr_mult)
checkpoint_config = dict(interval=80)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook'), dict(type='TensorboardLoggerHook')])   # as the reference's configs/yunet_n.py:14-17
dist_params = dict(backend='nccl')     # RCCL on ROCm
log_level = 'INFO'
load_from = None
resume_from = None
workflow = [('train', 1)]

data = dict(samples_per_gpu=256, workers_per_gpu=0,
            train=dict(type='SyntheticWiderFace', img_scale=(320, 320), iters_per_epoch=403))

# The reference's train pipeline (configs/yunet_n.py:36-56 there), executed on the GPU by
# yunet_amd.pipelines.DevicePipeline.  To train through it from decoded uint8 sources:
#   --cfg-options data.train.type=SyntheticSourceImages
train_pipeline = [
    dict(type='LoadImageFromFile', to_float32=True),
    dict(type='LoadAnnotations', with_bbox=True, with_keypoints=True),
    dict(type='RandomSquareCrop', crop_choice=[0.5, 0.7, 0.9, 1.1, 1.3, 1.5]),
    dict(type='Resize', img_scale=(320, 320), keep_ratio=False),
    dict(type='RandomFlip', flip_ratio=0.5),
    dict(type='Normalize', mean=[0., 0., 0.], std=[1., 1., 1.], to_rgb=False),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_bboxes_ignore', 'gt_keypointss']),
]
data['train']['pipeline'] = train_pipeline

_stages_n = [[3, 16, 16], [16, 64], [64, 64], [64, 64], [64, 64], [64, 64]]
model = dict(
    type='YuNet',
    backbone=dict(type='YuNetBackbone', stage_channels=_stages_n, downsample_idx=[0, 2, 3, 4],
                  out_idx=[3, 4, 5]),
    neck=dict(type='TFPN', in_channels=[64, 64, 64], out_idx=[0, 1, 2]),
    bbox_head=dict(
        type='YuNet_Head', num_classes=1, in_channels=64, shared_stacked_convs=1,
        stacked_convs=0, feat_channels=64,
        prior_generator=dict(type='MlvlPointGenerator', offset=0, strides=[8, 16, 32]),
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='sum', loss_weight=1.0),
        loss_bbox=dict(type='EIoULoss', loss_weight=5.0, reduction='sum'),
        use_kps=True, kps_num=5,
        loss_kps=dict(type='SmoothL1Loss', beta=0.1111111111111111, loss_weight=0.1),
        loss_obj=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='sum', loss_weight=1.0)),
    train_cfg=dict(assigner=dict(type='SimOTAAssigner', center_radius=2.5)),
    test_cfg=dict(nms_pre=-1, min_bbox_size=0, score_thr=0.02,
                  nms=dict(type='nms', iou_threshold=0.45), max_per_img=-1))
