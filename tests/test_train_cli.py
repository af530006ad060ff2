"""tools/train.py: the reference's command line (tools/train.py:24-160) -- argument surface, the KEY=VALUE
grammar of --cfg-options (mmcv.DictAction), work_dir / resume / gpu-id rules, --auto-scale-lr, the config dump --
and main() up to the hand-over to train_detector (which the GPU tests cover)."""
import importlib.util
import os
import sys
import warnings

import pytest
import torch

import yunet_amd
from yunet_amd.registry import Config, DictAction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool():
    spec = importlib.util.spec_from_file_location('yunet_train_tool', os.path.join(ROOT, 'tools', 'train.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


T = load_tool()
CFG = os.path.join(ROOT, 'configs', 'yunet_n.py')


@pytest.mark.parametrize('text,want', [
    ('1', 1), ('0.1', 0.1), ('x', 'x'), ('false', False), ('TRUE', True), ('1e-3', 1e-3),
    ('[1,2]', [1, 2]), ('1,2', [1, 2]), ('(1,2)', (1, 2)), ('[(a,b),(c,d)]', [('a', 'b'), ('c', 'd')]),
    ("'[a, b]'", ['a', 'b']), ('[]', []), ('[a,[b,c],(d,e)]', ['a', ['b', 'c'], ('d', 'e')]),
    ('(320,320)', (320, 320)), ('work_dirs/x', 'work_dirs/x'),
])
def test_cfg_option_value_grammar(text, want):
    got = DictAction.parse_value(text)
    assert got == want and type(got) is type(want)


def test_argument_surface_and_config_rules(tmp_path):
    a = T.parse_args([CFG])
    assert (a.gpu_id, a.launcher, a.seed, a.cfg_options, a.auto_resume, a.no_validate) == (0, 'none', None, None, False, False)
    cfg = T.prepare_config(a)
    assert cfg.work_dir == os.path.join('./work_dirs', 'yunet_n') and cfg.gpu_ids == [0] and cfg.auto_resume is False
    a = T.parse_args([CFG, '--work-dir', str(tmp_path), '--resume-from', 'ck.pth', '--auto-resume', '--gpu-id', '3',
                      '--seed', '7', '--diff-seed', '--deterministic', '--no-validate', '--launcher', 'pytorch',
                      '--local_rank', '2', '--cfg-options', 'optimizer.lr=0.02', 'lr_config.step=[10,20]',
                      'fp16.loss_scale=512.', 'data.samples_per_gpu=8', 'model.bbox_head.use_kps=true'])
    cfg = T.prepare_config(a)
    assert cfg.work_dir == str(tmp_path) and cfg.resume_from == 'ck.pth' and cfg.auto_resume is True
    assert cfg.gpu_ids == [3] and cfg.optimizer.lr == 0.02 and cfg.lr_config.step == [10, 20]
    assert cfg.fp16.loss_scale == 512.0 and cfg.data.samples_per_gpu == 8 and cfg.model.bbox_head.use_kps is True
    assert a.seed == 7 and a.diff_seed and a.no_validate and a.local_rank == 2
    with warnings.catch_warnings(record=True) as w:          # the deprecated spellings keep working
        warnings.simplefilter('always')
        assert T.prepare_config(T.parse_args([CFG, '--gpus', '4'])).gpu_ids == [0]
        assert T.prepare_config(T.parse_args([CFG, '--gpu-ids', '5', '6'])).gpu_ids == [5]
        o = T.parse_args([CFG, '--options', 'optimizer.lr=0.5'])
        assert o.cfg_options == {'optimizer.lr': 0.5}
        assert len(w) == 3
    with pytest.raises(ValueError):
        T.parse_args([CFG, '--options', 'a=1', '--cfg-options', 'b=2'])
    with pytest.raises(SystemExit):
        T.parse_args([CFG, '--gpus', '1', '--gpu-id', '1'])                       # mutually exclusive
    with pytest.raises(SystemExit):
        T.parse_args([CFG, '--cfg-options', 'novalue'])


def test_auto_scale_lr():
    from yunet_amd.runner import auto_scale_lr
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        cfg = T.prepare_config(T.parse_args([CFG, '--auto-scale-lr']))            # the shipped config has no such section
        assert len(w) == 1 and auto_scale_lr(cfg, False, lambda *a: None) == 0.01
    (open('/tmp/asl_cfg.py', 'w')).write(open(CFG).read() + '\nauto_scale_lr = dict(enable=False, base_batch_size=64)\n')
    cfg = T.prepare_config(T.parse_args(['/tmp/asl_cfg.py']))
    assert auto_scale_lr(cfg, False, lambda *a: None) == 0.01                     # present but not enabled
    cfg = T.prepare_config(T.parse_args(['/tmp/asl_cfg.py', '--auto-scale-lr', '--cfg-options', 'data.samples_per_gpu=16']))
    assert cfg.auto_scale_lr.enable is True
    msgs = []
    assert auto_scale_lr(cfg, False, msgs.append) == pytest.approx(0.01 * 16 / 64) and cfg.optimizer.lr == pytest.approx(0.0025)
    assert any('automatically scaled' in m for m in msgs)
    cfg = T.prepare_config(T.parse_args(['/tmp/asl_cfg.py', '--auto-scale-lr', '--cfg-options', 'data.samples_per_gpu=64']))
    assert auto_scale_lr(cfg, False, lambda *a: None) == 0.01                     # batch == base: unchanged


def test_config_dump_is_loadable_python(tmp_path):
    cfg = Config.fromfile(CFG)
    cfg.merge_from_dict({'fp16.loss_scale': 512., 'x.y': (1, [2, dict(a=None)], 'q"uote'), 'z': range(2), 't': (5,)})
    cfg.dump(str(tmp_path / 'd.py'))
    back = Config.fromfile(str(tmp_path / 'd.py'))
    a = {k: v for k, v in cfg.items() if k != 'filename'}
    a['z'] = [0, 1]
    assert a == {k: v for k, v in back.items() if k != 'filename'}
    assert cfg.dump() == cfg.pretty_text and 'filename' not in cfg.pretty_text.split('=')[0]
    assert type(back.x.y) is tuple and type(back.t) is tuple and back.model.bbox_head.type == 'YuNet_Head'


def test_main_hands_over_to_train_detector(tmp_path, monkeypatch):
    """main() on a synthetic data source with the GPU-side calls replaced: everything the tool does itself --
    parse, merge, work dir + config dump, seeding, model / source construction, the train_detector arguments."""
    seen = {}

    def fake_train(model, ds, cfg, **kw):
        seen.update(model=model, ds=ds, cfg=cfg, kw=kw)
        return ['history']

    monkeypatch.setattr(T.R, 'train_detector', fake_train)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda i: seen.setdefault('device', i))
    out = T.main([CFG, '--work-dir', str(tmp_path / 'w'), '--seed', '5', '--no-validate', '--max-iters', '3', '--gpu-id', '1',
                  '--cfg-options', 'data.samples_per_gpu=4', 'data.train.type=SyntheticWiderFace',
                  'data.train.img_scale=(160,160)', 'data.train.iters_per_epoch=2'])
    assert out == ['history'] and seen['device'] == 1
    assert type(seen['model']).__name__ == 'YuNet' and seen['ds'].bs == 4 and seen['ds'].iters_per_epoch == 2
    kw = seen['kw']
    assert kw['validate'] is False and kw['max_iters'] == 3 and kw['distributed'] is False
    assert kw['meta']['seed'] == 5 and 'YuNet_Head' in kw['meta']['config'] and kw['meta']['exp_name'] == 'yunet_n.py'
    dumped = Config.fromfile(str(tmp_path / 'w' / 'yunet_n.py'))
    assert dumped.data.samples_per_gpu == 4 and dumped.work_dir == str(tmp_path / 'w') and dumped.gpu_ids == [1]


def test_no_function_reads_an_undefined_name():
    """Static check over every python file of the product, the tools, the bench and the oracle: a function that
    reads a name no scope defines (tools/train.py once passed `a.no_validate` for `args.no_validate`, on a line no
    test executed) fails here instead of at the user's first run."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'dbg'))
    import undefined_names as U
    files = [f for pat in ('libfacedetection.train_amd/*.py', 'tools/*.py', 'tools/dbg/*.py', 'oracle/*.py', '*.py')
             for f in glob.glob(os.path.join(ROOT, pat))]
    assert len(files) > 40
    bad = [(os.path.relpath(f, ROOT),) + b for f in files for b in U.check(f)]
    assert not bad, bad


def test_mmdet_datasets_relocates_the_data_root(tmp_path, monkeypatch):
    """tools/train.py:113 update_data_root: MMDET_DATASETS replaces cfg.data_root inside the strings of cfg.data."""
    (tmp_path / 'c.py').write_text(open(CFG).read() + """
data_root = 'data/widerface/'
data = dict(samples_per_gpu=4,
            train=dict(type='RetinaFaceDataset', ann_file='data/widerface/labelv2/train/labelv2.txt',
                       img_prefix='data/widerface/WIDER_train/images/', pipeline=[dict(type='X', root='data/widerface/')]),
            val=dict(ann_file='data/widerface/labelv2/val/labelv2.txt', other='elsewhere/x.txt'))
""")
    cfg = T.prepare_config(T.parse_args([str(tmp_path / 'c.py')]))
    assert cfg.data.train.ann_file == 'data/widerface/labelv2/train/labelv2.txt' and cfg.data_root == 'data/widerface/'
    monkeypatch.setenv('MMDET_DATASETS', '/mnt/sets/wf/')
    cfg = T.prepare_config(T.parse_args([str(tmp_path / 'c.py')]))
    assert cfg.data_root == '/mnt/sets/wf/'
    assert cfg.data.train.ann_file == '/mnt/sets/wf/labelv2/train/labelv2.txt'
    assert cfg.data.train.img_prefix == '/mnt/sets/wf/WIDER_train/images/'
    assert cfg.data.val.ann_file == '/mnt/sets/wf/labelv2/val/labelv2.txt' and cfg.data.val.other == 'elsewhere/x.txt'
    assert cfg.data.train.pipeline[0]['root'] == 'data/widerface/'           # lists are not walked (as in the reference)
