"""CPU: the labelv2 reader (datasets.RetinaFaceDataset) against the UNMODIFIED reference class
(mmdet/datasets/retinaface.py executed under a two-module package skeleton: builder.DATASETS and
custom.CustomDataset are the only names it imports) on the reference's own
data/widerface/labelv2/val/labelv2.txt, plus format edge cases on a hand-written file."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

REF = '/root/reference'
LABELS = os.path.join(REF, 'data', 'widerface', 'labelv2', 'val', 'labelv2.txt')


def reference_class():
    pkg = types.ModuleType('refds')
    pkg.__path__ = []
    b = types.ModuleType('refds.builder')

    class _Reg:
        def register_module(self):
            return lambda c: c
    b.DATASETS = _Reg()
    c = types.ModuleType('refds.custom')
    c.CustomDataset = object
    pc = types.ModuleType('pycocotools')
    pc.__version__ = '12.0.2'
    sys.modules.update({'refds': pkg, 'refds.builder': b, 'refds.custom': c})
    sys.modules.setdefault('pycocotools', pc)
    spec = importlib.util.spec_from_file_location('refds.retinaface',
                                                  os.path.join(REF, 'mmdet', 'datasets', 'retinaface.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.RetinaFaceDataset


def _ref_instance(ann_file, min_size, test_mode):
    cls = reference_class()
    o = object.__new__(cls)
    o.NK, o.cat2label, o.min_size, o.test_mode = 5, {'FG': 0}, min_size, test_mode
    o.data_infos = o.load_annotations(ann_file)
    return o


@pytest.mark.skipif(not os.path.exists(LABELS), reason='reference tree not present')
@pytest.mark.parametrize('test_mode', [True])
def test_labelv2_val_matches_reference(test_mode):
    import yunet_amd
    ds = yunet_amd.build_dataset(dict(type='RetinaFaceDataset', ann_file=LABELS, img_prefix='', pipeline=[],
                                      test_mode=test_mode))
    ref = _ref_instance(LABELS, None, test_mode)
    assert len(ds) == len(ref.data_infos) == 3226
    for i in range(0, len(ds), 7):
        a, b = ds.data_infos[i], ref.data_infos[i]
        assert (a['filename'], a['width'], a['height']) == (b['filename'], b['width'], b['height'])
        x, y = ds.get_ann_info(i), ref.get_ann_info(i)
        assert set(x) == set(y)
        for k in x:
            assert x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and np.array_equal(x[k], y[k]), (i, k)


TRAIN_TXT = """# 0--Parade/a.jpg 1024 678
10 20 110 140 30.5 40.5 0.0 60 41 0.0 45 70 0.0 35 90 1.0 58 92 1.0 0.87
200 210 204 215 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 0.3
300 310 340 350 1
400 410 440 450 0
# 1--Handshaking/b.jpg 500 700
1 2 3 4 -1 -1 -1 5 6 0.0 -1 -1 -1 -1 -1 -1 -1 -1 -1 0.5
# 2--Empty/c.jpg 10 10
# 0--Parade/a.jpg 1024 678
7 8 90 100 0
"""


def test_labelv2_train_format_edge_cases(tmp_path):
    """Landmark flags, all -1 rows, ignore flags, min_size, empty images, a repeated header."""
    import yunet_amd.datasets as DS
    f = tmp_path / 'labelv2.txt'
    f.write_text(TRAIN_TXT)
    infos = DS.load_labelv2(str(f), min_size=8)
    assert [it['filename'] for it in infos] == ['0--Parade/a.jpg', '1--Handshaking/b.jpg']   # empty image dropped
    a = DS.ann_info(infos[0])                       # the repeated header replaced the first block
    assert a['bboxes'].tolist() == [[7, 8, 90, 100]] and a['bboxes_ignore'].shape == (0, 4)
    b = DS.ann_info(infos[1])
    assert b['bboxes'].shape == (0, 4) and b['bboxes_ignore'].tolist() == [[1, 2, 3, 4]]    # 2 px < min_size
    one = DS.parse_ann_line('10 20 110 140 30.5 40.5 0.0 60 41 0.0 45 70 0.0 35 90 1.0 58 92 1.0 0.87')
    assert one['kps'][:, 2].tolist() == [1.0] * 5 and not one['ignore']
    none = DS.parse_ann_line('200 210 204 215 ' + '-1 ' * 15 + '0.3')
    assert none['kps'][:, 2].tolist() == [0.0] * 5
    assert DS.parse_ann_line('300 310 340 350 1')['ignore'] and not DS.parse_ann_line('300 310 340 350 0')['ignore']
    with pytest.raises(AssertionError):
        DS.parse_ann_line('1 2 3 4')                 # bare boxes are test annotations
    assert DS.parse_ann_line('1 2 3 4', test_mode=True)['bbox'].tolist() == [1, 2, 3, 4]
    if os.path.exists(LABELS):
        ref = _ref_instance(str(f), 8, False)
        assert [it['filename'] for it in ref.data_infos] == [it['filename'] for it in infos]
        for i in range(len(infos)):
            x, y = DS.ann_info(infos[i]), ref.get_ann_info(i)
            for k in x:
                assert np.array_equal(x[k], y[k]) and x[k].dtype == y[k].dtype, (i, k)


def test_dataset_decodes_to_device_pipeline_sources(tmp_path):
    """__getitem__ : PIL decode -> uint8 BGR HWC + annotations, the input contract of SourceBatch."""
    from PIL import Image
    import yunet_amd
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / 'img' / '0--Parade')
    rgb = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / 'img' / '0--Parade' / 'a.png')
    (tmp_path / 'l.txt').write_text('# 0--Parade/a.png 64 48\n4 5 30 40 ' + '10 10 1.0 ' * 5 + '0.9\n')
    ds = yunet_amd.build_dataset(dict(type='RetinaFaceDataset', ann_file=str(tmp_path / 'l.txt'),
                                      img_prefix=str(tmp_path / 'img'), pipeline=[]))
    s = ds[0]
    assert s['img'].dtype == np.uint8 and s['img'].shape == (48, 64, 3)
    assert np.array_equal(s['img'][:, :, ::-1], rgb)           # BGR, like cv2.imread
    assert s['gt_bboxes'].shape == (1, 4) and s['gt_keypointss'].shape == (1, 5, 3)


def test_dataset_min_side_filter_exif_and_decode_ahead(tmp_path):
    """CustomDataset._filter_imgs (training drops images with a side below 32 px, custom.py:176-185);
    the EXIF orientation is applied like cv2.imread; RetinaFaceSource's decode-ahead pool returns the
    same samples, in the same order, as the synchronous path."""
    from PIL import Image
    import yunet_amd
    import yunet_amd.datasets as DS
    rng = np.random.default_rng(1)
    os.makedirs(tmp_path / 'img')
    txt = ''
    for i in range(6):
        rgb = rng.integers(0, 256, (40 + i, 50, 3), dtype=np.uint8)
        Image.fromarray(rgb).save(tmp_path / 'img' / f'{i}.png')
        txt += f'# {i}.png 50 {40 + i}\n4 5 30 36 ' + '10 10 1.0 ' * 5 + '0.9\n'
    txt += '# tiny.png 31 200\n1 2 20 30 ' + '10 10 1.0 ' * 5 + '0.9\n'
    (tmp_path / 'l.txt').write_text(txt)
    cfg = dict(type='RetinaFaceDataset', ann_file=str(tmp_path / 'l.txt'), img_prefix=str(tmp_path / 'img'), pipeline=[])
    ds = yunet_amd.build_dataset(cfg)
    assert len(ds) == 6 and all(it['filename'] != 'tiny.png' for it in ds.data_infos)
    assert len(yunet_amd.build_dataset(dict(cfg, test_mode=True))) == 7          # test mode keeps every image
    # EXIF orientation 6 (rotate 90 degrees clockwise to display): decoded upright, like cv2.imread
    rgb = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    im = Image.fromarray(rgb)
    ex = im.getexif()
    ex[0x0112] = 6
    im.save(tmp_path / 'rot.jpg', exif=ex, quality=100, subsampling=0)
    dec = DS.imread_bgr(str(tmp_path / 'rot.jpg'))
    assert dec.shape == (30, 20, 3)
    plain = np.asarray(Image.open(tmp_path / 'rot.jpg').convert('RGB'))          # un-rotated decode of the same file
    assert np.array_equal(dec[:, :, ::-1], np.rot90(plain, k=-1))
    # decode-ahead == synchronous
    a = DS.RetinaFaceSource.__new__(DS.RetinaFaceSource)
    b = DS.RetinaFaceSource.__new__(DS.RetinaFaceSource)
    from yunet_amd.samplers import DistributedGroupSampler
    for o, w in ((a, 0), (b, 3)):
        o.ds, o.bs, o.rank, o.world, o.seed = ds, 2, 0, 1, 5
        o.sampler = DistributedGroupSampler(ds, 2, 1, 0, seed=5)
        o.iters_per_epoch, o._perm_epoch, o._perm = len(o.sampler) // 2, None, None
        o.workers, o._pool, o._ahead = w, None, {}
    for it in (0, 1, 2, 3, 4, 7):                                                 # crosses an epoch, then jumps
        sa, sb = a._decoded(it), b._decoded(it)
        assert [s['filename'] for s in sa] == [s['filename'] for s in sb]
        assert all(np.array_equal(x['img'], y['img']) for x, y in zip(sa, sb))


@pytest.mark.skipif(not os.path.exists(LABELS), reason='reference tree not present')
@pytest.mark.parametrize('seed', range(12))
def test_random_label_files_match_reference(tmp_path, seed):
    """Random labelv2 training files -- every row kind (15 landmark values with visible / occluded / missing (-1)
    points, optional score, single ignore flag), tiny boxes against a random min_size, empty images, repeated image
    headers -- parsed by this reader and by the UNMODIFIED reference class: same images, same arrays."""
    import yunet_amd.datasets as DS
    rng = np.random.default_rng(100 + seed)
    lines, names = [], []
    for i in range(int(rng.integers(3, 25))):
        name = f'{int(rng.integers(0, 60))}--Ev/{int(rng.integers(0, 12))}.jpg' if rng.random() < 0.9 or not names else str(rng.choice(names))
        names.append(name)
        lines.append(f'# {name} {int(rng.integers(20, 1500))} {int(rng.integers(20, 1500))}')
        for _ in range(int(rng.integers(0, 7))):
            x1, y1 = rng.uniform(0, 900, 2)
            w, h = rng.uniform(1, 120, 2) if rng.random() < 0.7 else rng.uniform(1, 12, 2)
            box = '%.2f %.2f %.2f %.2f' % (x1, y1, x1 + w, y1 + h) if rng.random() < 0.5 else '%d %d %d %d' % (x1, y1, x1 + w, y1 + h)
            kind = rng.random()
            if kind < 0.6:
                pts = []
                missing_all = rng.random() < 0.2
                for _k in range(5):
                    if missing_all or rng.random() < 0.15:
                        pts += ['-1', '-1', '-1']
                    else:
                        pts += ['%.3f' % rng.uniform(0, 900), '%.3f' % rng.uniform(0, 900), str(rng.choice(['0.0', '1.0', '0', '1']))]
                row = box + ' ' + ' '.join(pts) + (' %.2f' % rng.random() if rng.random() < 0.7 else '')
            else:
                row = box + ' ' + str(int(rng.integers(0, 2)))
            lines.append(row)
    f = tmp_path / 'l.txt'
    f.write_text('\n'.join(lines) + '\n')
    min_size = None if rng.random() < 0.3 else int(rng.integers(2, 20))
    mine = DS.load_labelv2(str(f), min_size=min_size)
    ref = _ref_instance(str(f), min_size, False)
    assert [(it['filename'], it['width'], it['height']) for it in mine] == \
        [(it['filename'], it['width'], it['height']) for it in ref.data_infos]
    for i in range(len(mine)):
        x, y = DS.ann_info(mine[i]), ref.get_ann_info(i)
        assert set(x) == set(y)
        for k in x:
            assert x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and np.array_equal(x[k], y[k]), (i, k)
