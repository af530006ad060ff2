"""Epoch sample order: yunet_amd.samplers against orders produced by the UNMODIFIED reference samplers
(tests/golden/group_sampler.npz, written by oracle/make_golden_sampler.py; re-checked live where /root/reference
exists), the properties a distributed sampler must have, and RetinaFaceSource's use of it."""
import os
import sys

import numpy as np
import pytest

from yunet_amd import samplers as S

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'group_sampler.npz'))
NCASES = len([k for k in G.files if k.endswith('_flag')])


class FlagOnly:
    def __init__(self, flag):
        self.flag = flag

    def __len__(self):
        return len(self.flag)


@pytest.mark.parametrize('ci', range(NCASES))
def test_orders_equal_the_reference_samplers(ci):
    flag = G[f'c{ci}_flag']
    spg, world, seed = (int(v) for v in G[f'c{ci}_cfg'])
    for ep in G[f'c{ci}_epochs']:
        whole = []
        for r in range(world):
            s = S.DistributedGroupSampler(FlagOnly(flag), spg, world, r, seed=seed)
            s.set_epoch(int(ep))
            got = np.array(list(iter(s)), np.int64)
            assert len(s) == len(got) and np.array_equal(got, G[f'c{ci}_e{int(ep)}_r{r}']), (ci, ep, r)
            whole.append(got)
        whole = np.concatenate(whole)
        # every sample of the data set is drawn, the ranks' parts are equally long, and every batch of
        # samples_per_gpu images stays inside one aspect-ratio group
        assert set(whole.tolist()) == set(np.flatnonzero(np.ones_like(flag)).tolist())
        assert all(len(set(flag[b].tolist())) == 1 for b in whole.reshape(-1, spg))
    np.random.seed(seed + 17)
    g = S.GroupSampler(FlagOnly(flag), spg)
    got = np.array(list(iter(g)), np.int64)
    assert len(g) == len(got) and np.array_equal(got, G[f'c{ci}_group'])


def test_live_reference_when_present():
    sys.path.insert(0, os.path.join(HERE, '..', 'oracle'))
    import ref_stub
    if not ref_stub.available():
        pytest.skip('reference tree not present')
    import make_golden_sampler as M
    ref = M.reference_module()
    rng = np.random.default_rng(5)
    for _ in range(6):
        n, spg, world = int(rng.integers(5, 300)), int(rng.choice([1, 3, 8, 16])), int(rng.choice([1, 2, 4, 8]))
        flag = (rng.random(n) < rng.random()).astype(np.uint8)
        seed, ep = int(rng.integers(0, 100)), int(rng.integers(0, 50))
        for r in range(world):
            a = ref.DistributedGroupSampler(FlagOnly(flag), spg, world, r, seed=seed)
            b = S.DistributedGroupSampler(FlagOnly(flag), spg, world, r, seed=seed)
            a.set_epoch(ep)
            b.set_epoch(ep)
            assert list(iter(a)) == list(iter(b)) and len(a) == len(b)
        np.random.seed(seed)
        x = list(iter(ref.GroupSampler(FlagOnly(flag), spg)))
        np.random.seed(seed)
        assert x == list(iter(S.GroupSampler(FlagOnly(flag), spg)))


def test_retinaface_source_follows_the_distributed_group_sampler(tmp_path):
    """RetinaFaceSource._indices(it): iteration it of rank r is the it-th batch of DistributedGroupSampler(dataset,
    samples_per_gpu, world, r, seed) in epoch it // iters_per_epoch -- the order the reference's dist_train feeds."""
    from yunet_amd.datasets import RetinaFaceDataset, RetinaFaceSource
    rng = np.random.default_rng(2)
    lines = []
    for i in range(45):
        w, h = (int(v) for v in rng.integers(40, 400, 2))
        lines += [f'# ev/{i}.jpg {w} {h}', '3 4 30 40 ' + ' '.join(['-1'] * 15)]
    (tmp_path / 'l.txt').write_text('\n'.join(lines) + '\n')
    ds = RetinaFaceDataset(str(tmp_path / 'l.txt'), img_prefix=str(tmp_path))
    import yunet_amd
    pipe = yunet_amd.Config.fromfile(os.path.join(HERE, '..', 'configs', 'yunet_n.py')).data.train.pipeline
    world, bs = 2, 4
    srcs = [RetinaFaceSource(ds, pipe, samples_per_gpu=bs, rank=r, world=world, seed=9) for r in range(world)]
    ipe = srcs[0].iters_per_epoch
    ref = S.DistributedGroupSampler(ds, bs, world, 0, seed=9)
    assert ipe == len(ref) // bs and ipe * bs * world >= len(ds)
    for epoch in (0, 1, 3):
        seen = []
        for r in range(world):
            smp = S.DistributedGroupSampler(ds, bs, world, r, seed=9)
            smp.set_epoch(epoch)
            want = np.array(list(iter(smp))).reshape(-1, bs)
            for k in range(ipe):
                got = srcs[r]._indices(epoch * ipe + k)
                assert got == want[k].tolist()
                assert len({int(ds.flag[i]) for i in got}) == 1
                seen += got
        assert set(seen) == set(range(len(ds)))
