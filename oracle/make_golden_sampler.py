"""TEST INFRASTRUCTURE ONLY.  Writes tests/golden/group_sampler.npz: epoch orders produced by the UNMODIFIED
reference samplers (/root/reference/mmdet/datasets/samplers/group_sampler.py: GroupSampler,
DistributedGroupSampler), executed under the arithmetic-free mmcv stub (its one mmcv import is get_dist_info).

    python oracle/make_golden_sampler.py

Cases: random aspect-ratio flags (one group empty in one case, sizes that are not multiples of the batch),
several (samples_per_gpu, world size, seed, epoch) settings; for GroupSampler the global numpy generator is seeded
right before iterating, which is the only way its order is reproducible at all.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stub  # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden', 'group_sampler.npz')

CASES = [   # (n, p(flag = 1), samples_per_gpu, world, seed, epochs)
    (103, 0.7, 8, 2, 0, (0, 1, 5)),
    (64, 0.5, 16, 4, 3, (0, 2)),
    (37, 1.0, 4, 1, 11, (0, 1)),        # group 0 empty
    (10, 0.4, 16, 2, 7, (0,)),          # groups smaller than one batch: cyclic repetition
    (257, 0.65, 32, 8, 1, (0, 9)),
]


def reference_module():
    ref_stub._install_mmcv_stub()
    path = os.path.join(ref_stub.REF_ROOT, 'mmdet', 'datasets', 'samplers', 'group_sampler.py')
    spec = importlib.util.spec_from_file_location('ref_group_sampler', path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class FlagOnly:
    def __init__(self, flag):
        self.flag = flag

    def __len__(self):
        return len(self.flag)


def flags_of(case):
    n, p, *_ = case
    rng = np.random.default_rng(1000 + n)
    return (rng.random(n) < p).astype(np.uint8)


def main():
    ref = reference_module()
    out = {}
    for ci, case in enumerate(CASES):
        n, p, spg, world, seed, epochs = case
        flag = flags_of(case)
        out[f'c{ci}_flag'] = flag
        out[f'c{ci}_cfg'] = np.array([spg, world, seed], np.int64)
        out[f'c{ci}_epochs'] = np.array(epochs, np.int64)
        for ep in epochs:
            for r in range(world):
                s = ref.DistributedGroupSampler(FlagOnly(flag), spg, world, r, seed=seed)
                s.set_epoch(ep)
                idx = np.array(list(iter(s)), np.int64)
                assert len(idx) == len(s)
                out[f'c{ci}_e{ep}_r{r}'] = idx
        np.random.seed(seed + 17)
        g = ref.GroupSampler(FlagOnly(flag), spg)
        out[f'c{ci}_group'] = np.array(list(iter(g)), np.int64)
        assert len(out[f'c{ci}_group']) == len(g)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, {k: v.shape for k, v in list(out.items())[:6]})


if __name__ == '__main__':
    main()
