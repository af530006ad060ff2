import sys, time, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
sys.path.insert(0, os.path.join(sys.path[0], 'oracle'))
import torch, yunet_oracle as O, yunet_amd.synthetic as S
for thr in (8, 16, 32):
    torch.set_num_threads(thr)
    for bs in (16, 32, 64):
        arch = O.yunet_arch('n'); sd = O.init_state(arch, 0); opt = O.SGD(lr=1e-5)
        b = S.make_batch(bs, 320, 320, 1234)
        O.train_step(b, sd, arch, opt)
        t0 = time.time(); n = 0
        while time.time() - t0 < 4: O.train_step(b, sd, arch, opt); n += 1
        print(f'threads {thr} bs {bs}: {bs * n / (time.time() - t0):.1f} img/s', flush=True)
