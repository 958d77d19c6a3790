import sys, torch
sys.path.insert(0, '/root/repo')
from egonet_amd import tuner, _lib
dev = torch.device('cuda:0')
L = _lib.lib()
for name, args in [
    ('1x1 K=1024', (4096, 1, 1, 1024, 1024, 1024, 1024, 1, 1, 1, 0, False, False)),
    ('1x4 taps C=256', (4096, 1, 4, 256, 256, 1024, 1024, 1, 4, 1, 0, False, False)),
    ('1x8 taps C=128', (4096, 1, 8, 128, 128, 1024, 1024, 1, 8, 1, 0, False, False)),
    ('2x2 taps C=256', (4096, 2, 2, 256, 256, 1024, 1024, 2, 2, 1, 0, False, False)),
    ('1x1 K=1024 -> 96', (4096, 1, 1, 1024, 1024, 96, 96, 1, 1, 1, 0, False, False)),
]:
    cfg, times = tuner.tune(dev, args)
    best = sorted(times.items(), key=lambda kv: kv[1])[:4]
    fl = 2.0 * 4096 * 1024 * args[5]
    print(name, 'best', [(c, round(t * 1e3, 1)) for c, t in best], 'us ->', round(fl / (best[0][1] * 1e-3) * 1e-12, 1) if best else None, 'TFLOP/s', flush=True)
