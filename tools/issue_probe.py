import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from egonet_amd import configs, synth
from egonet_amd.model.heatmapModel import hrnet
from egonet_amd.train_hrnet import HRNetTrainStep
cfg = configs.w48_config('coordinates')
net = hrnet.get_pose_net(cfg, is_train=False)
net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
net = net.cuda().train()
tr = HRNetTrainStep(net, lr=1e-3)
g = torch.Generator().manual_seed(100)
x = synth.synth_crops(32, 3, 256, 256, seed=50).cuda()
tgt = torch.rand(32, 33, 64, 64, generator=g).cuda()
jt = (torch.rand(32, 33, 2, generator=g) * 256).cuda()
for _ in range(4):
    tr.step(x, tgt, jt)
torch.cuda.synchronize()
iss, tot = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(x, tgt, jt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    iss.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print('issue ms', ['%.1f' % v for v in iss])
print('total ms', ['%.1f' % v for v in tot])
import egonet_amd._lib as L
print('launches/step', L.lib().egn_launch_count())

# where do the sporadic 2x steps come from?  (a) cyclic GC  (b) the caching allocator asking the driver
import gc
gct = []
def _cb(phase, info):
    if phase == 'start':
        _cb.t = time.perf_counter()
    else:
        gct.append((info['generation'], (time.perf_counter() - _cb.t) * 1e3, info['collected']))
gc.callbacks.append(_cb)
for mode in ('gc on', 'gc frozen'):
    if mode == 'gc frozen':
        gc.collect(); gc.freeze()
    iss = []
    for _ in range(12):
        torch.cuda.synchronize()
        s0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        tr.step(x, tgt, jt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        s1 = torch.cuda.memory_stats()
        iss.append('%.0f/%.0f(seg+%d)' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, s1['segment.all.allocated'] - s0['segment.all.allocated']))
    print(mode, ' '.join(iss))
    print('  gc events (gen, ms, collected):', [(g, round(ms, 1), c) for g, ms, c in gct if ms > 2])
    gct.clear()
