import os, sys, torch
sys.path.insert(0, '/root/repo')
os.environ['EGN_WGW_ABL'] = '32'
from egonet_amd import _lib
L = _lib.lib()
n, cin, cout, h, w = 32, 48, 48, 64, 64
x = torch.randn(n, h, w, cin, device='cuda'); dy = torch.randn(n, h, w, cout, device='cuda')
need = L.egn_conv2d_wgrad_ws_bytes(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1)
ws = torch.zeros(need // 4, device='cuda'); dw = torch.empty(cout, cin, 3, 3, device='cuda')
for _ in range(3):
    _lib.check(L.egn_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, _lib.ptr(ws), need, _lib.current_stream()))
torch.cuda.synchronize()
import numpy as np
t = ws.cpu().numpy().view(np.uint64)[:256 * 32].reshape(256, 32)
nt = int(t[0, 0]); print('ntk', nt)
d = t[:, 1:1 + nt].astype(np.int64)
d = d - d[:, :1]
names = ['start', 'st0 top', 'st0 end', 'st1 top', 'st1 end', 'st2 top', 'st2 end', 'st3 top', 'st3 end', 'epi start', 'stages consumed', 'parked+bar', 'end']
for i in range(nt):
    print('%-10s median %7d  min %7d  max %7d cycles' % (names[i] if i < len(names) else i, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
print('block start spread (cycles):', int(t[:, 1].max() - t[:, 1].min()), ' end spread:', int(t[:, nt].max() - t[:, nt].min()), ' total span', int(t[:, nt].max() - t[:, 1].min()))
