import sys, numpy as np, torch
sys.path.insert(0, '.')
from monopsr_amd import _lib
from monopsr_amd.core import device_net as dn
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B,H,Wd,C,N = 2,12,12,64,64
rng = np.random.default_rng(0)
x = rng.standard_normal((B,H,Wd,C)).astype(np.float32)
w = (rng.standard_normal((N,C))/8).astype(np.float32)
ref = x.reshape(-1,C).astype(np.float64) @ w.T.astype(np.float64)
_lib.lib().mpsr_debug_set_conv_tile(tile)
got = dn.conv2d(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()).cpu().numpy().reshape(-1,N)
err = np.abs(got-ref)
bad = err > 1e-4
print('bad frac', bad.mean())
rows = np.nonzero(bad.any(1))[0]; cols = np.nonzero(bad.any(0))[0]
print('bad rows', rows[:50], len(rows)); print('bad cols', cols[:70], len(cols))
# check if got row r equals ref of some other row
if len(rows):
    r = rows[0]
    d = np.abs(ref - got[r][None,:]).sum(1); print('row', r, 'closest ref row', d.argmin(), d.min())
    c = cols[0]
    d = np.abs(ref - got[:,c][:,None]).sum(0); print('col', c, 'closest ref col', d.argmin(), d.min())
