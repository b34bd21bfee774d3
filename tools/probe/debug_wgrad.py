"""Where does a cg_wgrad result differ from the oracle? (development tool)
  python tools/probe/debug_wgrad.py nB L Ci Co seg use_shift [classic]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle as O
import hip_utils as H
from calciumgan_amd import geometry as geo, nets

a = [int(v) for v in sys.argv[1:]]
nB, L, Ci, Co, seg, use_shift = a[:6]
classic = a[6] if len(a) > 6 else 0
mode = a[7] if len(a) > 7 else 0
k = 24
rng = np.random.RandomState(7)
if mode == 0:
  x = H.int_tensor(rng, (nB, L, Ci), -2, 2)
  dy = H.int_tensor(rng, (nB, L // 2, Co), -2, 2)
else:
  # x = one-hot probes: x[b, l, c] = 1 only at (l0, c0); dy = 1 at (u0, n0)
  x = torch.zeros(nB, L, Ci); dy = torch.zeros(nB, L // 2, Co)
  x[0, 20, 5] = 1.0
  dy[0, 8, 3] = 1.0
nseg = (nB + seg - 1) // seg
shifts = rng.randint(-2, 3, size=nseg).astype(np.int32)
if not use_shift:
  shifts[:] = 0
def shuffle_batch(x, shifts, seg):
  out = []
  for b in range(x.shape[0]):
    out.append(O.phase_shuffle(x[b:b + 1], int(shifts[b // seg])) if hasattr(O, 'phase_shuffle') else x[b:b+1])
  return torch.cat(out)
W = torch.zeros(k, Ci, Co, requires_grad=True)
xs = x
if use_shift:
  import test_hip_kernels as T
  xs = T._shuffle_batch(x, shifts, seg)
(O.conv1d_same(xs, W, None, 2) * dy).sum().backward()
cip, cop = geo.pitch(Ci), geo.pitch(Co)
dw = torch.zeros(k, Ci, Co, dtype=torch.float32, device=H.DEV)
sh = torch.tensor(shifts, device=H.DEV)
d = nets._wgrad_desc(H.to_pitch(x, cip), H.to_pitch(dy, cop), dw, nB, L, cip,
                     L // 2, cop, k, 2, -geo.same_padding_left(k, 2), Ci, Co,
                     shifts=sh if use_shift else None, seg_size=seg, slot=None)
d.classic_staging = classic
H.run_wgrad(d)
H.sync()
got = dw.cpu().numpy(); ref = W.grad.numpy()
bad = got != ref
print('mismatch %d / %d' % (bad.sum(), bad.size))
print('bad per tap :', bad.sum((1, 2)))
print('bad per cx//16:', bad.reshape(k, -1, Co).sum((0, 2))[:Ci].reshape(-1, 2 if Ci % 2 == 0 else 1).sum(1)[:16] if False else bad.sum((0, 2))[:64])
print('bad per cg  :', bad.sum((0, 1))[:64])
if mode:
  print('ref nonzero:', np.argwhere(ref != 0)[:10].tolist())
  print('got nonzero:', np.argwhere(got != 0)[:20].tolist(), got[got != 0][:20])
