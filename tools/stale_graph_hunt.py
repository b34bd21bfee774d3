"""Root-cause hunt for the round-2 stale-graph corruption:
graph replays of train() -> validate() at another batch size -> eager train()
steps -> replay of the OLD graphs gave garbage penalties in ~40 % of processes.

  python tools/stale_graph_hunt.py <variant> [tag]

variants (one scenario per process; prints one line `HUNT <variant> <ok|BAD> ...`):
  plain      the full scenario, OLD graphs replayed
  noval      no validate() in between
  val8       validate() at the training batch size
  noeager    validate() at another batch size, no eager train() steps
  poison     plain, with every shared workspace filled with NaN before the replay
  guard      plain, but validate()'s state is built inside a guard check: every
             live tensor's checksum is taken before / after the B2 descriptors
             are built (tuning launches) and after validate()
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import oracle as O
from calciumgan_amd import nets
from calciumgan_amd.gan.algorithms import get_algorithm
from calciumgan_amd.gan.models import get_models

variant = sys.argv[1] if len(sys.argv) > 1 else 'plain'
tag = sys.argv[2] if len(sys.argv) > 2 else ''

hp = O.make_hparams(256, 16, 8, m=2)
hp.verbose = 0
gen, dis = get_models(hp, None)
gan = get_algorithm(hp, gen, dis, None)
rng = np.random.RandomState(0)
data = torch.tensor(rng.uniform(0, 1, (64, 256, 16)).astype(np.float32),
                    device='cuda')


def batch(i):
  jj = torch.arange(8 * (i % 8), 8 * (i % 8) + 8, device='cuda')
  return data.index_select(0, jj)


def persistent_tensors():
  """Every long-lived device tensor of the B = 8 state, by name."""
  st = gan._get_state(8)
  out = {}
  for name, p in (('dis', dis.net.params), ('gen', gen.net.params)):
    out[name + '_w'], out[name + '_m'], out[name + '_v'] = p.data, p.m, p.v
  for i, op in enumerate(dis.net.w_fwd + dis.net.w_dgrad):
    out['dpack%d' % i] = op.buf
  for i, op in enumerate(gen.net.w_fwd + gen.net.w_dgrad):
    out['gpack%d' % i] = op.buf
  pl = st['critic']
  out['coef'], out['bias_coef'] = pl.coef, pl.bias_coef
  out['gen_coef'] = st['gen'].coef
  g = st.get('graph')
  if g is not None:
    out['g_real'], out['stage_dev'] = g['real'], g['stage_dev']
  return out


def checksums():
  return {k: (float(t.double().sum()), float(t.double().abs().max()))
          for k, t in persistent_tensors().items()}


def diff(a, b, what):
  bad = [k for k in a if k in b and a[k] != b[k]]
  if bad:
    print('GUARD', what, 'changed:', bad)
  return bad


probe = None
if variant == 'memsetprobe':
  # beside every cg_rownorm of the step: hipMemsetAsync(probe, 0) -> atomics
  # of a known column sum into probe -> copy to a per-call result row.  In the
  # captured step the memset is a memset NODE of the graph.
  import ctypes
  from calciumgan_amd import _lib
  hip = ctypes.CDLL('libamdhip64.so')
  hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t,
                                 ctypes.c_void_p]
  ROWS = 512
  probe = dict(buf=torch.full((8,), 7.0, device='cuda'),
               ones=torch.ones(ROWS, 8, dtype=nets.act_dtype(), device='cuda'),
               res=torch.zeros(5, 8, device='cuda'), n=0)
  orig_call = _lib.call

  def call(name, *a):
    if name == 'cg_rownorm':
      st_ = nets._stream()
      rc = hip.hipMemsetAsync(probe['buf'].data_ptr(), 0, 32, st_)
      assert rc == 0, rc
      orig_call('cg_colsum', nets._p(probe['ones']), nets._p(probe['buf']),
                ROWS, 8, 8, None, st_)
      probe['res'][probe['n'] % 5].copy_(probe['buf'])
      probe['n'] += 1
    return orig_call(name, *a)
  _lib.call = call

for i in range(10):
  gan.train(batch(i))
torch.cuda.synchronize()
guard_hits = []
if variant == 'guard':
  c0 = checksums()
  gan._get_state(6)  # builds + tunes the B2 descriptors
  torch.cuda.synchronize()
  c1 = checksums()
  guard_hits += diff(c0, c1, 'building the B2 state (tuning launches)')
if variant not in ('noval', 'memsetprobe'):
  n = 8 if variant == 'val8' else 6
  v = gan.validate(data[:n])
  torch.cuda.synchronize()
  if variant == 'guard':
    c2 = checksums()
    guard_hits += diff(c1, c2, 'validate()')
if variant != 'noeager':
  gan._use_graph = False
  for i in range(5):
    gan.train(batch(i))
  torch.cuda.synchronize()
  gan._use_graph = True
st = gan._get_state(8)
assert 'graph' in st, 'the old graphs must still be there (hook not honoured)'
if variant == 'poison':
  for t in list(nets._SPLIT_WS.values()) + list(nets._PARTIALS_POOL.values()):
    t.fill_(float('nan'))
  for B2, s2 in gan._state.items():
    if B2 == 8:
      continue
    for t in s2['dws'].act + [x for x in s2['dws'].delta if x is not None]:
      t.fill_(float('nan'))
  torch.cuda.synchronize()
o = gan.train(batch(0))
torch.cuda.synchronize()
vals = [float(o[0]), float(o[1]), float(o[2])]
gp = st['gp'].cpu().numpy()
bad = not np.isfinite(vals).all() or abs(vals[1]) > 1e3 or vals[2] > 1e3
pl = st['critic']
info = dict(
    out=vals, gp=gp.tolist(),
    shifts=pl.shifts.cpu().numpy().reshape(-1).tolist(),
    stage=st['graph']['stage_dev'].cpu().numpy().tolist()[:16],
    d_out_absmax=float(st['dws'].d_out.abs().max()),
    gin_absmax=float(pl.gin.float().abs().max()),
    act_absmax=[float(t.float().abs().max()) for t in st['dws'].act],
    delta_absmax=[float(t.float().abs().max()) for t in st['dws'].delta
                  if t is not None],
    w_absmax=float(dis.net.params.data.abs().max()))
if probe is not None:
  res = probe['res'].cpu().numpy()
  print('MEMSETPROBE expected', ROWS, 'got rows', res[:, 0].tolist(),
        'all_ok', bool((res == ROWS).all()))
print('HUNT', variant, tag, 'BAD' if bad else 'ok', 'guard_hits=%s' % guard_hits,
      info if bad else vals)
