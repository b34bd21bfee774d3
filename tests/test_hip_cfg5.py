"""BASELINE.json configs[4] at its STATED size (seq_len 8192, 512 neurons, batch
256, mixed_float16) on the GPU, where the CPU oracle cannot be the checker:

* the three MFMA contractions of the critic's first layer over its whole
  3 x 256-sample batch -- a 6.4 GB fp16 input, a 12.9 GB f32 input gradient,
  byte offsets past 2^32 -- are each other's adjoints, EXACTLY, on
  small-integer data (the size-independent property of test_hip_fullsize.py;
  the commit "no 2 GiB limit on the batch" is what this guards);
* one full-batch train() sequence in fp16 with dynamic loss scaling: finite,
  sane losses, the loss scale unchanged or lowered, the step replayed as a
  hipGraph.
"""
import numpy as np
import pytest
import torch

import oracle as O
from calciumgan_amd import _lib
from calciumgan_amd import geometry as geo
from calciumgan_amd import nets

pytestmark = pytest.mark.gpu

import hip_utils as H  # noqa: E402


@pytest.fixture(autouse=True)
def _fp16_library():
  _lib.use('f16')
  yield
  _lib.use('bf16')
  torch.cuda.empty_cache()


def _int_pitched(gen, nB, L, C, cp, dtype, chunk=32):
  """(nB, L, cp) activations of small integers in [-2, 2], built chunk-wise
  (an f32 staging copy of the whole tensor would be 13 GB)."""
  out = torch.zeros(nB, L, cp, dtype=dtype, device=H.DEV)
  for b in range(0, nB, chunk):
    n = min(chunk, nB - b)
    out[b:b + n, :, :C] = torch.randint(-2, 3, (n, L, C), generator=gen,
                                        device=H.DEV, dtype=torch.int8).to(dtype)
  return out


def _dot(a_fn, b_fn, nB, chunk=32):
  """sum_b <a[b], b[b]> in f64, `chunk` samples at a time."""
  s = 0.0
  for b in range(0, nB, chunk):
    n = min(chunk, nB - b)
    s += float((a_fn(b, n).double() * b_fn(b, n).double()).sum())
  return s


@pytest.mark.parametrize('nB,L,Ci,Co,k,seg', [(768, 8192, 512, 64, 24, 256)])
def test_cfg5_first_layer_contractions_are_adjoint_at_full_batch(nB, L, Ci, Co, k,
                                                                  seg):
  dt = nets.act_dtype()
  assert dt == torch.float16
  gen = torch.Generator(device=H.DEV)
  gen.manual_seed(4321)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  Lo = L // 2
  xd = _int_pitched(gen, nB, L, Ci, cip, dt)
  gd = _int_pitched(gen, nB, Lo, Co, cop, dt)
  assert xd.numel() * 2 > 2**32          # byte offsets beyond 32 bits
  W = (torch.randint(-2, 3, (k, Ci, Co), generator=gen, device=H.DEV,
                     dtype=torch.int32).float() * 0.5)
  shifts = np.array([7, -10, 3], np.int32)
  sh = torch.tensor(shifts, device=H.DEV)

  ck = nets._ck_for(cip, 2, k, Lo)
  op = H.pack(W, [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k, parity_major=True)
  y = torch.zeros(nB, Lo, cop, dtype=torch.float32, device=H.DEV)
  d = H.conv_desc(xd, op.buf, y, nB, L, cip, k, 2, -pl, Lo, Co, Lo, cop, ck,
                  shifts=sh, seg_size=seg, out_f32=True,
                  w_parity_major=op.parity_major, w_narrow_last=op.narrow_last)
  H.run_conv(d)

  phases = nets._transpose_phases(k, pl)
  offs = [o for _, o in phases]
  ckd = nets._ck_for(cop, 1, k // 2, Lo)
  opd = H.pack(W, [(t0, -2, Ci * Co, 1, Co) for t0, _ in phases], Co, Ci, cop,
               ckd, k // 2)
  dxs = torch.zeros(nB, L, cip, dtype=torch.float32, device=H.DEV)
  assert dxs.numel() * 4 > 2**33
  dd = H.conv_desc(gd, opd.buf, dxs, nB, Lo, cop, k // 2, 1, offs[0], Lo, Ci, L,
                   cip, ckd, y_stride=2, y_off=0, out_f32=True, nphase=2,
                   w_phase_stride=opd.elems, off_phase_step=offs[1] - offs[0],
                   yoff_phase_step=1)
  H.run_conv(dd)

  dw = torch.zeros(k, Ci, Co, dtype=torch.float32, device=H.DEV)
  dwd = nets._wgrad_desc(xd, gd, dw, nB, L, cip, Lo, cop, k, 2, -pl, Ci, Co,
                         shifts=sh, seg_size=seg, slot=0)
  H.run_wgrad(dwd)
  H.sync()

  def x_shuffled(b, n):
    idx = torch.from_numpy(O.phase_shuffle_index(L, int(shifts[b // seg]))).to(
        H.DEV)
    return xd[b:b + n, :, :Ci].index_select(1, idx)

  ip_y = _dot(lambda b, n: y[b:b + n, :, :Co], lambda b, n: gd[b:b + n, :, :Co],
              nB)
  ip_x = _dot(x_shuffled, lambda b, n: dxs[b:b + n, :, :Ci], nB)
  ip_w = float((W.double() * dw.double()).sum())
  assert ip_y != 0.0
  assert ip_y == ip_x, (ip_y, ip_x)
  assert ip_y == ip_w, (ip_y, ip_w)
  # the LAST sample is as right as the first (a wrapped offset would land in an
  # earlier sample): its rows against a plain torch convolution
  b = nB - 1
  xs = x_shuffled(b, 1)[0].float().t()[None]                  # (1, Ci, L)
  ref = torch.nn.functional.conv1d(xs, W.permute(2, 1, 0).contiguous(),
                                   stride=2, padding=pl)[0].t()  # (Lo, Co)
  assert torch.equal(ref, y[b, :, :Co])


def test_cfg5_full_batch_fp16_train_steps():
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  import bench
  saved = nets._AUTOTUNE
  nets._AUTOTUNE = False   # static tiles: tuning at these sizes takes minutes
  try:
    hp = bench.make_hparams(8192, 512, 64, 10, True)
    hp.verbose = 0
    gen, dis = get_models(hp, None)
    gan = get_algorithm(hp, gen, dis, None)
    assert gan.precision == 'f16'
    g = torch.Generator(device=gan.device)
    g.manual_seed(1234)
    real = torch.rand(256, 8192, 512, generator=g, device=gan.device)
    assert real.numel() * 4 >= 2**32
    outs = []
    for _ in range(4):               # 2 eager calls, then hipGraph replays
      o = gan.train(real)
      outs.append([float(o[0]), float(o[1]), float(o[2])] +
                  [float(v) for v in o[3].values()])
    torch.cuda.synchronize()
    outs = np.array(outs)
    assert np.isfinite(outs).all(), outs
    # Sane, not garbage: the first step's penalty (mean over its n_critic
    # updates of a fresh critic) is O(1) -- measured 0.44 --, and at these sizes
    # the critic's gradient norm then grows by a few per update (6.5, 10 by the
    # fourth step, the same dynamics as the f32 oracle at cfg2:
    # test_hip_cfg2.py); a wrapped offset or an fp16 overflow gives inf / nan /
    # 1e20.  Sigmoid outputs against U[0, 1) data: metric errors below 1.
    assert 0.05 < outs[0, 2] < 3.0, outs[:, 2]
    assert (outs[:, 2] > 0).all() and (outs[:, 2] < 1e3).all(), outs[:, 2]
    assert (np.abs(outs[:, :2]) < 1e4).all(), outs[:, :2]
    assert (outs[:, 3:] >= 0).all() and (outs[:, 3:] < 1.0).all()
    for opt in (gan.dis_optimizer, gan.gen_optimizer):
      s = float(opt.loss_scale_state[0])
      assert 1.0 <= s <= nets.LOSS_SCALE_INIT, s
    # every update was either applied or skipped-and-halved, never lost
    applied = gan.dis_optimizer.iterations
    halvings = int(round(np.log2(nets.LOSS_SCALE_INIT /
                                 float(gan.dis_optimizer.loss_scale_state[0]))))
    assert applied + halvings == 4 * gan.n_critic, (applied, halvings)
    assert gan._get_state(256).get('graph') is not None
  finally:
    nets._AUTOTUNE = saved
