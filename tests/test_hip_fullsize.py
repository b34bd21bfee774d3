"""GPU property tests at BASELINE cfg2's full layer sizes (3 x 128 samples,
2048 steps, 102 channels), where the CPU oracle is too slow to be the checker.

The three MFMA contractions of one Conv1D layer are each other's adjoints:

    < conv(S x; W), g >  ==  < S x, dgrad(g; W) >  ==  < W, wgrad(S x, g) >

(S = the fused per-segment PhaseShuffle gather).  On small-integer data every
product and every partial sum is an integer below 2^24, so all three kernels
are exact in f32 whatever their summation order (tile shapes, K' splits, f32
atomics) and the three inner products, accumulated in f64, must be EQUAL --
bit for bit, at full size, through the C ABI.  A second property pins the
forward kernel itself: linearity, conv(x1 + x2) == conv(x1) + conv(x2).
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

BF16 = torch.bfloat16

# (nB, L, Ci, Co, k, seg_size): discriminator layers 1 and 3 of cfg2 with the
# critic's real | fake | interpolate batch
LAYERS = [(384, 2048, 102, 64, 24, 128), (384, 512, 128, 192, 24, 128)]


def _rand_int(gen, shape, lo, hi, scale=1.0):
  return (torch.randint(lo, hi + 1, shape, generator=gen, device=H.DEV,
                        dtype=torch.int32).float() * scale)


def _pitched(x, cp):
  out = torch.zeros(x.shape[0], x.shape[1], cp, dtype=BF16, device=H.DEV)
  out[:, :, :x.shape[2]] = x.to(BF16)
  return out


def _shuffled(x, shifts, seg):
  out = torch.empty_like(x)
  for s, sh in enumerate(shifts):
    idx = torch.from_numpy(O.phase_shuffle_index(x.shape[1], int(sh))).to(H.DEV)
    out[s * seg:(s + 1) * seg] = x[s * seg:(s + 1) * seg].index_select(1, idx)
  return out


@pytest.mark.parametrize('nB,L,Ci,Co,k,seg', LAYERS)
def test_conv_dgrad_wgrad_are_adjoint_at_full_size(nB, L, Ci, Co, k, seg):
  gen = torch.Generator(device=H.DEV)
  gen.manual_seed(1234)
  x = _rand_int(gen, (nB, L, Ci), -2, 2)
  g = _rand_int(gen, (nB, L // 2, Co), -2, 2)
  W = _rand_int(gen, (k, Ci, Co), -2, 2, 0.5)
  shifts = np.array([7, -10, 3], np.int32)[:nB // seg]
  sh = torch.tensor(shifts, device=H.DEV)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  Lo = L // 2
  xd, gd = _pitched(x, cip), _pitched(g, cop)

  # y = conv(S x; W), f32
  ck = nets._ck_for(cip, 2, k, Lo)
  op = H.pack(W, [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k)
  y = torch.zeros(nB, Lo, cop, dtype=torch.float32, device=H.DEV)
  d = H.conv_desc(xd, op.buf, y, nB, L, cip, k, 2, -pl, Lo, Co, Lo, cop, ck,
                  shifts=sh, seg_size=seg, out_f32=True)
  H.run_conv(d)

  # dxs = dgrad(g; W): gradient w.r.t. the shuffled input, f32
  phases = nets._transpose_phases(k, pl)
  offs = [o for _, o in phases]
  ckd = nets._ck_for(cop, 1, k // 2, Lo)
  opd = H.pack(W, [(t0, -2, Ci * Co, 1, Co) for t0, _ in phases], Co, Ci, cop,
               ckd, k // 2)
  dxs = torch.zeros(nB, L, cip, dtype=torch.float32, device=H.DEV)
  dd = H.conv_desc(gd, opd.buf, dxs, nB, Lo, cop, k // 2, 1, offs[0], Lo, Ci, L,
                   cip, ckd, y_stride=2, y_off=0, out_f32=True, nphase=2,
                   w_phase_stride=opd.elems, off_phase_step=offs[1] - offs[0],
                   yoff_phase_step=1)
  H.run_conv(dd)

  # dW = wgrad(S x, g): K'-split partial sums + reducing launch (the product
  # path's form)
  dw = torch.zeros(k, Ci, Co, dtype=torch.float32, device=H.DEV)
  dwd = nets._wgrad_desc(xd, gd, dw, nB, L, cip, Lo, cop, k, 2, -pl, Ci, Co,
                         shifts=sh, seg_size=seg, slot=0)
  H.run_wgrad(dwd)
  H.sync()

  xs = _shuffled(x, shifts, seg)
  ip_y = float((y[:, :, :Co].double() * g.double()).sum())
  ip_x = float((xs.double() * dxs[:, :, :Ci].double()).sum())
  ip_w = float((W.double() * dw.double()).sum())
  assert ip_y != 0.0
  assert ip_y == ip_x, (ip_y, ip_x)
  assert ip_y == ip_w, (ip_y, ip_w)
  # channel padding of the f32 outputs stays exactly zero
  assert float(y[:, :, Co:].abs().max()) == 0.0 if cop > Co else True
  assert float(dxs[:, :, Ci:].abs().max()) == 0.0 if cip > Ci else True


@pytest.mark.parametrize('nB,L,Ci,Co,k,seg', LAYERS[:1])
def test_conv_forward_is_linear_at_full_size(nB, L, Ci, Co, k, seg):
  gen = torch.Generator(device=H.DEV)
  gen.manual_seed(99)
  x1 = _rand_int(gen, (nB, L, Ci), -2, 2)
  x2 = _rand_int(gen, (nB, L, Ci), -2, 2)
  W = _rand_int(gen, (k, Ci, Co), -2, 2, 0.5)
  b = _rand_int(gen, (Co,), -3, 3)
  shifts = torch.tensor(np.array([-4, 9, 0], np.int32), device=H.DEV)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  Lo = L // 2
  ck = nets._ck_for(cip, 2, k, Lo)
  op = H.pack(W, [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k)
  outs = []
  for xin, bias in ((x1, None), (x2, None), (x1 + x2, None), (x1, b)):
    y = torch.zeros(nB, Lo, cop, dtype=torch.float32, device=H.DEV)
    d = H.conv_desc(_pitched(xin, cip), op.buf, y, nB, L, cip, k, 2, -pl, Lo,
                    Co, Lo, cop, ck, shifts=shifts, seg_size=seg, bias=bias,
                    out_f32=True)
    H.run_conv(d)
    outs.append(y)
  H.sync()
  assert torch.equal(outs[0] + outs[1], outs[2])
  assert torch.equal(outs[0][:, :, :Co] + b, outs[3][:, :, :Co])


def test_adam_at_full_parameter_count():
  """Keras Adam over the critic's whole flat parameter buffer (4 110 273
  values), three consecutive updates, against the oracle in f64."""
  import math
  n = 4110273
  gen = torch.Generator()
  gen.manual_seed(5)
  p0 = torch.randn(n, generator=gen) * 0.05
  pr, mr, vr = p0.double(), torch.zeros(n).double(), torch.zeros(n).double()
  pd = p0.clone().to(H.DEV)
  md = torch.zeros(n, device=H.DEV)
  vd = torch.zeros(n, device=H.DEV)
  for t in (1, 2, 3):
    g = torch.randn(n, generator=gen) * (0.1 / t)
    O.keras_adam(pr, g.double() * 0.125, mr, vr, t, 1e-4)
    lr_t = 1e-4 * math.sqrt(1 - 0.999**t) / (1 - 0.9**t)
    gd = g.to(H.DEV)
    _lib.call('cg_adam', H.p(pd), H.p(gd), H.p(md), H.p(vd), n, lr_t, 0.9,
              0.999, 1e-7, 0.125, None, H.stream())
  H.sync()
  # f32 state vs f64 oracle after three steps
  np.testing.assert_allclose(pd.cpu().numpy(), pr.float().numpy(), rtol=2e-6,
                             atol=1e-7)
  np.testing.assert_allclose(md.cpu().numpy(), mr.float().numpy(), rtol=2e-6,
                             atol=1e-9)
  # (1 - beta_2 in f32, as Keras computes it in the variable dtype, is
  # 0.99998713e-3: 1.3e-5 away from the f64 oracle's 1e-3)
  np.testing.assert_allclose(vd.cpu().numpy(), vr.float().numpy(), rtol=3e-5,
                             atol=1e-12)


def test_signal_metrics_and_layernorm_at_full_size():
  B, L, C = 128, 2048, 102
  gen = torch.Generator()
  gen.manual_seed(6)
  real = torch.rand(B, L, C, generator=gen)
  cp = geo.pitch(C)
  fake = torch.zeros(B, L, cp)
  fake[:, :, :C] = torch.rand(B, L, C, generator=gen)
  real_d, fake_d = real.to(H.DEV), fake.to(H.DEV)
  ref = O.signal_metrics(real, fake[:, :, :C], -0.5, 2.5, True)
  exp = [ref['signals_metrics/' + k].item() for k in ('min', 'max', 'mean',
                                                      'std')]
  for ws in (None, H.reduce_ws()):
    (buf,) = H.out_buffers(ws, 4)
    _lib.call('cg_signal_metrics', H.p(real_d), H.p(fake_d), H.p(buf), B * L, C,
              C, cp, -0.5, 2.5, H.p(ws), H.stream())
    H.sync()
    got = buf.cpu().numpy() / (B * L if ws is None else 1)
    np.testing.assert_allclose(got, exp, rtol=2e-4)
  # LayerNorm + LeakyReLU over all 262 144 rows: rows of the normalised
  # pre-activation have mean beta-weighted 0 / variance 1 (gamma = 1, beta = 0)
  rows = B * L
  y = (torch.randn(rows, C, generator=gen) * 3 + 1.5)
  yd = torch.zeros(rows, cp, dtype=BF16, device=H.DEV)
  yd[:, :C] = y.to(BF16)
  gamma = torch.ones(C, device=H.DEV)
  beta = torch.zeros(C, device=H.DEV)
  h = torch.zeros(rows, cp, dtype=BF16, device=H.DEV)
  mean = torch.zeros(rows, device=H.DEV)
  rstd = torch.zeros(rows, device=H.DEV)
  _lib.call('cg_ln_lrelu_fwd', H.p(yd), H.p(gamma), H.p(beta), H.p(h),
            H.p(mean), H.p(rstd), rows, C, cp, 1e-3, 1.0, H.stream())  # alpha 1
  H.sync()
  yq = yd[:, :C].float()
  np.testing.assert_allclose(mean.cpu().numpy(), yq.mean(1).cpu().numpy(),
                             rtol=1e-5, atol=1e-5)
  var = yq.var(1, unbiased=False)
  np.testing.assert_allclose(rstd.cpu().numpy(),
                             torch.rsqrt(var + 1e-3).cpu().numpy(), rtol=1e-5)
  xh = h[:, :C].float()
  assert float(xh.mean(1).abs().max()) < 2e-2      # bf16 output rounding
  assert abs(float(xh.var(1, unbiased=False).mean()) - 1.0) < 1e-2
  assert float(h[:, C:].float().abs().max()) == 0.0


# -- the specialised epilogues of the 32-row wave tiles at the benchmark's sizes --
# (nB, L, Ci, Co): the critic's five layers over its 3 x 128 batch
_CRITIC = [(384, 2048, 102, 64), (384, 1024, 64, 128), (384, 512, 128, 192),
           (384, 256, 192, 256), (384, 128, 256, 320)]


def _swp_vs_classic(make_desc, outputs, tile):
  """The same descriptor on the classic 256 x 64 tile kernel (run-time epilogue)
  and on software-pipelined tile `tile` (specialised epilogue), twice: every
  output tensor equal bit for bit."""
  import ctypes
  got = []
  for t in (0, tile, tile):
    for o in outputs:
      o.fill_(7.0)
    d = make_desc()
    d.tile, d.stage_ksteps, d.split_parity, d.ksplit = t, 2, 0, 0
    rc = _lib.load().cg_swconv(ctypes.byref(d), H.stream())
    if rc == _lib.CG_EINVAL:
      pytest.skip('tile not admissible for this shape')
    assert rc == 0
    H.sync()
    got.append([o.clone() for o in outputs])
  for ref, a, b in zip(*got):
    bad = int((ref.float() != a.float()).sum()) + int((ref.float() != b.float()).sum())
    assert bad == 0, '%d elements differ from the classic tile' % bad


@pytest.mark.parametrize('tile', [13, 14, 15])
@pytest.mark.parametrize('nB,L,Ci,Co', _CRITIC)
def test_lean_epilogues_equal_the_classic_tiles_at_full_size(tile, nB, L, Ci, Co):
  """Round 4's specialised epilogues (bias + LeakyReLU, LeakyReLU' mask in
  place, mask + PhaseShuffle adjoint) against the classic tile kernel on
  small-integer data at the critic's full layer sizes.  The small kernel tests
  cannot see what this one is for: a 16-byte buffer store whose data registers
  the compiler let the next instructions overwrite (scalar offset in the
  store: swconv_swp.hip, store_rows) lost about one element in 10^4, only under
  the memory back-pressure of a full-size launch."""
  nets_autotune = nets._AUTOTUNE
  nets._AUTOTUNE = False
  was = _lib.load().cg_debug_lean_epilogue(1)  # (off by default: no faster)
  try:
    gen = torch.Generator(device=H.DEV)
    gen.manual_seed(99)
    k = 24
    cip, cop = geo.pitch(Ci), geo.pitch(Co)
    pl = geo.same_padding_left(k, 2)
    x = _pitched(_rand_int(gen, (nB, L, Ci), -2, 2), cip)
    W = _rand_int(gen, (k, Ci, Co), -1, 1, 0.5)
    bias = _rand_int(gen, (Co,), -2, 2)
    ck = nets._ck_for(cip, 2, k, L // 2)
    op = H.pack(W, [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k, parity_major=True)
    shifts = torch.tensor([3, -7, 10], dtype=torch.int32, device=H.DEV)
    # forward: bias + LeakyReLU (kEpiLrelu), fused input-side PhaseShuffle
    y = torch.empty(nB, L // 2, cop, dtype=BF16, device=H.DEV)
    _swp_vs_classic(
        lambda: H.conv_desc(x, op.buf, y, nB, L, cip, k, 2, -pl, L // 2, Co,
                            L // 2, cop, ck, bias=bias, shifts=shifts,
                            seg_size=128, epilogue=_lib.EPI_LRELU,
                            w_parity_major=True, w_narrow_last=op.narrow_last),
        [y], tile)
    # tangent form: LeakyReLU' mask read from another tensor (kEpiMask)
    h = _pitched(_rand_int(gen, (nB, L // 2, Co), -2, 2), cop)
    _swp_vs_classic(
        lambda: H.conv_desc(x, op.buf, y, nB, L, cip, k, 2, -pl, L // 2, Co,
                            L // 2, cop, ck, mask_src=h, shifts=shifts,
                            seg_size=128, epilogue=_lib.EPI_MASK,
                            w_parity_major=True, w_narrow_last=op.narrow_last),
        [y], tile)
    # input gradient: two phases, mask + output-side PhaseShuffle adjoint
    # (kEpiMaskShift), reflected rows through the side buffer
    phases = nets._transpose_phases(k, pl)
    ckd = nets._ck_for(cop, 1, k // 2, L // 2)
    opd = H.pack(W, [(t0, -2, Ci * Co, 1, Co) for t0, _ in phases], Co, Ci, cop,
                 ckd, k // 2)
    g = _pitched(_rand_int(gen, (nB, L // 2, Co), -2, 2), cop)
    hx = _pitched(_rand_int(gen, (nB, L, Ci), -2, 2), cip)
    e = torch.empty(nB, L, cip, dtype=BF16, device=H.DEV)
    side = torch.empty(nB, 10, cip, dtype=BF16, device=H.DEV)
    offs = [o for _, o in phases]
    _swp_vs_classic(
        lambda: H.conv_desc(g, opd.buf, e, nB, L // 2, cop, k // 2, 1, offs[0],
                            L // 2, Ci, L, cip, ckd, y_stride=2, y_off=0, nphase=2,
                            w_phase_stride=opd.elems,
                            off_phase_step=offs[1] - offs[0], yoff_phase_step=1,
                            mask_src=hx, epilogue=_lib.EPI_MASK,
                            out_shifts=(shifts, 128, side, 10)),
        [e, side], tile)
  finally:
    nets._AUTOTUNE = nets_autotune
    _lib.load().cg_debug_lean_epilogue(was)


def test_flex_weight_gradients_equal_the_split_form_at_full_size():
  """cg_wgrad_batched over the critic's five layers at cfg2's full size (3 x 128
  samples): the flex form (round 5: layers side by side, contiguous K' shares,
  ~1.4 partial tiles per workgroup) against the K'-split form it replaces, on
  small-integer data -- every partial sum is an integer below 2^24, so both forms
  are exact and dW / dbias must be EQUAL bit for bit whatever the order of the
  sums.  (a new store form is believed only after a full-size
  bit-exact test; autodiff of calciumgan.py:145-185.)"""
  import ctypes
  lib = _lib.load()
  gen = torch.Generator(device=H.DEV)
  gen.manual_seed(77)
  k, seg = 24, 128
  descs = {0: [], 1: []}
  outs = {0: [], 1: []}
  keep = []
  for li, (nB, L, Ci, Co) in enumerate(_CRITIC):
    cip, cop = geo.pitch(Ci), geo.pitch(Co)
    x = _pitched(_rand_int(gen, (nB, L, Ci), -2, 2), cip)
    g = _pitched(_rand_int(gen, (nB, L // 2, Co), -1, 1), cop)
    sh = torch.tensor([5, -9, 2], dtype=torch.int32, device=H.DEV)
    keep += [x, g, sh]
    for mode in (0, 1):
      dw = torch.full((k, Ci, Co), 7.0, dtype=torch.float32, device=H.DEV)
      db = torch.full((Co,), 7.0, dtype=torch.float32, device=H.DEV)
      d = nets._wgrad_desc(x, g, dw, nB, L, cip, L // 2, cop, k, 2,
                           -geo.same_padding_left(k, 2), Ci, Co,
                           shifts=sh if li else None, seg_size=seg, dbias=db,
                           bias_rows=2 * seg * (L // 2), slot=(mode, li))
      assert d.partials and d.store
      descs[mode].append(d)
      outs[mode].append((dw, db))
  was = lib.cg_debug_wgrad_flex(-1)
  try:
    for mode in (0, 1):
      lib.cg_debug_wgrad_flex(mode)
      arr = (_lib.WgradDesc * 5)(*descs[mode])
      _lib.call('cg_wgrad_batched', arr, 5, H.stream())
      H.sync()
  finally:
    lib.cg_debug_wgrad_flex(was)
  # the flex form was taken (the plan exists at this size)
  arr = (_lib.WgradDesc * 5)(*descs[1])
  assert lib.cg_wgrad_flex_plan(arr, 5, 1, None, 0, None) > 0
  for (dwa, dba), (dwb, dbb) in zip(outs[0], outs[1]):
    assert float(dwa.abs().max()) > 7.0
    assert torch.equal(dwa, dwb)
    assert torch.equal(dba, dbb)
  # and one layer against a torch contraction of the same operands (layer 5:
  # small enough for an f64 einsum on the device)
  nB, L, Ci, Co = _CRITIC[4]
  x, g, sh = keep[12], keep[13], keep[14]
  xs = _shuffled(x[:, :, :Ci].double(), [5, -9, 2], seg)
  xp = torch.nn.functional.pad(xs, (0, 0, 11, 11))
  ref = torch.stack([torch.einsum('bui,buo->io', xp[:, t:t + L:2], g[:, :, :Co].double())
                     for t in range(k)])
  assert torch.equal(outs[1][4][0].double(), ref)
