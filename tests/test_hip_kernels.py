"""GPU parity tests of the individual gfx950 kernels against the CPU oracle
(torch autograd on the oracle's layer definitions).  All calls go through the
C ABI (ctypes).  MFMA kernels are checked BIT-EXACTLY on small-integer data
(exact in bf16, products/sums exact in f32); float kernels within a stated
tolerance."""
import ctypes

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
ALPHA = 0.3


def _shuffle_batch(x, shifts, seg):
  out = []
  for b in range(x.shape[0]):
    out.append(O.phase_shuffle(x[b:b + 1], int(shifts[b // seg])))
  return torch.cat(out, 0)


def test_pack_matches_numpy_layout():
  rng = np.random.RandomState(0)
  k, ci, co = 24, 10, 70
  W = torch.tensor(rng.randn(k, ci, co).astype(np.float32)).to(H.DEV)
  cx = geo.pitch(ci)
  op = H.pack(W, [(0, 1, ci * co, co, 1)], ci, co, cx, cx, k)
  H.sync()
  exp = H.numpy_pack(
      W.cpu().to(BF16).float().numpy(), cx, cx)
  got = op.buf.float().cpu().numpy().reshape(exp.shape)
  np.testing.assert_array_equal(got, exp)


CONV_CASES = [
    # nB, L, Ci, Co, k, seg, use_shift
    (3, 128, 102, 64, 24, 1, True),
    (2, 64, 16, 40, 24, 2, False),
    (4, 16, 64, 72, 8, 2, True),
    (2, 512, 64, 128, 24, 1, True),
    (5, 32, 192, 200, 24, 5, False),
    (96, 8, 128, 160, 24, 32, True),
]


@pytest.mark.parametrize('nB,L,Ci,Co,k,seg,use_shift', CONV_CASES)
def test_conv_fwd_bitexact(nB, L, Ci, Co, k, seg, use_shift):
  """Conv1D 'same' s=2 + bias + LeakyReLU with fused phase-shuffle gather
  (calciumgan.py:145-150 + :117-138) on exact data."""
  rng = np.random.RandomState(1)
  x = H.int_tensor(rng, (nB, L, Ci))
  W = H.int_tensor(rng, (k, Ci, Co), -2, 2, 0.5)
  b = H.int_tensor(rng, (Co,), -4, 4)
  nseg = (nB + seg - 1) // seg
  shifts = rng.randint(-2, 3, size=nseg).astype(np.int32)
  if not use_shift:
    shifts[:] = 0
  ref = O.leaky_relu(
      O.conv1d_same(_shuffle_batch(x, shifts, seg), W, b, 2)).to(BF16).float()
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  Lo = L // 2
  pl = geo.same_padding_left(k, 2)
  Wd, bd = W.to(H.DEV), b.to(H.DEV)
  ck = nets._ck_for(cip, 2, k, Lo)
  op = H.pack(Wd, [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k)
  xd = H.to_pitch(x, cip)
  y = torch.full((nB, Lo, cop), 7.0, dtype=BF16, device=H.DEV)
  sh = torch.tensor(shifts, device=H.DEV)
  d = H.conv_desc(xd, op.buf, y, nB, L, cip, k, 2, -pl, Lo, Co, Lo, cop, ck,
                  bias=bd, shifts=sh if use_shift else None, seg_size=seg,
                  epilogue=_lib.EPI_LRELU)
  H.run_conv(d)
  H.sync()
  got = y.float().cpu()
  np.testing.assert_array_equal(got[:, :, :Co].numpy(), ref.numpy())
  assert float(got[:, :, Co:].abs().max()) == 0.0 if cop > Co else True


ALL_TILES = ([(t, ks) for t in sorted(_lib.TILES) for ks in (2, 4)] +
             [(t, 2) for t in sorted(_lib.SWP_TILES)])


@pytest.fixture(params=[0, 1], ids=['generic_epi', 'lean_epi'])
def epi_mode(request):
  """Both epilogue forms of the 32-row software-pipelined tiles: the run-time
  one (default) and the specialised ones (cg_debug_lean_epilogue).  Tests that
  also sweep tiles parametrise it indirectly, with the specialised form only on
  the tiles that have it (13 / 14 / 15)."""
  was = _lib.load().cg_debug_lean_epilogue(request.param)
  yield request.param
  _lib.load().cg_debug_lean_epilogue(was)


LEAN_TILES = (13, 14, 15)
_PH = torch.zeros(1)  # placeholder tensor of the collection-time descriptors


def _admits(d, tile, ks, sp=0):
  """cg_swconv_check (host only) of descriptor d on (tile, stage depth,
  split-parity): the parametrisations below are built from it at collection
  time, so only admissible combinations become test cases (405
  run-time skips hid nothing, but would have hidden a real one)."""
  d.tile, d.stage_ksteps, d.split_parity = tile, ks, sp
  return _lib.load().cg_swconv_check(ctypes.byref(d)) == 0


def _force_tile(d, tile, ks, sp=0):
  """Override the static choice with an admissible (tile, stage depth,
  split-parity): cg_swconv must accept it."""
  d.tile, d.stage_ksteps, d.split_parity = tile, ks, sp
  assert _lib.load().cg_swconv(ctypes.byref(d), H.stream()) == 0


def _fwd_desc(xd, wbuf, y, nB, L, Ci, Co, k, sp, **kw):
  """Descriptor of the stride-2 forward tests (device tensors, or _PH at
  collection time)."""
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  Lo = L // 2
  ck = nets._ck_for(cip, 2, k, Lo)
  return H.conv_desc(xd, wbuf, y, nB, L, cip, k, 2, -geo.same_padding_left(k, 2),
                     Lo, Co, Lo, cop, ck, w_parity_major=bool(sp),
                     w_narrow_last=nets.narrow_last_rule(bool(sp), ck, cip, k, Ci),
                     **kw)


def _dgrad_desc(dyd, wbuf, y, nB, L, Ci, Co, k, **kw):
  """Two-phase transposed convolution (input gradient of a stride-2 conv)."""
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  offs = [o for _, o in nets._transpose_phases(k, geo.same_padding_left(k, 2))]
  ck = nets._ck_for(cop, 1, k // 2, L // 2)
  elems = _lib.load().cg_packed_elems(Ci, k // 2, cop, ck)
  return H.conv_desc(dyd, wbuf, y, nB, L // 2, cop, k // 2, 1, offs[0], L // 2,
                     Ci, L, cip, ck, y_stride=2, y_off=0, nphase=2,
                     w_phase_stride=elems, off_phase_step=offs[1] - offs[0],
                     yoff_phase_step=1, **kw)


def _tile_cases(shapes, admit, sps=(0,), tiles=None):
  """pytest params (tile, ks, *shape, sp, epi_mode) for every admissible
  combination; epi_mode 1 (specialised epilogues) only on LEAN_TILES."""
  out = []
  for shape in shapes:
    for tile, ks in (tiles or ALL_TILES):
      for sp in sps:
        if not admit(tile, ks, sp, *shape):
          continue
        for em in (0, 1):
          if em and tile not in LEAN_TILES:
            continue
          out.append(pytest.param(
              tile, ks, *shape, sp, em,
              id='t{}k{}-{}-sp{}-{}'.format(tile, ks, 'x'.join(map(str, shape)), sp,
                                            'lean' if em else 'generic')))
  return out


FWD_SHAPES = [(2, 1024, 64, 192, 24, 1), (6, 128, 96, 102, 24, 2),
              (3, 512, 32, 64, 8, 3), (3, 1024, 102, 64, 24, 2),
              (4, 64, 70, 40, 8, 4)]


def _fwd_admits(tile, ks, sp, nB, L, Ci, Co, k, seg):
  d = _fwd_desc(_PH, _PH, _PH, nB, L, Ci, Co, k, sp, bias=_PH, shifts=_PH,
                seg_size=seg, epilogue=_lib.EPI_LRELU)
  return _admits(d, tile, ks, sp)


@pytest.mark.parametrize('tile,ks,nB,L,Ci,Co,k,seg,sp,epi_mode',
                         _tile_cases(FWD_SHAPES, _fwd_admits, sps=(0, 1)),
                         indirect=['epi_mode'])
def test_conv_fwd_every_tile(tile, ks, nB, L, Ci, Co, k, seg, sp, epi_mode):
  """Stride-2 forward with phase shuffle + bias + LeakyReLU on every workgroup
  tile (both MFMA shapes, 4x1 and 2x2 waves) and both weight-stage depths,
  including partial column tiles (192, 102) and several samples per tile;
  sp = 1: parity-major weights with split-parity window staging."""
  rng = np.random.RandomState(2)
  x = H.int_tensor(rng, (nB, L, Ci))
  W = H.int_tensor(rng, (k, Ci, Co), -2, 2, 0.5)
  b = H.int_tensor(rng, (Co,), -4, 4)
  nseg = (nB + seg - 1) // seg
  shifts = rng.randint(-2, 3, size=nseg).astype(np.int32)
  ref = O.leaky_relu(
      O.conv1d_same(_shuffle_batch(x, shifts, seg), W, b, 2)).to(BF16).float()
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  Lo = L // 2
  pl = geo.same_padding_left(k, 2)
  ck = nets._ck_for(cip, 2, k, Lo)
  op = H.pack(W.to(H.DEV), [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k,
              parity_major=bool(sp))
  xd = H.to_pitch(x, cip)
  y = torch.full((nB, Lo, cop), 7.0, dtype=BF16, device=H.DEV)
  sh = torch.tensor(shifts, device=H.DEV)
  bd = b.to(H.DEV)
  # Ci = 102 / 70 with parity-major weights: the last 32-channel chunk holds
  # 6 real channels and is packed / walked narrow (cg_pack_desc.narrow_last)
  assert op.narrow_last == (bool(sp) and Ci in (102, 70))
  d = H.conv_desc(xd, op.buf, y, nB, L, cip, k, 2, -pl, Lo, Co, Lo, cop, ck,
                  bias=bd, shifts=sh, seg_size=seg, epilogue=_lib.EPI_LRELU,
                  w_parity_major=bool(sp), w_narrow_last=op.narrow_last)
  _force_tile(d, tile, ks, sp)
  H.sync()
  got = y.float().cpu()
  np.testing.assert_array_equal(got[:, :, :Co].numpy(), ref.numpy())
  if cop > Co:
    assert float(got[:, :, Co:].abs().max()) == 0.0
  if sp and _admits(d, tile, ks, 0):
    # the same parity-major (and narrow) operand with both parities resident
    y.fill_(7.0)
    _force_tile(d, tile, ks, 0)
    H.sync()
    np.testing.assert_array_equal(y.float().cpu()[:, :, :Co].numpy(),
                                  ref.numpy())


def _row_scale_admits(tile, ks, sp, nB, L, Ci, Co, k, epi):
  d = _fwd_desc(_PH, _PH, _PH, nB, L, Ci, Co, k, 1, bias=_PH, epilogue=epi,
                mask_src=_PH if epi == _lib.EPI_MASK else None, row_scale=_PH)
  return _admits(d, tile, ks, 0)


@pytest.mark.parametrize(
    'tile,ks,nB,L,Ci,Co,k,epi,sp,epi_mode',
    _tile_cases([(5, 512, 102, 64, 24, e) for e in (_lib.EPI_NONE, _lib.EPI_MASK)] +
                [(3, 128, 64, 128, 24, e) for e in (_lib.EPI_NONE, _lib.EPI_MASK)],
                _row_scale_admits, tiles=[(t, 2) for t in sorted(_lib.SWP_TILES)]),
    indirect=['epi_mode'])
def test_conv_fwd_row_scale(tile, ks, nB, L, Ci, Co, k, epi, sp, epi_mode):
  """cg_conv_desc.row_scale: y = epi((acc + bias) * row_scale[sample]) on the
  software-pipelined tiles -- the penalty's v = coef_b * g folded into the
  tangent chain's first launch.  Power-of-two scales on small-integer data:
  bit-exact against the convolution of the pre-scaled input; the classic tiles,
  split-K and the fused LayerNorm refuse the field."""
  rng = np.random.RandomState(11)
  x = H.int_tensor(rng, (nB, L, Ci))
  W = H.int_tensor(rng, (k, Ci, Co), -2, 2, 0.5)
  b = H.int_tensor(rng, (Co,), -4, 4)
  scale = torch.tensor(2.0 ** rng.randint(-3, 3, size=nB), dtype=torch.float32)
  scale[1] = -scale[1]
  lin = O.conv1d_same(x, W, b, 2) * scale[:, None, None]
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  Lo = L // 2
  pl = geo.same_padding_left(k, 2)
  ck = nets._ck_for(cip, 2, k, Lo)
  op = H.pack(W.to(H.DEV), [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k,
              parity_major=True)
  xd = H.to_pitch(x, cip)
  y = torch.full((nB, Lo, cop), 7.0, dtype=BF16, device=H.DEV)
  mask = None
  ref = lin
  if epi == _lib.EPI_MASK:
    m = H.int_tensor(rng, (nB, Lo, Co), -1, 1)
    mask = H.to_pitch(m, cop)
    ref = lin * torch.where(m > 0, 1.0, nets.LEAKY_ALPHA)
  sd = scale.to(H.DEV)
  d = H.conv_desc(xd, op.buf, y, nB, L, cip, k, 2, -pl, Lo, Co, Lo, cop, ck,
                  bias=b.to(H.DEV), epilogue=epi, mask_src=mask,
                  w_parity_major=True, w_narrow_last=op.narrow_last,
                  row_scale=sd)
  _force_tile(d, tile, 2, 0)
  H.sync()
  got = y.float().cpu()
  np.testing.assert_array_equal(got[:, :, :Co].numpy(),
                                ref.to(BF16).float().numpy())
  if cop > Co:
    assert float(got[:, :, Co:].abs().max()) == 0.0
  lib = _lib.load()
  d.tile = 0                                   # a classic tile
  assert lib.cg_swconv_check(ctypes.byref(d)) == _lib.CG_EINVAL
  assert lib.cg_swconv(ctypes.byref(d), H.stream()) == _lib.CG_EINVAL


def _split_k_cases():
  out = []
  for nB, L, Ci, Co, k, epi in [(3, 256, 128, 192, 24, 1), (2, 128, 102, 64, 24, 2),
                                (4, 64, 256, 320, 24, 0), (2, 512, 256, 128, 24, 2)]:
    cip = geo.pitch(Ci)
    ck = nets._ck_for(cip, 2, k, L // 2)
    for tile in [0, 1, 2, 4, 8, 9, 10, 11, 12, 13, 14, 15]:
      for ksplit in (2, 4):
        if (cip // ck) % ksplit:
          continue
        d = _fwd_desc(_PH, _PH, _PH, nB, L, Ci, Co, k, 1,
                      bias=_PH if epi == 1 else None, shifts=_PH, seg_size=1,
                      epilogue=epi, mask_src=_PH if epi == 2 else None)
        d.ksplit, d.split_ws, d.split_ws_elems = ksplit, _PH.data_ptr(), 1 << 40
        if _admits(d, tile, 2, 0):
          out.append((tile, ksplit, nB, L, Ci, Co, k, epi))
  return out


@pytest.mark.parametrize('tile,ksplit,nB,L,Ci,Co,k,epi', _split_k_cases())
def test_conv_fwd_split_k(tile, ksplit, nB, L, Ci, Co, k, epi):
  """cg_conv_desc.ksplit: several workgroups per output tile, each over a share
  of the channel chunks, f32 partial sums + a finishing launch (bias /
  LeakyReLU / in-place mask).  Exact on integer data, so identical to the
  unsplit launch and to the oracle; includes the narrow last chunk (Ci = 102),
  which only the last split walks (tile kernels; the software-pipelined tiles
  9-12 split whole chunks only and answer CG_EINVAL there)."""
  rng = np.random.RandomState(12)
  x = H.int_tensor(rng, (nB, L, Ci))
  W = H.int_tensor(rng, (k, Ci, Co), -2, 2, 0.5)
  b = H.int_tensor(rng, (Co,), -4, 4)
  shifts = rng.randint(-2, 3, size=nB).astype(np.int32)
  pre = O.conv1d_same(_shuffle_batch(x, shifts, 1), W, b if epi == 1 else None,
                      2)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  Lo = L // 2
  pl = geo.same_padding_left(k, 2)
  ck = nets._ck_for(cip, 2, k, Lo)
  op = H.pack(W.to(H.DEV), [(0, 1, Ci * Co, Co, 1)], Ci, Co, cip, ck, k,
              parity_major=True)
  y = torch.full((nB, Lo, cop), 7.0, dtype=BF16, device=H.DEV)
  hmask = None
  if epi == 2:  # in place: y holds the activations whose sign masks the result
    hm = torch.tensor(rng.randn(nB, Lo, Co).astype(np.float32))
    y = H.to_pitch(hm, cop)
    hmask = hm.to(BF16).float()
    ref = (pre * torch.where(hmask > 0, 1.0, ALPHA)).to(BF16).float()
  elif epi == 1:
    ref = O.leaky_relu(pre).to(BF16).float()
  else:
    ref = pre.to(BF16).float()
  ws = torch.full((ksplit * nB * Lo * cop,), float('nan'), device=H.DEV)
  d = H.conv_desc(H.to_pitch(x, cip), op.buf, y, nB, L, cip, k, 2, -pl, Lo, Co,
                  Lo, cop, ck, bias=b.to(H.DEV) if epi == 1 else None,
                  shifts=torch.tensor(shifts, device=H.DEV), seg_size=1,
                  epilogue=epi, mask_src=y if epi == 2 else None,
                  w_parity_major=True, w_narrow_last=op.narrow_last)
  d.ksplit, d.split_ws, d.split_ws_elems = ksplit, ws.data_ptr(), ws.numel()
  _force_tile(d, tile, 2, 0)
  H.sync()
  got = y.float().cpu()
  np.testing.assert_array_equal(got[:, :, :Co].numpy(), ref.numpy())
  if cop > Co:
    assert float(got[:, :, Co:].abs().max()) == 0.0


def _dgrad_admits(tile, ks, sp, nB, L, Ci, Co, k):
  d = _dgrad_desc(_PH, _PH, _PH, nB, L, Ci, Co, k, out_f32=True)
  if L // 2 >= _lib.tile_shape(tile)[0]:
    d.rowsumsq = _PH.data_ptr()
  return _admits(d, tile, ks, 0)


@pytest.mark.parametrize('tile,ks,nB,L,Ci,Co,k,sp,epi_mode',
                         _tile_cases([(2, 512, 102, 128, 24), (5, 128, 192, 256, 24)],
                                     _dgrad_admits),
                         indirect=['epi_mode'])
def test_conv_dgrad_every_tile(tile, ks, nB, L, Ci, Co, k, sp, epi_mode):
  """Two-phase transposed convolution (f32 out, strided rows) + the fused
  per-sample sum of squares on every tile."""
  rng = np.random.RandomState(3)
  W = H.int_tensor(rng, (k, Ci, Co), -1, 1, 0.5)
  dy = H.int_tensor(rng, (nB, L // 2, Co), -2, 2)
  x = torch.zeros(nB, L, Ci, requires_grad=True)
  (O.conv1d_same(x, W, None, 2) * dy).sum().backward()
  ref = x.grad
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  phases = nets._transpose_phases(k, pl)
  ck = nets._ck_for(cop, 1, k // 2, L // 2)
  op = H.pack(W.to(H.DEV), [(t0, -2, Ci * Co, 1, Co) for t0, _ in phases], Co,
              Ci, cop, ck, k // 2)
  y = torch.full((nB, L, cip), 3.0, dtype=torch.float32, device=H.DEV)
  offs = [o for _, o in phases]
  dyd = H.to_pitch(dy, cop)
  d = H.conv_desc(dyd, op.buf, y, nB, L // 2, cop, k // 2, 1, offs[0], L // 2,
                  Ci, L, cip, ck, y_stride=2, y_off=0, out_f32=True, nphase=2,
                  w_phase_stride=op.elems, off_phase_step=offs[1] - offs[0],
                  yoff_phase_step=1)
  rows, _ = _lib.tile_shape(tile)
  ssq = None
  if L // 2 >= rows:  # one sample per tile: the sum of squares can be fused
    ssq = torch.zeros(nB, dtype=torch.float32, device=H.DEV)
    d.rowsumsq = ssq.data_ptr()
  _force_tile(d, tile, ks)
  H.sync()
  got = y.cpu()
  np.testing.assert_array_equal(got[:, :, :Ci].numpy(), ref.numpy())
  if cip > Ci:
    assert float(got[:, :, Ci:].abs().max()) == 0.0
  if ssq is not None:
    want = (ref**2).sum(dim=(1, 2)).numpy()
    np.testing.assert_allclose(ssq.cpu().numpy(), want, rtol=1e-5)
    # the ordered form: one slot per workgroup, summed by a finishing launch and
    # STORED (the output starts poisoned); twice the same bits
    d.tile, d.stage_ksteps = tile, ks
    need = _lib.load().cg_rowsumsq_ws_elems(ctypes.byref(d))
    assert need > 0
    wsq = torch.full((need,), float('nan'), device=H.DEV)
    d.rowsumsq_ws, d.rowsumsq_ws_elems = wsq.data_ptr(), need
    runs = []
    for _ in range(2):
      ssq.fill_(12345.0)
      H.run_conv(d)
      H.sync()
      runs.append(ssq.clone())
    np.testing.assert_allclose(runs[0].cpu().numpy(), want, rtol=1e-5)
    assert torch.equal(runs[0], runs[1])


DGRAD_CASES = [
    (3, 128, 102, 64, 24),
    (2, 64, 16, 40, 24),
    (4, 16, 64, 72, 8),
    (2, 512, 64, 128, 24),
    (48, 16, 96, 128, 24),
]


@pytest.mark.parametrize('nB,L,Ci,Co,k', DGRAD_CASES)
def test_conv_dgrad_bitexact(nB, L, Ci, Co, k):
  """Input-gradient of Conv1D 'same' s=2 == the two-phase transposed conv."""
  rng = np.random.RandomState(3)
  W = H.int_tensor(rng, (k, Ci, Co), -2, 2, 0.5)
  dy = H.int_tensor(rng, (nB, L // 2, Co))
  x = torch.zeros(nB, L, Ci, requires_grad=True)
  (O.conv1d_same(x, W, None, 2) * dy).sum().backward()
  ref = x.grad
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  phases = nets._transpose_phases(k, pl)
  ck = nets._ck_for(cop, 1, k // 2, L // 2)
  op = H.pack(W.to(H.DEV), [(t0, -2, Ci * Co, 1, Co) for t0, _ in phases], Co,
              Ci, cop, ck, k // 2)
  y = torch.full((nB, L, cip), 3.0, dtype=torch.float32, device=H.DEV)
  offs = [o for _, o in phases]
  d = H.conv_desc(H.to_pitch(dy, cop), op.buf, y, nB, L // 2, cop, k // 2, 1,
                  offs[0], L // 2, Ci, L, cip, ck, y_stride=2, y_off=0,
                  out_f32=True, nphase=2, w_phase_stride=op.elems,
                  off_phase_step=offs[1] - offs[0], yoff_phase_step=1)
  H.run_conv(d)
  H.sync()
  got = y.cpu()
  np.testing.assert_array_equal(got[:, :, :Ci].numpy(), ref.numpy())
  if cip > Ci:
    assert float(got[:, :, Ci:].abs().max()) == 0.0


@pytest.mark.parametrize('B,L,Ci,Co,k', [(2, 64, 32, 320, 24),
                                          (3, 128, 128, 102, 24),
                                          (2, 8, 32, 40, 24)])
def test_conv_transpose_fwd_bitexact(B, L, Ci, Co, k):
  """Conv1DTranspose (gan/models/utils.py:65-94) + bias."""
  rng = np.random.RandomState(4)
  x = H.int_tensor(rng, (B, L, Ci))
  Wt = H.int_tensor(rng, (k, 1, Co, Ci), -2, 2, 0.5)
  b = H.int_tensor(rng, (Co,), -4, 4)
  ref = O.conv1d_transpose_same(x, Wt, b, 2)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  phases = nets._transpose_phases(k, pl)
  offs = [o for _, o in phases]
  ck = nets._ck_for(cip, 1, k // 2, L)
  op = H.pack(Wt.to(H.DEV), [(t0, -2, Co * Ci, 1, Ci) for t0, _ in phases], Ci,
              Co, cip, ck, k // 2)
  y = torch.zeros(B, 2 * L, cop, dtype=torch.float32, device=H.DEV)
  d = H.conv_desc(H.to_pitch(x, cip), op.buf, y, B, L, cip, k // 2, 1, offs[0],
                  L, Co, 2 * L, cop, ck, y_stride=2, bias=b.to(H.DEV),
                  out_f32=True, nphase=2, w_phase_stride=op.elems,
                  off_phase_step=offs[1] - offs[0], yoff_phase_step=1)
  H.run_conv(d)
  H.sync()
  np.testing.assert_array_equal(y.cpu()[:, :, :Co].numpy(), ref.numpy())


def _ln_desc(xd, wbuf, y, B, L, Ci, Co, k, **kw):
  """Conv1DTranspose forward (two 12-tap phases) of the fused-LayerNorm test."""
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  offs = [o for _, o in nets._transpose_phases(k, geo.same_padding_left(k, 2))]
  ck = nets._ck_for(cip, 1, k // 2, L)
  elems = _lib.load().cg_packed_elems(Co, k // 2, cip, ck)
  return H.conv_desc(xd, wbuf, y, B, L, cip, k // 2, 1, offs[0], L, Co, 2 * L,
                     cop, ck, y_stride=2, nphase=2, w_phase_stride=elems,
                     off_phase_step=offs[1] - offs[0], yoff_phase_step=1, **kw)


def _ln_cases():
  out = []
  for B, L, Ci, Co, k in [(3, 256, 128, 102, 24), (2, 512, 192, 128, 24),
                          (5, 16, 64, 40, 24), (2, 4, 32, 6, 8)]:
    for tile, ks in [(5, 2), (5, 4), (6, 2), (6, 4), (7, 2), (7, 4), (8, 2),
                     (8, 4), (11, 2), (12, 2), (15, 2)]:
      tm = _lib.tile_shape(tile)[0]
      if not ((L % tm == 0) if L >= tm else (tm % L == 0)):
        continue
      d = _ln_desc(_PH, _PH, _PH, B, L, Ci, Co, k, bias=_PH,
                   ln=(_PH, _PH, _PH, _PH, _PH))
      if _admits(d, tile, ks):
        out.append((tile, ks, B, L, Ci, Co, k))
  return out


@pytest.mark.parametrize('tile,ks,B,L,Ci,Co,k', _ln_cases())
def test_conv_transpose_layernorm_fused(tile, ks, B, L, Ci, Co, k):
  """Conv1DTranspose + LayerNormalization + LeakyReLU (calciumgan.py:61-70) in
  one launch == the same launch without the fusion followed by the separate
  cg_ln_lrelu_fwd pass: the stored pre-activation bit for bit, the statistics
  and the activation to rounding."""
  rng = np.random.RandomState(21)
  x = torch.tensor(rng.randn(B, L, Ci).astype(np.float32))
  Wt = torch.tensor(rng.randn(k, 1, Co, Ci).astype(np.float32) * 0.05)
  b = torch.tensor(rng.randn(Co).astype(np.float32) * 0.2)
  gam = torch.tensor(rng.rand(Co).astype(np.float32) + 0.5).to(H.DEV)
  bet = torch.tensor(rng.randn(Co).astype(np.float32) * 0.1).to(H.DEV)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  phases = nets._transpose_phases(k, pl)
  offs = [o for _, o in phases]
  ck = nets._ck_for(cip, 1, k // 2, L)
  op = H.pack(Wt.to(H.DEV), [(t0, -2, Co * Ci, 1, Ci) for t0, _ in phases], Ci,
              Co, cip, ck, k // 2)
  rows = B * 2 * L
  xd = H.to_pitch(x, cip)
  z = lambda *s, dt=BF16: torch.zeros(*s, dtype=dt, device=H.DEV)
  y0, h0 = z(B, 2 * L, cop), z(B, 2 * L, cop)
  m0, r0 = z(rows, dt=torch.float32), z(rows, dt=torch.float32)
  y1, h1 = z(B, 2 * L, cop), z(B, 2 * L, cop)
  m1, r1 = z(rows, dt=torch.float32), z(rows, dt=torch.float32)
  common = dict(y_stride=2, bias=b.to(H.DEV), nphase=2, w_phase_stride=op.elems,
                off_phase_step=offs[1] - offs[0], yoff_phase_step=1)
  d0 = H.conv_desc(xd, op.buf, y0, B, L, cip, k // 2, 1, offs[0], L, Co, 2 * L,
                   cop, ck, **common)
  H.run_conv(d0)
  _lib.call('cg_ln_lrelu_fwd', H.p(y0), H.p(gam), H.p(bet), H.p(h0), H.p(m0),
            H.p(r0), rows, Co, cop, 1e-3, ALPHA, H.stream())
  d1 = H.conv_desc(xd, op.buf, y1, B, L, cip, k // 2, 1, offs[0], L, Co, 2 * L,
                   cop, ck, ln=(gam, bet, h1, m1, r1), **common)
  d1.tile, d1.stage_ksteps = tile, ks
  assert _lib.load().cg_swconv_check(ctypes.byref(d1)) == 0
  H.run_conv(d1)
  # forward-only form (no statistics buffers): same activation, y not written
  y2 = torch.full((B, 2 * L, cop), 5.0, dtype=BF16, device=H.DEV)
  h2 = z(B, 2 * L, cop)
  d2 = H.conv_desc(xd, op.buf, y2, B, L, cip, k // 2, 1, offs[0], L, Co, 2 * L,
                   cop, ck, ln=(gam, bet, h2, None, None), **common)
  d2.tile, d2.stage_ksteps = tile, ks
  H.run_conv(d2)
  H.sync()
  assert torch.equal(h2, h1)
  assert float((y2.float() - 5.0).abs().max()) == 0.0
  # the MFMA shapes of d0's tile and the 32x32x16 tiles may differ in the last
  # f32 bit of a partial sum -> at most one bf16 ulp on a rounding tie
  ya, yb = y0.float().cpu().numpy(), y1.float().cpu().numpy()
  np.testing.assert_allclose(yb, ya, rtol=2 ** -7, atol=1e-6)
  assert np.mean(ya != yb) < 1e-3
  # (a pre-activation that rounds the other way moves its row's mean by one
  # bf16 ulp / Co)
  np.testing.assert_allclose(m1.cpu().numpy(), m0.cpu().numpy(), rtol=1e-4,
                             atol=1e-3)
  np.testing.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=2e-3)
  ha, hb = h0.float().cpu().numpy(), h1.float().cpu().numpy()
  np.testing.assert_allclose(hb, ha, rtol=2 ** -6, atol=2e-3)
  assert np.mean(ha != hb) < 2e-2
  if cop > Co:
    assert float(y1[:, :, Co:].float().abs().max()) == 0.0
    assert float(h1[:, :, Co:].float().abs().max()) == 0.0
  # and against the f64 oracle of the three reference layers
  ref = O.leaky_relu(O.layer_norm(
      O.conv1d_transpose_same(x.to(BF16).double(), Wt.to(BF16).double(),
                              b.double(), 2),
      gam.cpu().double(), bet.cpu().double()))
  np.testing.assert_allclose(hb[:, :, :Co], ref.numpy(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize('B,L,Ci,Co,k', [(2, 64, 32, 320, 24),
                                          (3, 128, 128, 102, 24)])
def test_conv_transpose_dgrad_bitexact(B, L, Ci, Co, k):
  rng = np.random.RandomState(5)
  Wt = H.int_tensor(rng, (k, 1, Co, Ci), -2, 2, 0.5)
  dy = H.int_tensor(rng, (B, 2 * L, Co))
  x = torch.zeros(B, L, Ci, requires_grad=True)
  (O.conv1d_transpose_same(x, Wt, None, 2) * dy).sum().backward()
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  ck = nets._ck_for(cop, 2, k, L)
  op = H.pack(Wt.to(H.DEV), [(0, 1, Co * Ci, Ci, 1)], Co, Ci, cop, ck, k)
  y = torch.zeros(B, L, cip, dtype=torch.float32, device=H.DEV)
  d = H.conv_desc(H.to_pitch(dy, cop), op.buf, y, B, 2 * L, cop, k, 2, -11, L,
                  Ci, L, cip, ck, out_f32=True)
  H.run_conv(d)
  H.sync()
  np.testing.assert_array_equal(y.cpu()[:, :, :Ci].numpy(), x.grad.numpy())


def test_dense_sigmoid_and_mask_epilogues():
  rng = np.random.RandomState(6)
  B, L, C = 2, 64, 102
  x = torch.tensor(rng.randn(B, L, C).astype(np.float32)).to(BF16).float()
  W = torch.tensor(rng.randn(C, C).astype(np.float32) * 0.1)
  b = torch.tensor(rng.randn(C).astype(np.float32))
  cp = geo.pitch(C)
  op = H.pack(W.to(H.DEV), [(0, 1, 0, C, 1)], C, C, cp, cp, 1)
  y = torch.zeros(B, L, cp, dtype=torch.float32, device=H.DEV)
  d = H.conv_desc(H.to_pitch(x, cp), op.buf, y, B, L, cp, 1, 1, 0, L, C, L, cp,
                  cp, bias=b.to(H.DEV), epilogue=_lib.EPI_SIGMOID, out_f32=True)
  H.run_conv(d)
  H.sync()
  ref = torch.sigmoid(x @ W.to(BF16).float() + b)
  np.testing.assert_allclose(
      y.cpu()[:, :, :C].numpy(), ref.numpy(), rtol=2e-5, atol=2e-6)
  assert float(y[:, :, C:].abs().max()) == 0.0
  # mask epilogue, in place over the mask source
  h = torch.tensor(rng.randn(B, L, C).astype(np.float32))
  hd = H.to_pitch(h, cp)
  d = H.conv_desc(H.to_pitch(x, cp), op.buf, hd, B, L, cp, 1, 1, 0, L, C, L, cp,
                  cp, mask_src=hd, epilogue=_lib.EPI_MASK)
  H.run_conv(d)
  H.sync()
  hq = h.to(BF16).float()
  ref = (x @ W.to(BF16).float()) * torch.where(hq > 0, 1.0, ALPHA)
  np.testing.assert_allclose(
      hd.float().cpu()[:, :, :C].numpy(), ref.to(BF16).float().numpy(),
      rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize('rows,Ci,Co,epi', [(1000, 102, 102, 3), (77, 32, 6, 0),
                                            (4096, 128, 128, 3), (33, 64, 40, 0),
                                            (50, 96, 70, 3)])
def test_dense_rows_streaming(rows, Ci, Co, epi):
  """cg_dense_rows == the 1-tap cg_swconv it replaces for the generator's last
  Dense (+ sigmoid): exact on integer data without activation, f32 tolerance
  with the sigmoid; padding channels zero; ragged row count."""
  rng = np.random.RandomState(21)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  x = H.int_tensor(rng, (1, rows, Ci), -3, 3)
  W = H.int_tensor(rng, (Ci, Co), -2, 2, 0.25)
  b = H.int_tensor(rng, (Co,), -2, 2, 0.5)
  op = H.pack(W.to(H.DEV), [(0, 1, 0, Co, 1)], Ci, Co, cip, 32, 1)
  y = torch.full((rows, cop), 9.0, dtype=torch.float32, device=H.DEV)
  xd = H.to_pitch(x, cip)
  bd = b.to(H.DEV)
  _lib.call('cg_dense_rows', H.p(xd), H.p(op.buf), H.p(bd), H.p(y), rows, cip,
            Co, cop, epi, H.stream())
  H.sync()
  ref = x[0] @ W + b
  got = y.cpu()
  if epi == 3:
    np.testing.assert_allclose(got[:, :Co].numpy(), torch.sigmoid(ref).numpy(),
                               rtol=2e-6, atol=1e-6)
  else:
    np.testing.assert_array_equal(got[:, :Co].numpy(), ref.numpy())
  if cop > Co:
    assert float(got[:, Co:].abs().max()) == 0.0


@pytest.mark.parametrize('n,B,L,Ci,C,epi', [(5, 3, 64, 128, 102, 3),
                                            (2, 70, 16, 64, 102, 0),
                                            (1, 2, 2048, 128, 128, 3),
                                            (8, 1, 32, 96, 97, 3),
                                            # the LDS-panel form (configs[4]: 512 -> 512)
                                            (2, 3, 32, 512, 512, 3),
                                            (3, 1, 64, 256, 200, 0),
                                            (1, 2, 16, 128, 130, 3),
                                            (5, 2, 80, 384, 300, 3)])
def test_dense_rows_interp_equals_dense_rows_then_interp_pack(n, B, L, Ci, C, epi):
  """cg_dense_rows_interp -- the generator's output Dense (+ sigmoid,
  calciumgan.py:96-101) over the fake batches of all n critic updates of a step,
  with the interpolation and packing of the critic's inputs (wgan_gp.py:38-41) in
  its epilogue -- against the two launches it replaces, cg_dense_rows (f32 fake
  batch) + one cg_interp_pack per update: every byte of every update's
  [real | fake_k | x^_k] equal."""
  rng = np.random.RandomState(33)
  cip = geo.pitch(Ci)
  cp = 128 if (Ci <= 128 and C <= 128) else geo.pitch(C)
  cf = (C + 7) // 8 * 8
  h = torch.tensor(rng.randn(1, n * B * L, Ci).astype(np.float32))
  W = torch.tensor(rng.randn(Ci, C).astype(np.float32) * 0.2)
  b = torch.tensor(rng.randn(C).astype(np.float32) * 0.1)
  real = torch.tensor(rng.rand(B, L, C).astype(np.float32)).to(H.DEV)
  alpha = torch.tensor(rng.rand(n * B).astype(np.float32)).to(H.DEV)
  op = H.pack(W.to(H.DEV), [(0, 1, 0, C, 1)], Ci, C, cip, 32, 1)
  hd, bd = H.to_pitch(h, cip), b.to(H.DEV)
  fake = torch.zeros(n * B * L, cf, dtype=torch.float32, device=H.DEV)
  _lib.call('cg_dense_rows', H.p(hd), H.p(op.buf), H.p(bd), H.p(fake), n * B * L,
            cip, C, cf, epi, H.stream())
  ref = []
  for k in range(n):
    x0 = torch.full((3 * B, L, cp), 7.0, dtype=BF16, device=H.DEV)
    _lib.call('cg_interp_pack', H.p(real), H.p(fake[k * B * L:]),
              H.p(alpha[k * B:]), H.p(x0), B, L, C, C, cf, cp, 1, H.stream())
    ref.append(x0)
  got = [torch.full((3 * B, L, cp), 9.0, dtype=BF16, device=H.DEV) for _ in range(n)]
  ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in got])
  _lib.call('cg_dense_rows_interp', H.p(hd), H.p(op.buf), H.p(bd), H.p(real),
            H.p(alpha), ptrs, n, B, L, cip, C, C, cp, epi, H.stream())
  H.sync()
  for k in range(n):
    assert float(ref[k].float().abs().max()) > 0
    assert torch.equal(ref[k].view(torch.int16), got[k].view(torch.int16)), k
  # alpha == NULL (callers that take layer 1 of x^ from the other two segments,
  # cg_lrelu_mix): [real | fake_k] as above, the third segment untouched -- by the
  # fused launch and by cg_interp_pack alike
  got2 = [torch.full((3 * B, L, cp), 9.0, dtype=BF16, device=H.DEV) for _ in range(n)]
  ptrs2 = (ctypes.c_void_p * n)(*[t.data_ptr() for t in got2])
  _lib.call('cg_dense_rows_interp', H.p(hd), H.p(op.buf), H.p(bd), H.p(real),
            None, ptrs2, n, B, L, cip, C, C, cp, epi, H.stream())
  x2 = torch.full((3 * B, L, cp), 9.0, dtype=BF16, device=H.DEV)
  _lib.call('cg_interp_pack', H.p(real), H.p(fake), None, H.p(x2), B, L, C, C, cf,
            cp, 1, H.stream())
  H.sync()
  for k in range(n):
    assert torch.equal(got2[k][:2 * B].view(torch.int16),
                       got[k][:2 * B].view(torch.int16)), k
    assert bool((got2[k][2 * B:].float() == 9.0).all()), k
  assert torch.equal(x2[:2 * B].view(torch.int16), ref[0][:2 * B].view(torch.int16))
  assert bool((x2[2 * B:].float() == 9.0).all())


@pytest.mark.parametrize('rows,Ci,Co,epi', [(1000, 512, 512, 3), (77, 256, 200, 0),
                                            (4100, 512, 300, 0), (33, 384, 130, 3),
                                            (9000, 128, 102, 0)])
def test_dense_rows_wide(rows, Ci, Co, epi):
  """The LDS-panel form of cg_dense_rows (K or N beyond 128: BASELINE
  configs[4]'s 512 -> 512 per-timestep Dense) and its activation-typed twin
  cg_dense_rows_act (the input gradient dh = dz W^T): exact on integer data,
  f32 tolerance through the sigmoid, padding columns zero, ragged row counts,
  several 128-column panels, row blocks that wrap the grid."""
  rng = np.random.RandomState(22)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  x = H.int_tensor(rng, (1, rows, Ci), -3, 3)
  W = H.int_tensor(rng, (Ci, Co), -2, 2, 0.25)
  b = H.int_tensor(rng, (Co,), -2, 2, 0.5)
  op = H.pack(W.to(H.DEV), [(0, 1, 0, Co, 1)], Ci, Co, cip, 32, 1)
  xd = H.to_pitch(x, cip)
  bd = b.to(H.DEV)
  ref = x[0] @ W
  if Ci > 128 or Co > 128:  # (else cg_dense_rows takes its register form)
    cf = (Co + 7) // 8 * 8
    y = torch.full((rows, cf), 9.0, dtype=torch.float32, device=H.DEV)
    _lib.call('cg_dense_rows', H.p(xd), H.p(op.buf), H.p(bd), H.p(y), rows, cip,
              Co, cf, epi, H.stream())
    H.sync()
    got = y.cpu()
    if epi == 3:
      np.testing.assert_allclose(got[:, :Co].numpy(),
                                 torch.sigmoid(ref + b).numpy(), rtol=2e-6,
                                 atol=1e-6)
    else:
      np.testing.assert_array_equal(got[:, :Co].numpy(), (ref + b).numpy())
    if cf > Co:
      assert float(got[:, Co:].abs().max()) == 0.0
  # activation-typed output, no bias: every value below is exact in bf16
  ya = torch.full((rows, cop), 9.0, dtype=BF16, device=H.DEV)
  _lib.call('cg_dense_rows_act', H.p(xd), H.p(op.buf), H.p(ya), rows, cip, Co,
            cop, H.stream())
  H.sync()
  gota = ya.float().cpu()
  np.testing.assert_array_equal(gota[:, :Co].numpy(),
                                ref.to(BF16).float().numpy())
  if cop > Co:
    assert float(gota[:, Co:].abs().max()) == 0.0


@pytest.mark.parametrize('rows,Ci,Co', [(4096, 102, 102), (70, 32, 256),
                                        (33000, 512, 512), (1000, 130, 40)])
def test_dense_wgrad_streaming(rows, Ci, Co):
  """cg_dense_wgrad: dW (+)= x^T g for the per-timestep Dense, exact on integer
  data whatever the number of 128 x 128 tiles and row ranges; ragged row count.
  Without a workspace the partial tiles meet through f32 atomics and accumulate
  into dW; with one they are added in a fixed order and dW is stored."""
  rng = np.random.RandomState(23)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  x = H.int_tensor(rng, (1, rows, Ci), -2, 2)
  g = H.int_tensor(rng, (1, rows, Co), -2, 2)
  xd, gd = H.to_pitch(x, cip), H.to_pitch(g, cop)
  ref = x[0].double().t() @ g[0].double()
  need = _lib.load().cg_dense_wgrad_ws_elems(rows, Ci, Co)
  assert need > 0
  ws = torch.full((need,), float('nan'), device=H.DEV)
  for use_ws in (False, True):
    dw = torch.full((Ci, Co), 3.0, dtype=torch.float32, device=H.DEV)
    _lib.call('cg_dense_wgrad', H.p(xd), H.p(gd), H.p(dw), rows, cip, cop, Ci,
              Co, H.p(ws) if use_ws else None, need if use_ws else 0, H.stream())
    H.sync()
    # (+= onto the 3.0 without a workspace, = with one)
    np.testing.assert_array_equal(dw.cpu().double().numpy(),
                                  ref.numpy() + (0.0 if use_ws else 3.0))


WGRAD_CASES = [
    (3, 128, 102, 64, 24, 1, True),
    (2, 64, 16, 40, 24, 2, False),
    (4, 16, 64, 72, 8, 2, True),
    (6, 256, 64, 128, 24, 2, True),
    (96, 8, 128, 160, 24, 32, True),
    # ring-staged sweep: many tiles per workgroup (the ring wraps), interior and
    # edge tiles, a channel pitch that is not a whole 64-column chunk
    (5, 1024, 40, 72, 24, 2, True),
    (9, 512, 64, 64, 24, 3, True),
    (70, 128, 32, 64, 24, 2, True),
]


def _wgrad_test_desc(x, g, dw, nB, L, Ci, Co, k, **kw):
  return nets._wgrad_desc(x, g, dw, nB, L, geo.pitch(Ci), L // 2, geo.pitch(Co), k,
                          2, -geo.same_padding_left(k, 2), Ci, Co, **kw)


def _wgrad_cases():
  """(case, partials, classic); the partial-sum form only where three K' splits
  exist at the shape."""
  out = []
  for case in WGRAD_CASES:
    nB, L, Ci, Co, k, seg, use_shift = case
    for partials in (False, 'add', 'store'):
      for classic in (0, 1):
        if partials:
          d = _wgrad_test_desc(_PH, _PH, _PH, nB, L, Ci, Co, k, seg_size=seg)
          d.classic_staging, d.nsplit = classic, 3
          if _lib.load().cg_wgrad_partials_elems(ctypes.byref(d)) <= 0:
            continue
        out.append(case + (partials, classic))
  return out


@pytest.mark.parametrize('nB,L,Ci,Co,k,seg,use_shift,partials,classic',
                         _wgrad_cases())
def test_conv_wgrad_bitexact(nB, L, Ci, Co, k, seg, use_shift, partials, classic):
  """partials: the K' splits store partial sums that a second launch adds into
  dw -- 'add': dw ACCUMULATES (it starts at 1), 'store': dw and dbias are stored
  (cg_wgrad_desc.store, the ordered mode's form); False: f32 atomics.  (Until
  round 5 the partial-sum cases were skipped by mistake: the size query answered
  an error for descriptors with `store` set, and the test took that for "one K'
  split" -- found when the skips were turned into collection-time checks.)
  classic = 1: register-staged tiles instead of the LDS-DMA ring (the 24-tap
  cases whose samples span whole tiles take the ring by default)."""
  rng = np.random.RandomState(7)
  x = H.int_tensor(rng, (nB, L, Ci), -2, 2)
  dy = H.int_tensor(rng, (nB, L // 2, Co), -2, 2)
  nseg = (nB + seg - 1) // seg
  shifts = rng.randint(-2, 3, size=nseg).astype(np.int32)
  if not use_shift:
    shifts[:] = 0
  W = torch.zeros(k, Ci, Co, requires_grad=True)
  (O.conv1d_same(_shuffle_batch(x, shifts, seg), W, None, 2) *
   dy).sum().backward()
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  dw = torch.ones(k, Ci, Co, dtype=torch.float32, device=H.DEV)
  sh = torch.tensor(shifts, device=H.DEV)
  dbias = torch.zeros(Co, dtype=torch.float32, device=H.DEV)
  nb_bias = max(1, (2 * nB) // 3)  # bias gradient over the first samples only
  d = nets._wgrad_desc(H.to_pitch(x, cip), H.to_pitch(dy, cop), dw, nB, L, cip,
                       L // 2, cop, k, 2, -geo.same_padding_left(k, 2), Ci, Co,
                       shifts=sh if use_shift else None, seg_size=seg,
                       dbias=dbias, bias_rows=nb_bias * (L // 2),
                       slot=0 if partials else None)
  d.classic_staging = classic
  if partials:
    d.nsplit = 3  # several K' splits whatever the shape
    need = _lib.load().cg_wgrad_partials_elems(ctypes.byref(d))
    assert need > 0
    ws = torch.full((need,), float('nan'), device=H.DEV)
    d.partials, d.partials_elems = ws.data_ptr(), need
    d.store = int(partials == 'store')
    if d.store:
      dbias.fill_(5.0)
  H.run_wgrad(d)
  H.sync()
  np.testing.assert_array_equal(dw.cpu().numpy(),
                                W.grad.numpy() + (0.0 if partials == 'store' else 1.0))
  np.testing.assert_array_equal(dbias.cpu().numpy(),
                                dy[:nb_bias].sum((0, 1)).numpy())


def test_conv_transpose_wgrad_bitexact():
  rng = np.random.RandomState(8)
  B, L, Ci, Co, k = 3, 64, 32, 72, 24
  x = H.int_tensor(rng, (B, L, Ci), -2, 2)
  dy = H.int_tensor(rng, (B, 2 * L, Co), -2, 2)
  Wt = torch.zeros(k, 1, Co, Ci, requires_grad=True)
  (O.conv1d_transpose_same(x, Wt, None, 2) * dy).sum().backward()
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  dw = torch.zeros(k, 1, Co, Ci, dtype=torch.float32, device=H.DEV)
  d = nets._wgrad_desc(H.to_pitch(dy, cop), H.to_pitch(x, cip), dw, B, 2 * L,
                       cop, L, cip, k, 2, -11, Co, Ci)
  H.run_wgrad(d)
  H.sync()
  np.testing.assert_array_equal(dw.cpu().numpy(), Wt.grad.numpy())


@pytest.mark.parametrize('B,L,Ci,Co', [(2, 512, 102, 102), (70, 1, 32, 256)])
def test_dense_wgrad_bitexact(B, L, Ci, Co):
  rng = np.random.RandomState(9)
  x = H.int_tensor(rng, (B, L, Ci), -2, 2)
  dy = H.int_tensor(rng, (B, L, Co), -2, 2)
  ref = torch.einsum('blc,bld->cd', x, dy)
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  dw = torch.zeros(Ci, Co, dtype=torch.float32, device=H.DEV)
  d = nets._wgrad_desc(H.to_pitch(x, cip), H.to_pitch(dy, cop), dw, B, L, cip,
                       L, cop, 1, 1, 0, Ci, Co)
  H.run_wgrad(d)
  H.sync()
  np.testing.assert_array_equal(dw.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize('rows,C', [(300, 102), (64, 320), (17, 40),
                                    # pitches whose 8-channel groups are not a power
                                    # of two: the 8-lanes-per-row forward (round 5)
                                    (1000, 192), (33, 400), (129, 224), (70, 96),
                                    (4099, 320)])
def test_layernorm_lrelu_fwd_bwd(rows, C):
  rng = np.random.RandomState(10)
  cp = geo.pitch(C)
  y = torch.tensor(rng.randn(1, rows, C).astype(np.float32) * 2 + 0.5)
  gam = torch.tensor(rng.rand(C).astype(np.float32) + 0.5)
  bet = torch.tensor(rng.randn(C).astype(np.float32) * 0.1)
  dh = torch.tensor(rng.randn(1, rows, C).astype(np.float32))
  yq = y.to(BF16).float().requires_grad_(True)
  g_ = gam.clone().requires_grad_(True)
  b_ = bet.clone().requires_grad_(True)
  h_ref = O.leaky_relu(O.layer_norm(yq, g_, b_))
  dhq = dh.to(BF16).float()
  yd, dhd = H.to_pitch(y, cp), H.to_pitch(dh, cp)
  h = torch.zeros(1, rows, cp, dtype=BF16, device=H.DEV)
  mean = torch.zeros(rows, device=H.DEV)
  rstd = torch.zeros(rows, device=H.DEV)
  gd, bd = gam.to(H.DEV), bet.to(H.DEV)
  _lib.call('cg_ln_lrelu_fwd', H.p(yd), H.p(gd), H.p(bd), H.p(h), H.p(mean),
            H.p(rstd), rows, C, cp, 1e-3, ALPHA, H.stream())
  H.sync()
  np.testing.assert_allclose(
      h.float().cpu()[0, :, :C].numpy(), h_ref.detach()[0].numpy(), rtol=1e-2,
      atol=1e-2)
  if cp > C:
    assert float(h[:, :, C:].float().abs().max()) == 0.0
  # backward uses the bf16 h for the mask, as the product path does
  hq = h.float().cpu()[:, :, :C]
  mask = torch.where(hq > 0, 1.0, ALPHA)
  (O.layer_norm(yq, g_, b_) * (dhq * mask)).sum().backward()
  # ws None: f32 atomics into zeroed outputs; else the ordered reduction, which
  # stores (outputs start poisoned) and repeats bit for bit
  for ws in (None, H.reduce_ws()):
    dy = torch.zeros(1, rows, cp, dtype=BF16, device=H.DEV)
    dg, db, dbias = H.out_buffers(ws, C, C, C)
    _lib.call('cg_ln_lrelu_bwd', H.p(dhd), H.p(h), H.p(yd), H.p(mean),
              H.p(rstd), H.p(gd), H.p(dy), H.p(dg), H.p(db), H.p(dbias), rows, C,
              cp, ALPHA, H.p(ws), H.stream())
    H.sync()
    np.testing.assert_allclose(
        dy.float().cpu()[0, :, :C].numpy(), yq.grad[0].numpy(), rtol=2e-2,
        atol=2e-2)
    np.testing.assert_allclose(dg.cpu().numpy(), g_.grad.numpy(), rtol=1e-3,
                               atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), b_.grad.numpy(), rtol=1e-3,
                               atol=1e-3)
    # fused bias gradient of the producing conv = column sums of the stored dy
    np.testing.assert_allclose(dbias.cpu().numpy(),
                               dy.float().cpu()[0, :, :C].sum(0).numpy(),
                               rtol=1e-4, atol=1e-4)
    if ws is not None:
      first = [t.clone() for t in (dg, db, dbias)]
      _lib.call('cg_ln_lrelu_bwd', H.p(dhd), H.p(h), H.p(yd), H.p(mean),
                H.p(rstd), H.p(gd), H.p(dy), H.p(dg), H.p(db), H.p(dbias), rows,
                C, cp, ALPHA, H.p(ws), H.stream())
      H.sync()
      for a, b in zip(first, (dg, db, dbias)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('nB,L,Ci,Co,k,seg,m', [(6, 128, 102, 64, 24, 2, 10),
                                                (5, 32, 64, 128, 24, 1, 3),
                                                (4, 512, 32, 64, 8, 4, 2),
                                                (3, 64, 16, 40, 24, 1, 0)])
def test_dgrad_with_fused_unshuffle(nB, L, Ci, Co, k, seg, m, epi_mode):
  """Input gradient of a strided conv whose input was phase-shuffled, with the
  shuffle adjoint + LeakyReLU' mask in the launch's epilogue
  (cg_conv_desc.out_shifts + cg_unshuffle_fixup) == the plain launch followed
  by cg_unshuffle_mask: identical on every row that receives one contribution,
  one bf16 rounding apart on the <= m rows per sample that receive two."""
  rng = np.random.RandomState(41)
  W = H.int_tensor(rng, (k, Ci, Co), -1, 1, 0.5)
  dy = H.int_tensor(rng, (nB, L // 2, Co), -2, 2)
  h = torch.tensor(rng.randn(nB, L, Ci).astype(np.float32))
  nseg = (nB + seg - 1) // seg
  shifts = rng.randint(-m, m + 1, size=nseg).astype(np.int32)
  if m:
    shifts[0], shifts[-1] = m, -m
  cip, cop = geo.pitch(Ci), geo.pitch(Co)
  pl = geo.same_padding_left(k, 2)
  phases = nets._transpose_phases(k, pl)
  offs = [o for _, o in phases]
  ck = nets._ck_for(cop, 1, k // 2, L // 2)
  op = H.pack(W.to(H.DEV), [(t0, -2, Ci * Co, 1, Co) for t0, _ in phases], Co,
              Ci, cop, ck, k // 2)
  dyd, hd = H.to_pitch(dy, cop), H.to_pitch(h, cip)
  sh = torch.tensor(shifts, device=H.DEV)
  common = dict(y_stride=2, y_off=0, nphase=2, w_phase_stride=op.elems,
                off_phase_step=offs[1] - offs[0], yoff_phase_step=1)
  z = lambda *s_: torch.zeros(*s_, dtype=BF16, device=H.DEV)
  e, d0 = z(nB, L, cip), z(nB, L, cip)
  da = H.conv_desc(dyd, op.buf, e, nB, L // 2, cop, k // 2, 1, offs[0], L // 2,
                   Ci, L, cip, ck, **common)
  H.run_conv(da)
  _lib.call('cg_unshuffle_mask', H.p(e), H.p(hd), H.p(d0), H.p(sh), nB, L, cip,
            seg, ALPHA, H.stream())
  sr = max(1, m)
  side = z(nB, sr, cip)
  d1 = torch.full((nB, L, cip), 9.0, dtype=BF16, device=H.DEV)
  db = H.conv_desc(dyd, op.buf, d1, nB, L // 2, cop, k // 2, 1, offs[0], L // 2,
                   Ci, L, cip, ck, mask_src=hd, epilogue=_lib.EPI_MASK,
                   out_shifts=(sh, seg, side, sr), **common)
  H.run_conv(db)
  _lib.call('cg_unshuffle_fixup', H.p(side), H.p(hd), H.p(d1), H.p(sh), nB, L,
            cip, seg, sr, ALPHA, H.stream())
  H.sync()
  a, b = d0.float().cpu().numpy(), d1.float().cpu().numpy()
  twice = np.zeros((nB, L), bool)  # rows with a direct and a reflected source
  for i in range(nB):
    s_ = int(shifts[i // seg])
    if s_ > 0:
      twice[i, L - 1 - s_:L - 1] = True
    elif s_ < 0:
      twice[i, 1:-s_ + 1] = True
  np.testing.assert_array_equal(b[~twice], a[~twice])
  # (the first contribution is rounded to bf16 on its own: half an ulp of it)
  emax = float(e.float().abs().max())
  np.testing.assert_allclose(b[twice], a[twice], rtol=2 ** -7,
                             atol=2 ** -8 * emax)
  assert float(np.abs(a).max()) > 0


@pytest.mark.parametrize('w,m', [(16, 3), (64, 10), (8, 1)])
def test_unshuffle_mask_is_adjoint_of_shuffle(w, m):
  rng = np.random.RandomState(11)
  nB, C, seg = 6, 40, 2
  e = H.int_tensor(rng, (nB, w, C), -3, 3)
  h = torch.tensor(rng.randn(nB, w, C).astype(np.float32))
  shifts = rng.randint(-m, m + 1, size=3).astype(np.int32)
  hx = h.to(BF16).float().requires_grad_(True)
  # forward: pre -> lrelu -> shuffle; adjoint applied to e
  pre = hx
  out = _shuffle_batch(O.leaky_relu(pre), shifts, seg)
  (out * e).sum().backward()
  cp = geo.pitch(C)
  ed = H.to_pitch(e, cp)
  # mask source: the POST-activation (same sign as pre)
  hd = H.to_pitch(O.leaky_relu(hx.detach()), cp)
  delta = torch.zeros(nB, w, cp, dtype=BF16, device=H.DEV)
  sh = torch.tensor(shifts, device=H.DEV)
  _lib.call('cg_unshuffle_mask', H.p(ed), H.p(hd), H.p(delta), H.p(sh), nB, w,
            cp, seg, ALPHA, H.stream())
  H.sync()
  np.testing.assert_allclose(
      delta.float().cpu()[:, :, :C].numpy(),
      hx.grad.to(BF16).float().numpy(), rtol=0, atol=0)


def test_discriminator_head_kernels():
  rng = np.random.RandomState(12)
  nB, Lt, C, seg = 6, 8, 40, 2
  cp = geo.pitch(C)
  h = torch.tensor(rng.randn(nB, Lt, C).astype(np.float32)).to(BF16).float()
  w = torch.tensor(rng.randn(Lt * C).astype(np.float32))
  wq = w.to(BF16).float()
  b = torch.tensor([0.25])
  coef = torch.tensor([-0.5, 0.5, 1.0])
  hd = H.to_pitch(h, cp)
  wd, bd, cd = w.to(H.DEV), b.to(H.DEV), coef.to(H.DEV)
  out = torch.zeros(nB, device=H.DEV)
  _lib.call('cg_dense1_fwd', H.p(hd), H.p(wd), H.p(bd), H.p(out), nB, Lt, C, cp,
            H.stream())
  ref = h.reshape(nB, -1) @ wq + 0.25
  H.sync()
  np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5,
                             atol=1e-5)
  delta = torch.zeros(nB, Lt, cp, dtype=BF16, device=H.DEV)
  _lib.call('cg_dense1_bwd', H.p(wd), H.p(cd), H.p(hd), H.p(delta), nB, Lt, C,
            cp, seg, ALPHA, H.stream())
  H.sync()
  cb = coef.repeat_interleave(seg).reshape(nB, 1, 1)
  refd = cb * wq.reshape(1, Lt, C) * torch.where(h > 0, 1.0, ALPHA)
  np.testing.assert_array_equal(
      delta.float().cpu()[:, :, :C].numpy(), refd.to(BF16).float().numpy())
  # both in one pass over h (what the step launches): the same bits
  out2 = torch.zeros(nB, device=H.DEV)
  delta2 = torch.full((nB, Lt, cp), 7.0, dtype=BF16, device=H.DEV)
  _lib.call('cg_dense1_fwd_bwd', H.p(hd), H.p(wd), H.p(bd), H.p(out2), H.p(cd),
            H.p(delta2), nB, Lt, C, cp, seg, ALPHA, H.stream())
  H.sync()
  assert torch.equal(out2, out) and torch.equal(delta2, delta)
  bc = torch.tensor([-0.5, 0.5, 0.0]).to(H.DEV)
  refw = (cb * h).sum(0).reshape(-1)
  for ws in (None, H.reduce_ws()):
    dw, db = H.out_buffers(ws, Lt * C, 1)
    _lib.call('cg_dense1_wgrad', H.p(hd), H.p(cd), H.p(bc), H.p(dw), H.p(db), nB,
              Lt, C, cp, seg, H.p(ws), H.stream())
    H.sync()
    np.testing.assert_allclose(dw.cpu().numpy(), refw.numpy(), rtol=1e-5,
                               atol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), [0.0], atol=1e-6)


@pytest.mark.parametrize('slope', [0.25, 0.2, 1.0])
def test_lrelu_mix_inverts_the_activation_and_interpolates(slope):
  """cg_lrelu_mix: out = act(a act^-1(h_a) + (1 - a) act^-1(h_b)), act(y) = max(y,
  slope y) -- the critic's first layer on x^ from its outputs on real and fake
  (round 5).  Exact where the arithmetic is (slope 1/4, mixing factors in {0, 1/4,
  1/2, 1}, small integers); within one bf16 ulp of the f64 formula on random data;
  slopes outside (0, 1] are refused (the activation is then not invertible / not
  increasing on both branches)."""
  rng = np.random.RandomState(23)
  B, n = 5, 64 * 40
  if slope == 0.25:
    ya = rng.randint(-32, 33, (B, n)).astype(np.float32) * 4
    yb = rng.randint(-32, 33, (B, n)).astype(np.float32) * 4
    mix = np.array([0.0, 0.25, 0.5, 1.0, 0.75], np.float32)
  else:
    ya = rng.randn(B, n).astype(np.float32)
    yb = rng.randn(B, n).astype(np.float32)
    mix = rng.rand(B).astype(np.float32)
  act = lambda y: np.maximum(y, slope * y)
  ha = torch.tensor(act(ya)).to(BF16)
  hb = torch.tensor(act(yb)).to(BF16)
  out = torch.zeros(B, n, dtype=BF16, device=H.DEV)
  ha_d, hb_d, mix_d = ha.to(H.DEV), hb.to(H.DEV), torch.tensor(mix).to(H.DEV)
  _lib.call('cg_lrelu_mix', H.p(ha_d), H.p(hb_d), H.p(mix_d), H.p(out), B, n,
            slope, H.stream())
  H.sync()
  inv = lambda h: np.where(h > 0, h, h / slope)
  pa, pb = inv(ha.float().numpy().astype(np.float64)), inv(hb.float().numpy().astype(np.float64))
  want = act(mix[:, None].astype(np.float64) * pa + (1 - mix[:, None].astype(np.float64)) * pb)
  got = out.float().cpu().numpy()
  if slope == 0.25:
    np.testing.assert_array_equal(got, want.astype(np.float32))
  else:
    np.testing.assert_allclose(got, want, rtol=2.0**-7, atol=1e-30)
  lib = _lib.load()
  for bad in (0.0, -0.1, 1.5):
    assert lib.cg_lrelu_mix(H.p(out), H.p(out), H.p(out), H.p(out), B, n,
                            ctypes.c_float(bad), H.stream()) != 0
  assert lib.cg_lrelu_mix(H.p(out), H.p(out), H.p(out), H.p(out), B, 12,
                          ctypes.c_float(0.2), H.stream()) != 0


def test_wgan_gp_elementwise_kernels():
  rng = np.random.RandomState(13)
  B, L, C = 4, 32, 102
  cp = geo.pitch(C)
  real = torch.tensor(rng.rand(B, L, C).astype(np.float32))
  fake_p = torch.zeros(B, L, cp)
  fake = torch.tensor(rng.rand(B, L, C).astype(np.float32))
  fake_p[:, :, :C] = fake
  alpha = torch.tensor(rng.rand(B).astype(np.float32))
  x0 = torch.zeros(3 * B, L, cp, dtype=BF16, device=H.DEV)
  real_d, fake_d, alpha_d = real.to(H.DEV), fake_p.to(H.DEV), alpha.to(H.DEV)
  _lib.call('cg_interp_pack', H.p(real_d), H.p(fake_d), H.p(alpha_d), H.p(x0),
            B, L, C, C, cp, cp, 1, H.stream())
  H.sync()
  got = x0.float().cpu()
  inter = O.interpolation(real, fake, alpha)
  np.testing.assert_array_equal(got[:B, :, :C].numpy(),
                                real.to(BF16).float().numpy())
  np.testing.assert_array_equal(got[B:2 * B, :, :C].numpy(),
                                fake.to(BF16).float().numpy())
  np.testing.assert_allclose(got[2 * B:, :, :C].numpy(),
                             inter.to(BF16).float().numpy(), rtol=8e-3)
  assert float(got[:, :, C:].abs().max()) == 0.0
  # penalty norm / finalize / scale
  g = torch.tensor(rng.randn(B, L * cp).astype(np.float32) * 0.05)
  g = g.to(BF16).float()  # the input gradient is stored in bf16
  gd = g.to(H.DEV).to(BF16)
  norm = torch.zeros(B, device=H.DEV)
  gp = torch.zeros(1, device=H.DEV)
  coef = torch.zeros(B, device=H.DEV)
  (norm_o,) = H.out_buffers(True, B)
  _lib.call('cg_rownorm', H.p(gd), H.p(norm_o), B, L * cp, H.p(H.reduce_ws()),
            H.stream())
  _lib.call('cg_rownorm', H.p(gd), H.p(norm), B, L * cp, None, H.stream())
  H.sync()
  np.testing.assert_allclose(norm_o.cpu().numpy(), norm.cpu().numpy(), rtol=1e-6)
  _lib.call('cg_gp_finalize', H.p(norm), H.p(gp), H.p(coef), B, 10.0, 0,
            H.stream())
  a0 = torch.zeros(B, L * cp, dtype=BF16, device=H.DEV)
  _lib.call('cg_scale_rows', H.p(gd), H.p(coef), H.p(a0), B, L * cp, H.stream())
  H.sync()
  gr = g.clone().requires_grad_(True)
  nr = gr.pow(2).sum(1).sqrt()
  gpr = ((nr - 1)**2).mean()
  (10.0 * gpr).backward()
  np.testing.assert_allclose(norm.cpu().numpy(), nr.detach().numpy(), rtol=1e-5)
  np.testing.assert_allclose(gp.cpu().numpy(), [gpr.item()], rtol=1e-5)
  np.testing.assert_allclose(a0.float().cpu().numpy(), gr.grad.numpy(),
                             rtol=1e-2, atol=1e-6)
  # losses
  d_out = torch.tensor(rng.randn(3 * B).astype(np.float32)).to(H.DEV)
  out = torch.zeros(2, device=H.DEV)
  _lib.call('cg_critic_loss', H.p(d_out), H.p(gp), 10.0, H.p(out), B, H.stream())
  H.sync()
  do = d_out.cpu()
  exp0 = -do[:B].mean() + do[B:2 * B].mean() + 10.0 * gp.cpu()[0]
  np.testing.assert_allclose(out.cpu().numpy(),
                             [exp0.item(), -do[B:2 * B].mean().item()],
                             rtol=1e-5)
  # cg_gp_finalize + cg_critic_loss as the one launch the step uses: from the
  # sums of squares (squared = 1), coef scaled by 2
  nsq = (norm * norm).clone()
  gp2, coef2, out2 = (torch.zeros(1, device=H.DEV), torch.zeros(B, device=H.DEV),
                      torch.zeros(2, device=H.DEV))
  _lib.call('cg_gp_critic_loss', H.p(nsq), H.p(gp2), H.p(coef2), H.p(d_out),
            H.p(out2), B, 10.0, 1, 2.0, H.stream())
  H.sync()
  np.testing.assert_allclose(nsq.cpu().numpy(), norm.cpu().numpy(), rtol=1e-6)
  np.testing.assert_allclose(gp2.cpu().numpy(), gp.cpu().numpy(), rtol=1e-5)
  np.testing.assert_allclose(coef2.cpu().numpy(), 2.0 * coef.cpu().numpy(),
                             rtol=1e-5)
  np.testing.assert_allclose(out2.cpu().numpy(), out.cpu().numpy(), rtol=1e-5)
  # ... and with the penalty norm's slot sums and v = coef_b * g in the same
  # launch (cg_gp_loss_scale): equal, bit for bit, to slots summed in order ->
  # cg_gp_critic_loss -> cg_scale_rows
  P = 5
  slots = torch.tensor(rng.rand(B, P).astype(np.float32) * 3.0).to(H.DEV)
  ssum = torch.zeros(B, device=H.DEV)
  for j in range(P):
    ssum = ssum + slots[:, j]  # slot order
  gp3, coef3, out3 = (torch.zeros(1, device=H.DEV), torch.zeros(B, device=H.DEV),
                      torch.zeros(2, device=H.DEV))
  nsq3 = ssum.clone()
  _lib.call('cg_gp_critic_loss', H.p(nsq3), H.p(gp3), H.p(coef3), H.p(d_out),
            H.p(out3), B, 10.0, 1, 1.0, H.stream())
  a3 = torch.zeros(B, L * cp, dtype=BF16, device=H.DEV)
  _lib.call('cg_scale_rows', H.p(gd), H.p(coef3), H.p(a3), B, L * cp, H.stream())
  norm4, gp4, coef4, out4 = (torch.full((B,), 7.0, device=H.DEV),
                             torch.zeros(1, device=H.DEV),
                             torch.zeros(B, device=H.DEV),
                             torch.zeros(2, device=H.DEV))
  a4 = torch.zeros(B, L * cp, dtype=BF16, device=H.DEV)
  _lib.call('cg_gp_loss_scale', H.p(slots), P, H.p(norm4), H.p(gp4), H.p(coef4),
            H.p(d_out), H.p(out4), B, 10.0, 1.0, H.p(gd), H.p(a4), L * cp,
            H.stream())
  H.sync()
  for x, y in ((norm4, nsq3), (gp4, gp3), (coef4, coef3), (out4, out3)):
    assert torch.equal(x, y), (x, y)
  assert torch.equal(a4.view(torch.int16), a3.view(torch.int16))
  # without rows to scale (validate()): the reductions alone
  norm5 = torch.zeros(B, device=H.DEV)
  gp5 = torch.zeros(1, device=H.DEV)
  _lib.call('cg_gp_loss_scale', H.p(slots), P, H.p(norm5), H.p(gp5), H.p(coef4),
            H.p(d_out), H.p(out4), B, 10.0, 1.0, None, None, 0, H.stream())
  H.sync()
  assert torch.equal(norm5, nsq3) and torch.equal(gp5, gp3)


def test_adam_colsum_sigmoid_lrelu_metrics():
  rng = np.random.RandomState(14)
  n = 10007
  p0 = torch.tensor(rng.randn(n).astype(np.float32))
  g = torch.tensor(rng.randn(n).astype(np.float32))
  m0 = torch.tensor(rng.randn(n).astype(np.float32) * 0.1)
  v0 = torch.tensor(rng.rand(n).astype(np.float32) * 0.1)
  pr, mr, vr = p0.double(), m0.double(), v0.double()
  O.keras_adam(pr, g.double() * 0.5, mr, vr, 3, 1e-3)
  pd, gd, md, vd = (t.clone().to(H.DEV) for t in (p0, g, m0, v0))
  import math
  lr_t = 1e-3 * math.sqrt(1 - 0.999**3) / (1 - 0.9**3)
  _lib.call('cg_adam', H.p(pd), H.p(gd), H.p(md), H.p(vd), n, lr_t, 0.9, 0.999,
            1e-7, 0.5, None, H.stream())
  H.sync()
  np.testing.assert_allclose(pd.cpu().numpy(), pr.float().numpy(), rtol=1e-6,
                             atol=1e-7)
  np.testing.assert_allclose(md.cpu().numpy(), mr.float().numpy(), rtol=1e-6,
                             atol=1e-7)
  np.testing.assert_allclose(vd.cpu().numpy(), vr.float().numpy(), rtol=1e-6,
                             atol=1e-7)
  # colsum
  rows, C = 1000, 102
  cp = geo.pitch(C)
  x = H.int_tensor(rng, (1, rows, C), -3, 3)
  xd = H.to_pitch(x, cp)
  for ws in (None, H.reduce_ws()):
    (out,) = H.out_buffers(ws, C)
    _lib.call('cg_colsum', H.p(xd), H.p(out), rows, C, cp, H.p(ws), H.stream())
    H.sync()
    np.testing.assert_array_equal(out.cpu().numpy(), x[0].sum(0).numpy())
  # rows wider than one 2048-channel slab (the generator's input Dense bias
  # gradient at sequence length 8192: 256 * 32 columns), and an odd tail slab
  for rows_w, cw in ((37, 8192), (300, 2080)):
    xw = H.int_tensor(rng, (1, rows_w, cw), -3, 3)
    for ws in (None, H.reduce_ws()):
      (outw,) = H.out_buffers(ws, cw)
      _lib.call('cg_colsum', H.p(H.to_pitch(xw, cw)), H.p(outw), rows_w, cw, cw,
                H.p(ws), H.stream())
      H.sync()
      np.testing.assert_array_equal(outw.cpu().numpy(), xw[0].sum(0).numpy())
  # sigmoid bwd
  fake = torch.tensor(rng.rand(1, rows, C).astype(np.float32))
  dfake = torch.zeros(1, rows, cp)
  dfake[:, :, :C] = torch.tensor(rng.randn(1, rows, C).astype(np.float32))
  dz = torch.zeros(1, rows, cp, dtype=BF16, device=H.DEV)
  dfake = dfake.to(BF16).float()  # arrives as a bf16 activation gradient
  dfake_d, fake_d = dfake.to(H.DEV).to(BF16), fake.to(H.DEV)
  _lib.call('cg_sigmoid_bwd', H.p(dfake_d), H.p(fake_d), H.p(dz), rows, C, C,
            cp, H.stream())
  H.sync()
  ref = dfake[:, :, :C] * fake * (1 - fake)
  np.testing.assert_allclose(dz.float().cpu()[:, :, :C].numpy(),
                             ref.to(BF16).float().numpy(), rtol=8e-3,
                             atol=1e-7)
  # metrics
  real = torch.tensor(rng.rand(3, 50, C).astype(np.float32))
  fk = torch.tensor(rng.rand(3, 50, C).astype(np.float32))
  real_d, fk_d = real.to(H.DEV), fk.to(H.DEV)
  refm = O.signal_metrics(real, fk, -1.0, 3.0, True)
  exp = [refm['signals_metrics/' + k].item() for k in ('min', 'max', 'mean',
                                                        'std')]
  for ws in (None, H.reduce_ws()):
    (buf,) = H.out_buffers(ws, 4)
    _lib.call('cg_signal_metrics', H.p(real_d), H.p(fk_d), H.p(buf), 150, C, C,
              C, -1.0, 3.0, H.p(ws), H.stream())
    H.sync()
    # (sums without a workspace; the ordered form stores the means)
    got = buf.cpu().numpy() / (150 if ws is None else 1)
    np.testing.assert_allclose(got, exp, rtol=1e-4)


@pytest.fixture(params=[0, 2], ids=['split_forms', 'flex'])
def wgrad_form(request):
  """cg_wgrad_batched's K'-split forms (plain / halves) and the flex form
  (forced: the small test shapes would otherwise take the split forms)."""
  was = _lib.load().cg_debug_wgrad_flex(request.param)
  yield request.param
  _lib.load().cg_debug_wgrad_flex(was)


def test_wgrad_batched_equals_individual_launches(wgrad_form):
  """cg_wgrad_batched (one launch over several layers) accumulates exactly
  what the per-layer launches do; a batch that cannot be fused (a 1-tap Dense
  gradient among them) falls back to individual launches."""
  rng = np.random.RandomState(31)
  layers = [(4, 512, 64, 128), (4, 256, 128, 192), (4, 128, 192, 64)]
  k = 24
  descs_a, descs_b, outs_a, outs_b, keep = [], [], [], [], []
  for nB, L, Ci, Co in layers:
    x = H.to_pitch(H.int_tensor(rng, (nB, L, Ci), -2, 2), geo.pitch(Ci))
    g = H.to_pitch(H.int_tensor(rng, (nB, L // 2, Co), -2, 2), geo.pitch(Co))
    sh = torch.tensor(rng.randint(-3, 4, size=2).astype(np.int32), device=H.DEV)
    keep += [x, g, sh]
    for descs, outs in ((descs_a, outs_a), (descs_b, outs_b)):
      dw = torch.zeros(k, Ci, Co, dtype=torch.float32, device=H.DEV)
      db = torch.zeros(Co, dtype=torch.float32, device=H.DEV)
      d = nets._wgrad_desc(x, g, dw, nB, L, geo.pitch(Ci), L // 2,
                           geo.pitch(Co), k, 2, -geo.same_padding_left(k, 2),
                           Ci, Co, shifts=sh, seg_size=2, dbias=db,
                           bias_rows=3 * (L // 2),
                           slot=len(descs) if descs is descs_b else None)
      descs.append(d)
      outs.append((dw, db))
  for d in descs_a:
    H.run_wgrad(d)
  arr = (_lib.WgradDesc * len(descs_b))(*descs_b)
  # (the flex form plans at this size only when forced: mode 2)
  planned = _lib.load().cg_wgrad_flex_plan(arr, len(descs_b), 2, None, 0, None) > 0
  assert planned
  _lib.call('cg_wgrad_batched', arr, len(descs_b), H.stream())
  H.sync()
  for (dwa, dba), (dwb, dbb) in zip(outs_a, outs_b):
    assert float(dwa.abs().max()) > 0
    assert torch.equal(dwa, dwb) and torch.equal(dba, dbb)
  # mixed batch: falls back, same results
  B, L, Ci, Co = 2, 512, 102, 102
  x = H.to_pitch(H.int_tensor(rng, (B, L, Ci), -2, 2), geo.pitch(Ci))
  g = H.to_pitch(H.int_tensor(rng, (B, L, Co), -2, 2), geo.pitch(Co))
  dws = [torch.zeros(1, Ci, Co, dtype=torch.float32, device=H.DEV)
         for _ in range(2)]
  dense = [nets._wgrad_desc(x, g, dw, B, L, geo.pitch(Ci), L, geo.pitch(Co), 1,
                            1, 0, Ci, Co) for dw in dws]
  H.run_wgrad(dense[0])
  for dw, _ in outs_b:
    dw.zero_()
  mixed = [descs_b[0], dense[1], descs_b[1]]
  arr = (_lib.WgradDesc * 3)(*mixed)
  descs_b[0].dbias = None
  descs_b[1].dbias = None
  _lib.call('cg_wgrad_batched', arr, 3, H.stream())
  H.sync()
  assert torch.equal(dws[0], dws[1])
  assert torch.equal(outs_b[0][0], outs_a[0][0])
  assert torch.equal(outs_b[1][0], outs_a[1][0])


@pytest.mark.parametrize('rows,C,centre,spread', [(3000, 102, 50.0, 1.0),
                                                  (70000, 64, -200.0, 2.0),
                                                  (517, 320, 50.0, 50.0)])
def test_batchnorm_statistics_of_off_centre_channels(rows, C, centre, spread):
  """cg_bn_stats on channels whose mean is large against their spread:
  Keras' BatchNormalization takes tf.nn.moments, mean((y - mean)^2); a one-pass
  E[y^2] - E[y]^2 in f32 cancels there (|mean| / std = 50 .. 100: the round-4
  kernel was 1e-2 off in the variance and clamped negative results to 0).  The
  kernel sums around each block's own first row and combines the blocks' (count,
  mean, M2); checked against float64 moments of the same bf16-stored values."""
  rng = np.random.RandomState(14)
  cp = geo.pitch(C)
  offs = centre * (1.0 + 0.1 * rng.rand(C)).astype(np.float32)
  y = torch.tensor(rng.randn(1, rows, C).astype(np.float32) * spread + offs)
  yd = H.to_pitch(y, cp)
  yq = yd.float().cpu()[0, :, :C].double()
  mean_r = yq.mean(0)
  var_r = ((yq - mean_r)**2).mean(0)
  mean = torch.zeros(C, device=H.DEV)
  var = torch.zeros(C, device=H.DEV)
  mm = torch.zeros(C, device=H.DEV)
  mv = torch.ones(C, device=H.DEV)
  first = None
  for _ in range(2):
    _lib.call('cg_bn_stats', H.p(yd), rows, C, cp, H.p(mean), H.p(var), H.p(mm),
              H.p(mv), 0.99, H.p(H.reduce_ws()), H.stream())
    H.sync()
    if first is None:
      first = (mean.clone(), var.clone())
  assert torch.equal(first[0], mean) and torch.equal(first[1], var)  # ordered sums
  np.testing.assert_allclose(mean.cpu().numpy(), mean_r.numpy(), rtol=2e-6)
  np.testing.assert_allclose(var.cpu().numpy(), var_r.numpy(), rtol=1e-4)


@pytest.mark.parametrize('rows,C,act', [(3000, 102, 1), (517, 320, 0), (64, 16, 1)])
def test_batchnorm_fwd_bwd(rows, C, act):
  """cg_bn_stats / cg_bn_apply / cg_bn_bwd (layers.BatchNormalization, training
  mode: biased batch variance over the rows, eps 1e-3, moving averages with
  momentum 0.99) against torch autograd on the same bf16-stored values; the
  inference form (moving statistics) against the closed formula; the ordered
  column sums repeat bit for bit."""
  rng = np.random.RandomState(4)
  cp = geo.pitch(C)
  y = torch.tensor(rng.randn(1, rows, C).astype(np.float32) * 1.5 + 0.3)
  gam = torch.tensor(rng.uniform(0.5, 1.5, C).astype(np.float32))
  bet = torch.tensor(rng.randn(C).astype(np.float32) * 0.2)
  dh = torch.tensor(rng.randn(1, rows, C).astype(np.float32))
  alpha = ALPHA if act else 1.0
  yd, dhd = H.to_pitch(y, cp), H.to_pitch(dh, cp)
  yq = yd.float().cpu()[:, :, :C].clone().requires_grad_(True)
  g_, b_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
  mean_r = yq.mean(dim=(0, 1))
  var_r = ((yq - mean_r)**2).mean(dim=(0, 1))
  t = (yq - mean_r) * torch.rsqrt(var_r + 1e-3) * g_ + b_
  h_ref = torch.where(t > 0, t, alpha * t)
  mean = torch.zeros(C, device=H.DEV)
  var = torch.zeros(C, device=H.DEV)
  mm = torch.full((C,), 0.25, device=H.DEV)
  mv = torch.full((C,), 2.0, device=H.DEV)
  ws = H.reduce_ws()
  gd, bd = gam.to(H.DEV), bet.to(H.DEV)
  _lib.call('cg_bn_stats', H.p(yd), rows, C, cp, H.p(mean), H.p(var), H.p(mm),
            H.p(mv), 0.99, H.p(ws), H.stream())
  h = torch.zeros(1, rows, cp, dtype=BF16, device=H.DEV)
  _lib.call('cg_bn_apply', H.p(yd), H.p(mean), H.p(var), H.p(gd), H.p(bd), H.p(h),
            rows, C, cp, 1e-3, alpha, H.stream())
  H.sync()
  np.testing.assert_allclose(mean.cpu().numpy(), mean_r.detach().numpy(),
                             rtol=1e-4, atol=1e-5)
  np.testing.assert_allclose(var.cpu().numpy(), var_r.detach().numpy(), rtol=1e-3)
  np.testing.assert_allclose(mm.cpu().numpy(),
                             0.25 * 0.99 + 0.01 * mean_r.detach().numpy(),
                             rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(mv.cpu().numpy(),
                             2.0 * 0.99 + 0.01 * var_r.detach().numpy(), rtol=1e-5)
  np.testing.assert_allclose(h.float().cpu()[0, :, :C].numpy(),
                             h_ref.detach()[0].numpy(), rtol=1e-2, atol=1e-2)
  if cp > C:
    assert float(h[:, :, C:].float().abs().max()) == 0.0
  # backward: the mask comes from the bf16 h, as on the product path
  dhq = dhd.float().cpu()[:, :, :C]
  hq = h.float().cpu()[:, :, :C]
  mask = torch.where(hq > 0, 1.0, alpha) if act else torch.ones_like(hq)
  (t * (dhq * mask)).sum().backward()
  dy = torch.zeros(1, rows, cp, dtype=BF16, device=H.DEV)
  dg, db = H.out_buffers(True, C, C)
  runs = []
  for _ in range(2):
    _lib.call('cg_bn_bwd', H.p(dhd), H.p(h) if act else None, H.p(yd), H.p(mean),
              H.p(var), H.p(gd), H.p(dy), H.p(dg), H.p(db), rows, C, cp, 1e-3,
              alpha, act, H.p(ws), H.stream())
    H.sync()
    runs.append((dg.clone(), db.clone(), dy.clone()))
  for a, b in zip(*runs):
    assert torch.equal(a, b)
  np.testing.assert_allclose(dg.cpu().numpy(), g_.grad.numpy(), rtol=2e-3,
                             atol=2e-3 * float(g_.grad.abs().max()))
  np.testing.assert_allclose(db.cpu().numpy(), b_.grad.numpy(), rtol=2e-3,
                             atol=2e-3 * float(b_.grad.abs().max()))
  np.testing.assert_allclose(dy.float().cpu()[0, :, :C].numpy(),
                             yq.grad[0].numpy(), rtol=2e-2, atol=2e-2)
  # inference: the moving statistics in place of the batch's
  out = torch.zeros(1, rows, cp, dtype=BF16, device=H.DEV)
  _lib.call('cg_bn_apply', H.p(yd), H.p(mm), H.p(mv), H.p(gd), H.p(bd), H.p(out),
            rows, C, cp, 1e-3, 1.0, H.stream())
  H.sync()
  ref = ((yq.detach() - mm.cpu()) * torch.rsqrt(mv.cpu() + 1e-3) * gam + bet)
  np.testing.assert_allclose(out.float().cpu()[0, :, :C].numpy(), ref[0].numpy(),
                             rtol=1e-2, atol=1e-2)
