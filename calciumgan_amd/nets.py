"""Device-side state and kernel schedules of the CalciumGAN generator and
discriminator (reference: gan/models/calciumgan.py:22-103, :141-192).

PyTorch is plumbing only here: it owns device memory, the stream and (in
parallel.py) the RCCL process group.  Every arithmetic step is a call into the
gfx950 kernel library through the C ABI (include/calciumgan_hip.h); launch
descriptors are built once per batch size and replayed.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from . import geometry as geo
from ._lib import ConvDesc, PackDesc, WgradDesc

BF16 = torch.bfloat16


def act_dtype():
  """torch dtype of activations / packed operands under the active build of
  the kernel library (_lib.use): bfloat16, or float16 for mixed_float16."""
  return torch.float16 if _lib.active() == 'f16' else torch.bfloat16


def _precision_tag():
  return 1 if _lib.active() == 'f16' else 0
LEAKY_ALPHA = 0.3  # Keras LeakyReLU() default; gan/models/utils.py:6-8
LN_EPS = 1e-3  # Keras LayerNormalization default epsilon
BN_EPS = 1e-3  # Keras BatchNormalization defaults [ext]
BN_MOMENTUM = 0.99


def _p(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def precision_of(hp):
  """'f16' under --mixed_precision (the reference's mixed_float16 policy,
  main.py:22-30), else 'bf16'."""
  return 'f16' if getattr(hp, 'mixed_precision', False) else 'bf16'


def _require_gpu(hp=None):
  # raises HipLibraryError when the extension is missing; every later launch of
  # this process goes to the build this model computes with
  _lib.use(precision_of(hp))
  if not torch.cuda.is_available():
    raise RuntimeError(
        'calciumgan_amd: no HIP device visible; the MI355X kernels are the only '
        'compute path (no CPU fallback)')


class FlatParams(object):
  """All parameters of one model in ONE flat f32 buffer (one Adam launch, one
  RCCL all-reduce), with per-tensor views in Keras get_weights() order."""

  def __init__(self, shapes, device, frozen=()):
    """frozen: indices of non-trainable tensors (BatchNormalization's moving
    statistics): part of get_weights() / checkpoints, outside the parameter
    count, and never given a gradient (Adam leaves a zero-gradient entry
    alone)."""
    self.shapes = [tuple(s) for s in shapes]
    self.frozen = frozenset(int(i) for i in frozen)
    sizes = [int(np.prod(s)) for s in self.shapes]
    # 16-byte align every tensor so vector loads stay aligned
    self.offsets = []
    off = 0
    for n in sizes:
      self.offsets.append(off)
      off += geo.round_up(n, 4)
    self.numel = off
    self.count = int(sum(n for i, n in enumerate(sizes) if i not in self.frozen))
    self.data = torch.zeros(off, dtype=torch.float32, device=device)
    self.grad = torch.zeros_like(self.data)
    self.m = torch.zeros_like(self.data)
    self.v = torch.zeros_like(self.data)
    self.views = [
        self.data[o:o + n].view(s)
        for o, n, s in zip(self.offsets, sizes, self.shapes)
    ]
    self.grad_views = [
        self.grad[o:o + n].view(s)
        for o, n, s in zip(self.offsets, sizes, self.shapes)
    ]

  def get_weights(self):
    return [v.detach().cpu().numpy().copy() for v in self.views]

  def set_weights(self, weights):
    if len(weights) != len(self.views):
      raise ValueError('expected {} arrays, got {}'.format(
          len(self.views), len(weights)))
    for v, w in zip(self.views, weights):
      w = np.asarray(w, dtype=np.float32)
      if tuple(w.shape) != tuple(v.shape):
        raise ValueError('shape mismatch {} vs {}'.format(w.shape, v.shape))
      v.copy_(torch.from_numpy(w))


# Ordered reductions (default).  Every sum that used to meet through f32 atomics
# -- the penalty norm, conv / LayerNorm / Dense bias and scale gradients, the
# critic head's weight gradient, the signal metrics -- goes through per-block
# partial rows and a finishing launch that adds them in a fixed order and STORES
# the result: a process replays itself bit for bit (also across processes: the
# static tile choice is the default), and no gradient buffer needs
# zeroing.  CALCIUMGAN_DETERMINISTIC=0: the atomics (+= onto zeroed buffers).
# (CALCIUMGAN_WGRAD_PARTIALS=0 brings cg_wgrad's atomics back: they add onto
# zeroed gradients, so it switches the whole mode off)
DETERMINISTIC = (
    __import__('os').environ.get('CALCIUMGAN_DETERMINISTIC', '1') != '0' and
    __import__('os').environ.get('CALCIUMGAN_WGRAD_PARTIALS', '1') != '0')
_REDUCE_WS = {}


def reduce_ws(device):
  """The shared workspace of the ordered reductions (cg_reduce_ws_elems floats,
  one per device, precision and STREAM: what makes sharing it safe is that the
  launches that use it are ordered on one stream -- ADVICE r4), or None."""
  if not DETERMINISTIC:
    return None
  key = (str(device), _lib.active(),
         torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
  ws = _REDUCE_WS.get(key)
  if ws is None:
    n = _lib.load().cg_reduce_ws_elems()
    ws = _REDUCE_WS[key] = torch.empty(n, dtype=torch.float32, device=device)
  return ws


# CALCIUMGAN_DEFER_FINISH=0: every ordered reduction of the generator backward is
# finished by its own launch instead of one launch for all of them (A/B)
_DEFER_FINISH = __import__('os').environ.get('CALCIUMGAN_DEFER_FINISH', '1') != '0'
_REDUCE_WS_REGIONS = {}


def reduce_ws_regions(device, n):
  """n workspaces of the ordered reductions for reductions whose finishing
  launches are deferred to ONE launch (cg_finish_defer / cg_finish_flush): each
  keeps its partial rows until the flush.  Per device, precision and stream."""
  key = (str(device), _lib.active(),
         torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)
  ws = _REDUCE_WS_REGIONS.get(key)
  if ws is None or ws.shape[0] < n:
    per = _lib.load().cg_reduce_ws_elems()
    ws = _REDUCE_WS_REGIONS[key] = torch.empty(n, per, dtype=torch.float32,
                                               device=device)
  return ws


# CALCIUMGAN_NARROW_LAST=0: padded channel chunks are walked in full
_NARROW_LAST = __import__('os').environ.get('CALCIUMGAN_NARROW_LAST', '1') != '0'


def narrow_last_rule(parity_major, CK, Cx, taps, C_real):
  """Whether a packed operand's last 32-channel chunk is laid out narrow (one
  8-channel group per tap): a 102-of-128 kind of pitch, parity-major weights."""
  return bool(_NARROW_LAST and parity_major and CK == 32 and Cx >= 64 and
              taps <= 32 and Cx - 32 < C_real <= Cx - 24)


class PackedOperand(object):
  """bf16 MFMA operand of one (layer, direction), produced from the f32 master
  tensor by cg_pack_weights in the K order cg_swconv walks."""

  def __init__(self, src, phases, C_real, N_real, Cx, CK, taps,
               parity_major=False):
    """parity_major: even taps first, then odd taps -- the order a stride-2
    launch walks them with w_parity_major (needed for split-parity staging)."""
    lib = _lib.load()
    self.parity_major = bool(parity_major)
    # channel padding of the last 32-channel chunk (102 -> 128: 6 real
    # channels of 32): packed as ONE 8-channel group per tap, so a launch
    # walks 32 K groups of that chunk instead of taps * 4
    self.narrow_last = narrow_last_rule(self.parity_major, CK, Cx, taps, C_real)
    self.elems = lib.cg_packed_elems(N_real, taps, Cx, CK)
    if self.elems < 0:
      raise ValueError('bad packing geometry')
    self.taps, self.Cx, self.CK, self.N = taps, Cx, CK, N_real
    self.nphase = len(phases)
    self.buf = torch.zeros(
        self.nphase * self.elems, dtype=act_dtype(), device=src.device)
    self.src = src
    self.descs = []
    for i, (tap0, tap_step, s_tap, s_c, s_n) in enumerate(phases):
      d = PackDesc()
      d.src = src.data_ptr()
      d.dst = self.buf.data_ptr() + 2 * i * self.elems
      d.taps, d.tap0, d.tap_step = taps, tap0, tap_step
      d.s_tap, d.s_c, d.s_n = s_tap, s_c, s_n
      d.C_real, d.N_real, d.Cx, d.CK = C_real, N_real, Cx, CK
      d.parity_major = int(self.parity_major)
      d.narrow_last = int(self.narrow_last)
      self.descs.append(d)

  def repack(self):
    st = _stream()
    for d in self.descs:
      _lib.call('cg_pack_weights', ctypes.byref(d), st)


class PackPlan(object):
  """All PackedOperands of one model packed by ONE kernel launch
  (cg_pack_batched): the descriptor table lives in device memory."""

  def __init__(self, operands, device):
    lib = _lib.load()
    descs = [d for op in operands for d in op.descs]
    self.n = len(descs)
    arr = (PackDesc * self.n)(*descs)
    self.blocks = lib.cg_pack_plan_build(arr, self.n, None, 0)
    if self.blocks < 0:
      raise ValueError('bad packing geometry')
    nbytes = lib.cg_pack_plan_bytes(self.n, self.blocks)
    host = torch.zeros(nbytes, dtype=torch.uint8)
    lib.cg_pack_plan_build(arr, self.n, host.data_ptr(), nbytes)
    self.dev = host.to(device)
    self._keep = operands

  def run(self):
    _lib.call('cg_pack_batched', _p(self.dev), self.n, self.blocks, _stream())


def _transpose_phases(k, pad_left):
  """Tap walk of the two output phases of a stride-2 transposed convolution
  (SURVEY Appendix A.2): out[2u+p] = sum_jj src[u + off_p + jj] * W[tap0_p - 2jj].
  Returns [(tap0, off)] for p = 0, 1."""
  out = []
  for p in (0, 1):
    kk0 = (p + pad_left) & 1
    off = (p + pad_left - kk0) // 2 - (k // 2 - 1)
    out.append((kk0 + k - 2, off))
  return out


CK_TARGET = {1: 32, 2: 32}  # preferred channel chunk per source stride


def _ck_for(Cx, stride, taps, Lu):
  """Channel chunk of a packed operand: a divisor of the pitch, multiple of 8,
  >= 32, valid for both row tiles the launcher may pick.  Preference: the chunk
  that keeps one workgroup's LDS under half the CU (two resident workgroups
  hide each other's staging): 32."""
  cands = [d for d in range(32, Cx + 1, 8) if Cx % d == 0]
  tgt = CK_TARGET[stride]
  pref = sorted(cands, key=lambda d: (d > tgt, abs(d - tgt)))
  tms = [tm for tm in (256, 64)
         if ((Lu % tm == 0) if Lu >= tm else (tm % Lu == 0))]
  if not tms:
    raise ValueError('unsupported per-sample length {}'.format(Lu))
  for d in pref:
    if all(geo.lds_bytes(d, stride, taps, Lu, tm) <= geo.LDS_BYTES for tm in tms):
      return d
  # fall back to the small tile only
  for d in pref:
    if geo.lds_bytes(d, stride, taps, Lu, 64) <= geo.LDS_BYTES:
      return d
  raise ValueError('no channel chunk fits LDS (Cx={}, taps={})'.format(Cx, taps))


def _conv_desc(x, w, y, nB, Lx, Cx, taps, stride, off, Lu, N, Ly, Cy, CK,
               y_stride=1, y_off=0, bias=None, mask_src=None, shifts=None,
               seg_size=1, epilogue=_lib.EPI_NONE, out_f32=False, nphase=1,
               w_phase_stride=0, off_phase_step=0, yoff_phase_step=0,
               rowsumsq=None, w_parity_major=False, ln=None, out_shifts=None,
               w_narrow_last=False, alpha=LEAKY_ALPHA, row_scale=None):
  d = ConvDesc()
  d.w_parity_major = int(bool(w_parity_major) and stride == 2)
  d.w_narrow_last = int(bool(w_narrow_last) and d.w_parity_major)
  d.split_parity = 0
  d._keep = (x, w, y, bias, mask_src, shifts, rowsumsq)  # borrowed pointers
  d.rowsumsq = rowsumsq.data_ptr() if rowsumsq is not None else None
  d.row_scale = row_scale.data_ptr() if row_scale is not None else None
  if row_scale is not None:
    d._keep = d._keep + (row_scale,)
  d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
  d.bias = bias.data_ptr() if bias is not None else None
  d.mask_src = mask_src.data_ptr() if mask_src is not None else None
  d.shifts = shifts.data_ptr() if shifts is not None else None
  d.nB, d.Lx, d.Cx, d.seg_size = nB, Lx, Cx, seg_size
  d.taps, d.stride, d.off, d.Lu = taps, stride, off, Lu
  d.N, d.Ly, d.Cy, d.y_stride, d.y_off = N, Ly, Cy, y_stride, y_off
  d.CK = CK
  d.epilogue, d.out_f32, d.alpha = epilogue, int(out_f32), alpha
  d.nphase, d.w_phase_stride = nphase, w_phase_stride
  d.off_phase_step, d.yoff_phase_step = off_phase_step, yoff_phase_step
  n_tiles_n = (N + 63) // 64
  small, tm = geo.tile_rows(Lu, nB * Lu, n_tiles_n * nphase)
  if geo.lds_bytes(CK, stride, taps, Lu, tm) > geo.LDS_BYTES:
    small, tm = 1, 64
  # stride-2 windows with a 32-channel chunk: 128-row tiles fit three
  # workgroups per CU (measured +10 % over 256-row tiles at two per CU)
  if small == 0 and stride == 2 and CK <= 32 and Lu % 128 == 0:
    small = 2
  d.tile = small
  d.ksplit = 0
  nchunks = Cx // CK
  if (_SPLIT_K and x.is_cuda and not out_f32 and rowsumsq is None and
      ln is None and out_shifts is None and row_scale is None and
      nchunks % 2 == 0 and
      epilogue in (_lib.EPI_NONE, _lib.EPI_LRELU, _lib.EPI_MASK) and
      nB * Lu * n_tiles_n * nphase <= _SPLIT_K_MAX_TILES * 64):
    # few output tiles (the tangent chain's single segment): let the tuner try
    # 2 / 4 workgroups per tile, each over a share of the channel chunks.
    # Workspaces are shared per size (launches are stream-ordered).
    smax = 4 if nchunks % 4 == 0 else 2
    need = smax * nB * Ly * Cy
    ws = _SPLIT_WS.get(need)
    if ws is None:
      ws = _SPLIT_WS[need] = torch.empty(need, dtype=torch.float32,
                                         device=x.device)
    d._keep = d._keep + (ws,)
    d.split_ws, d.split_ws_elems = ws.data_ptr(), need
  if out_shifts is not None:
    sh_t, d.out_seg_size, side_t, d.side_rows = out_shifts
    d._keep = d._keep + (sh_t, side_t)
    d.out_shifts, d.side = sh_t.data_ptr(), side_t.data_ptr()
  if ln is not None:
    # LayerNorm + LeakyReLU in the producing launch (gamma, beta, h, mean,
    # rstd): the row statistics need the whole row in one workgroup
    d._keep = d._keep + tuple(ln)
    d.epilogue = _lib.EPI_LN_LRELU
    d.ln_gamma, d.ln_beta, d.ln_h, d.ln_mean, d.ln_rstd = (
        None if t is None else t.data_ptr() for t in ln)
    d.ln_eps = LN_EPS
    d.tile = 6  # CG_TILE_128x128_M32
  _autotune_tile(d)
  if rowsumsq is not None and DETERMINISTIC and x.is_cuda:
    # ordered penalty norm: one slot per workgroup of the chosen tile
    need = _lib.load().cg_rowsumsq_ws_elems(ctypes.byref(d))
    if need <= 0:
      # (never fall back to the atomics norm silently in ordered mode)
      raise RuntimeError('calciumgan_amd: no ordered penalty-norm workspace for '
                         'this launch (tile {}); CALCIUMGAN_DETERMINISTIC=0 '
                         'selects the atomics form'.format(d.tile))
    if need > 0:
      wsq = torch.empty(need, dtype=torch.float32, device=x.device)
      d._keep = d._keep + (wsq,)
      d.rowsumsq_ws, d.rowsumsq_ws_elems = wsq.data_ptr(), need
      d._ssq = (wsq, need // nB)  # the slots and their count per sample
  return d


# CALCIUMGAN_FOLD_SCALE=0: the penalty's v = coef_b * g as its own pass over g
# (cg_scale_rows into X0) instead of a per-sample scale in the tangent chain's
# first launch + a pass over the x^ segment of delta_1
_FOLD_SCALE = __import__('os').environ.get('CALCIUMGAN_FOLD_SCALE', '1') != '0'
# Critic layer 1 on x^ = a real + (1 - a) fake taken from the layer's outputs on
# real and fake (the convolution is linear: cg_lrelu_mix) instead of a third of the
# layer's launch; x^ itself is then never formed.  0: convolve x^ like the other two
# segments (rounds 1-4).
_L1_LINEAR = __import__('os').environ.get('CALCIUMGAN_L1_LINEAR', '1') != '0'
# ... for segments of at least this many layer-1 output rows (B * L / 2): below,
# a third of the layer's launch is shorter than the extra launch (cfg2: 131 072)
_L1_LINEAR_MIN_ROWS = int(__import__('os').environ.get(
    'CALCIUMGAN_L1_LINEAR_MIN_ROWS', '16384'))
# Split-K candidates of the tile tuner (CALCIUMGAN_SPLIT_K=0: never): only for
# launches whose output is at most this many 64x64 tiles
_SPLIT_K = __import__('os').environ.get('CALCIUMGAN_SPLIT_K', '1') != '0'
_SPLIT_K_MAX_TILES = 2048
_SPLIT_WS = {}
# CALCIUMGAN_FUSE_UNSHUFFLE=0: cg_unshuffle_mask as its own pass after every
# input-gradient launch of the critic
_FUSE_UNSHUFFLE = __import__('os').environ.get('CALCIUMGAN_FUSE_UNSHUFFLE',
                                                '1') != '0'
# CALCIUMGAN_FUSE_LN=0 keeps LayerNorm a separate pass (A/B, debugging)
_FUSE_LN = __import__('os').environ.get('CALCIUMGAN_FUSE_LN', '1') != '0'


def _ln_fusable(lay, CK, taps):
  """Conv1DTranspose + LayerNorm in one launch: rows of at most 128 channels
  (one 128-column tile) on the 32x32x16 tiles (uniform 32-channel K walk)."""
  return (_FUSE_LN and lay.cout <= 128 and CK % 32 == 0 and geo.lds_bytes(
      CK, 1, taps, lay.lin, 128, 128, 32) <= geo.LDS_BYTES)


# Tile / weight-stage choice by measurement: the best of the CG_TILE_* shapes x
# {64, 128}-deep weight stages depends on how the launch quantises over 256 CUs
# and on LDS residency, so each distinct launch geometry is timed once (3 runs
# per candidate, a few milliseconds in total) when its descriptor is first
# built.  All candidates walk K in the same order; the 16x16x32 and 32x32x16
# MFMA shapes may differ in the last f32 bit of a 32-deep partial sum, so a
# process keeps ONE choice per geometry.  Since round 5 the tuner is OPT-IN
# (CALCIUMGAN_AUTOTUNE=1): the static choice below costs nothing measurable on the
# whole step (profiles/r04_static_tiles_whole_step.txt: 12.67 / 12.81 ms static vs
# 12.74 / 12.72 tuned) and makes every run -- bench.py, main.py, the tests --
# replay bit for bit across processes.
_TILE_CACHE = {}
_AUTOTUNE = __import__('os').environ.get('CALCIUMGAN_AUTOTUNE', '0') == '1'
# CALCIUMGAN_TILE_CACHE=<file.json>: choices are loaded from / saved to this
# file, so later processes (profiler passes, the other ranks' restarts) launch
# exactly the tuned configuration without re-timing candidates.
_TILE_CACHE_FILE = __import__('os').environ.get('CALCIUMGAN_TILE_CACHE')
# CALCIUMGAN_TUNE_LOG=<file.jsonl>: every tuned geometry with all candidates'
# times (per-geometry tables under profiles/ come from this)
_TUNE_LOG = __import__('os').environ.get('CALCIUMGAN_TUNE_LOG')
# CALCIUMGAN_SWP_TILES=0: the software-pipelined tiles are not offered to the tuner
_SWP_TILES = __import__('os').environ.get('CALCIUMGAN_SWP_TILES', '1') != '0'


def load_tile_cache(path):
  """Merge a saved tile table (CALCIUMGAN_TILE_CACHE format) into this
  process's choices; returns the number of geometries read."""
  import json
  with open(path) as f:
    table = json.load(f)
  for k, v in table.items():
    _TILE_CACHE[tuple(int(t) for t in k.split(','))] = tuple(v)
  return len(table)


def _load_tile_cache():
  import os
  if _TILE_CACHE_FILE and os.path.exists(_TILE_CACHE_FILE):
    load_tile_cache(_TILE_CACHE_FILE)


def _save_tile_cache():
  import json
  from . import parallel
  if parallel.env_rank() != 0:  # one writer under torchrun
    return
  if _TILE_CACHE_FILE and _TILE_CACHE:
    with open(_TILE_CACHE_FILE, 'w') as f:
      json.dump({','.join(str(t) for t in k): list(v)
                 for k, v in _TILE_CACHE.items()}, f, indent=0)


_load_tile_cache()
__import__('atexit').register(_save_tile_cache)


def _autotune_tile(d):
  if not torch.cuda.is_available():
    return
  # everything that decides which candidates are VALID is part of the key (a
  # choice tuned for an unconstrained launch must not reach a constrained one)
  key = (d.stride, d.taps, d.nB, d.Lx, d.Cx, d.Lu, d.N, d.CK, d.nphase,
         d.epilogue, d.out_f32, d.w_narrow_last, int(bool(d.rowsumsq)),
         int(bool(d.out_shifts)), int(bool(d.split_ws)), _precision_tag())
  if d.row_scale:  # (appended only when set: saved tables keep their keys)
    key = key + (1,)
  from . import parallel
  multi = parallel.world_size() > 1
  best = _TILE_CACHE.get(key)
  if multi:
    # every rank builds the same descriptors in the same order: rank 0 tunes
    # (or reads its cache) and all ranks launch ITS choice -- identical kernels
    # and reduction orders on every rank
    if parallel.rank() != 0:
      best = parallel.broadcast_object(None)
      if best is not None:
        _TILE_CACHE[key] = best
      return _apply_tile_choice(d, best)
  if best is None and not _AUTOTUNE:
    # no timing: the software-pipelined tile that wins most geometries when it
    # is tuned (8 waves of 32 x 64 wave tiles; the 128-column form for the
    # fused LayerNorm), if the library admits it for this launch; else the
    # static tile-kernel choice stays
    best = _static_swp_choice(d)
    if best is not None:
      _TILE_CACHE[key] = best  # (every rank holds the table it launches)
    if multi:
      parallel.broadcast_object(best)
    return _apply_tile_choice(d, best)
  if best is None:
    lib = _lib.load()
    st = _stream()
    cands = []
    for small, (tm, tn, mf) in _lib.TILES.items():
      ok = (d.Lu % tm == 0) if d.Lu >= tm else (tm % d.Lu == 0)
      if d.rowsumsq and d.Lu < tm:
        ok = False
      if mf == 32 and d.CK % 32:
        ok = False
      if d.epilogue == _lib.EPI_LN_LRELU:
        if tn != 128:
          ok = False
      elif tn > 64 and d.N <= 64:
        ok = False
      if ok and geo.lds_bytes(d.CK, d.stride, d.taps, d.Lu, tm, tn,
                              mf) <= geo.LDS_BYTES:
        cands.append(small)
    cands = [(small, ks, 0, 1) for small in cands for ks in (2, 4)]
    swp = []
    if (_SWP_TILES and d.CK == 32 and
        d.taps % d.stride == 0 and (d.taps // d.stride) % 2 == 0 and
        d.taps // d.stride >= 6 and (d.stride == 1 or d.w_parity_major)):
      # software-pipelined tiles (two waves per SIMD, swconv_swp.hip)
      for small, (tm, tn) in _lib.SWP_TILES.items():
        ok = (d.Lu % tm == 0) if d.Lu >= tm else (tm % d.Lu == 0)
        if d.rowsumsq and d.Lu < tm:
          ok = False
        if d.epilogue == _lib.EPI_LN_LRELU:
          if tn != 128:
            ok = False
        elif tn > 64 and d.N <= 64:
          ok = False
        if ok:
          swp.append((small, 2, 0, 1))
    if d.stride == 2 and d.w_parity_major:
      # split-parity staging: half the LDS window, twice the staging phases
      cands += [(small, ks, 1, 1) for small, ks, _, _ in list(cands)
                if geo.lds_bytes(d.CK, 2, d.taps, d.Lu, _lib.TILES[small][0],
                                 _lib.TILES[small][1], _lib.TILES[small][2],
                                 split_parity=True) <= geo.LDS_BYTES]
    if d.split_ws:
      nch = d.Cx // d.CK
      cands += [(small, ks, sp, z) for small, ks, sp, _ in list(cands)
                for z in (2, 4) if nch % z == 0]
      if not d.w_narrow_last:
        swp += [(small, ks, sp, z) for small, ks, sp, _ in list(swp)
                for z in (2, 4) if nch % z == 0]
    cands += swp
    times = {}
    y_saved = d.y
    scratch = None
    if d.mask_src and d.mask_src == d.y:
      # in-place launch (penalty tangent): tune on a scratch output so live
      # activations are not rewritten
      nbytes = d.nB * d.Ly * d.Cy * (4 if d.out_f32 else 2)
      scratch = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
      d.y = scratch.data_ptr()
    for small, ks, sp, z in cands:
      d.tile = small
      d.stage_ksteps = ks
      d.split_parity = sp
      d.ksplit = z
      if lib.cg_swconv(ctypes.byref(d), st) != 0:
        continue
      s = torch.cuda.Event(enable_timing=True)
      e = torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(3):
        lib.cg_swconv(ctypes.byref(d), st)
      e.record()
      e.synchronize()
      times[(small, ks, sp, z)] = s.elapsed_time(e)
    d.y = y_saved
    d.split_parity = 0
    d.ksplit = 0
    del scratch
    if times:
      best = min(times, key=times.get)
      _TILE_CACHE[key] = best
      if _TUNE_LOG:
        import json
        with open(_TUNE_LOG, 'a') as f:
          f.write(json.dumps({
              'key': list(key),
              'times_us': {','.join(map(str, c)): round(t / 3 * 1e3, 2)
                           for c, t in sorted(times.items(),
                                              key=lambda kv: kv[1])}}) + '\n')
  if multi:
    parallel.broadcast_object(best)
  _apply_tile_choice(d, best)


# CALCIUMGAN_STATIC_TILES="a,b,..|c,d,.." (development): the static preference
# order of the software-pipelined tiles, plain launches | fused-LayerNorm ones
_STATIC_ORDER = __import__('os').environ.get('CALCIUMGAN_STATIC_TILES')


def _static_swp_choice(d):
  if not _SWP_TILES or d.CK != 32:
    return None
  order = (15, 12) if d.epilogue == _lib.EPI_LN_LRELU else (14, 13, 10)
  if _STATIC_ORDER:
    plain, _, ln = _STATIC_ORDER.partition('|')
    pick = ln if d.epilogue == _lib.EPI_LN_LRELU else plain
    if pick:
      order = tuple(int(t) for t in pick.split(','))
  lib = _lib.load()
  saved = (d.tile, d.stage_ksteps, d.split_parity, d.ksplit)
  pick = None
  # a launch with far fewer 256 x 64 tiles than CUs (the generator backward's
  # first layer at B = 128: 320 -> 32 channels, 32 tiles): 128-row tiles, two
  # workgroups per tile over halves of the channel chunks (54 -> 30 us; the one
  # geometry of cfg2 where the static tile was not within 6 % of the tuner's
  # best: profiles/r05_static_vs_tuned_tiles.txt; 128-column tiles for the
  # under-filled launches with wide outputs, 3-6 % ahead in that table, were
  # worth 0.1 % of the step in a same-box A/B and are not a rule)
  tiles = ((d.nB * d.Lu + 255) // 256) * ((d.N + 63) // 64) * d.nphase
  if (d.split_ws and tiles <= 64 and d.epilogue != _lib.EPI_LN_LRELU and
      (d.Cx // d.CK) % 2 == 0 and not _STATIC_ORDER):
    d.tile, d.stage_ksteps, d.split_parity, d.ksplit = 13, 2, 0, 2
    if lib.cg_swconv_check(ctypes.byref(d)) == 0:
      pick = (13, 2, 0, 2)
  if pick is None:
    for tile in order:
      d.tile, d.stage_ksteps, d.split_parity, d.ksplit = tile, 2, 0, 0
      if lib.cg_swconv_check(ctypes.byref(d)) == 0:
        pick = (tile, 2, 0, 1)
        break
  d.tile, d.stage_ksteps, d.split_parity, d.ksplit = saved
  return pick


def _apply_tile_choice(d, best):
  """Set a tuned (tile, stage depth, split-parity, split-K) choice on a
  descriptor after checking that the library accepts it for THIS launch (a
  file-loaded table may come from another build); the static default stays
  otherwise."""
  if best is None:
    return
  saved = (d.tile, d.stage_ksteps, d.split_parity, d.ksplit)
  d.tile, d.stage_ksteps, d.split_parity = best[:3]
  d.ksplit = best[3] if len(best) > 3 and d.split_ws else 0
  if _lib.load().cg_swconv_check(ctypes.byref(d)) != 0:
    d.tile, d.stage_ksteps, d.split_parity, d.ksplit = saved


# K'-split partial sums of cg_wgrad: plain stores + a reducing launch instead of
# f32 atomics (CALCIUMGAN_WGRAD_PARTIALS=0: atomics).  Workspaces are shared by
# the descriptors of one slot (= position in a batched launch): launches are
# stream-ordered, so a slot's buffer is free again when the next batch starts.
_WGRAD_PARTIALS = __import__('os').environ.get('CALCIUMGAN_WGRAD_PARTIALS',
                                                '1') != '0'
_PARTIALS_POOL = {}
# CALCIUMGAN_WGRAD_XCD=0: plain block order instead of the XCD-grouped one
_WGRAD_XCD = __import__('os').environ.get('CALCIUMGAN_WGRAD_XCD', '1') != '0'


def _wgrad_desc(x, g, dw, nB, Lx, Cx, Lu, Cg, taps, stride, off, Cx_real,
                Cg_real, shifts=None, seg_size=1, dbias=None, bias_rows=0,
                slot=None):
  d = WgradDesc()
  d._keep = (x, g, dw, shifts, dbias)  # pointers below borrow these
  d.dbias = dbias.data_ptr() if dbias is not None else None
  d.bias_rows = bias_rows
  d.x, d.g, d.dw = x.data_ptr(), g.data_ptr(), dw.data_ptr()
  d.shifts = shifts.data_ptr() if shifts is not None else None
  d.nB, d.Lx, d.Cx, d.seg_size = nB, Lx, Cx, seg_size
  d.Lu, d.Cg = Lu, Cg
  d.taps, d.stride, d.off = taps, stride, off
  d.Cx_real, d.Cg_real = Cx_real, Cg_real
  d.nsplit = 0
  d.no_xcd_group = 0 if _WGRAD_XCD else 1
  if slot is not None and _WGRAD_PARTIALS and x.is_cuda:
    need = _lib.load().cg_wgrad_partials_elems(ctypes.byref(d))
    if need > 0:
      buf = _PARTIALS_POOL.get((slot, need))
      if buf is None:
        buf = torch.empty(need, dtype=torch.float32, device=x.device)
        _PARTIALS_POOL[(slot, need)] = buf
      d._keep = d._keep + (buf,)
      d.partials, d.partials_elems = buf.data_ptr(), need
      d.store = 1 if DETERMINISTIC else 0
    elif need == 0 and DETERMINISTIC:
      # one K' split (or the taps == 1 form, pinned to one): every dw element
      # has a single owner, stored directly
      d.nsplit = 1
      d.store = 1
  return d


# Optional per-launch timing of the two MFMA kernel families (bench.py's
# roofline leg): a list that receives (family, start_event, end_event), events
# recorded on the stream the kernel is launched on.
_PROFILE = None


def set_profile(records):
  global _PROFILE
  _PROFILE = records


def _timed(name, family, d, st):
  if _PROFILE is None:
    _lib.call(name, ctypes.byref(d), st)
    return
  s = torch.cuda.Event(enable_timing=True)
  e = torch.cuda.Event(enable_timing=True)
  s.record()
  _lib.call(name, ctypes.byref(d), st)
  e.record()
  _PROFILE.append((family, s, e))


def _run_conv(d, st):
  _timed('cg_swconv', 'swconv', d, st)


_BATCH_WGRAD = __import__('os').environ.get('CALCIUMGAN_WGRAD_BATCH', '1') != '0'


def _run_wgrads(descs, st):
  """Independent weight gradients of one backward pass as one cg_wgrad_batched
  call (a single launch when they share the pipelined stride-2 form: each
  layer's accumulator flush then overlaps the next layer's main loop)."""
  if _PROFILE is not None or len(descs) == 1 or not _BATCH_WGRAD:
    for d in descs:
      _run_wgrad(d, st)
    return
  arr = (WgradDesc * len(descs))(*descs)
  _lib.call('cg_wgrad_batched', arr, len(descs), st)


def _run_wgrad(d, st):
  _timed('cg_wgrad', 'wgrad', d, st)


def glorot_uniform(rng, shape, fan_in, fan_out):
  limit = math.sqrt(6.0 / (fan_in + fan_out))
  return rng.uniform(-limit, limit, size=shape).astype(np.float32)


# ===========================================================================
# Discriminator
# ===========================================================================
class DiscriminatorNet(object):
  """5 x [Conv1D(k, s=2, 'same') -> LeakyReLU -> PhaseShuffle] -> Flatten ->
  Dense(1)  (calciumgan.py:141-192)."""

  def __init__(self, hp, device, rng):
    _require_gpu(hp)
    self.precision = precision_of(hp)
    self.h_dtype = act_dtype()  # torch dtype of activations / operands
    geo.validate_hparams(hp)
    self.hp = hp
    self.alpha = geo.activation_alpha(hp)  # x -> max(x, alpha x)
    self.device = device
    self.k = hp.kernel_size
    self.pl = geo.same_padding_left(self.k, hp.strides)
    self.layers = geo.discriminator_layers(hp)
    shapes = []
    for lay in self.layers:
      shapes += [(self.k, lay.cin, lay.cout), (lay.cout,)]
    last = self.layers[-1]
    self.flat = last.lout * last.cout
    shapes += [(self.flat, 1), (1,)]
    self.params = FlatParams(shapes, device)
    init = []
    for lay in self.layers:
      init.append(
          glorot_uniform(rng, (self.k, lay.cin, lay.cout), self.k * lay.cin,
                         self.k * lay.cout))
      init.append(np.zeros(lay.cout, np.float32))
    init.append(glorot_uniform(rng, (self.flat, 1), self.flat, 1))
    init.append(np.zeros(1, np.float32))
    self.params.set_weights(init)
    # packed operands
    self.w_fwd, self.w_dgrad = [], []
    phases = _transpose_phases(self.k, self.pl)
    self.dgrad_offs = [o for _, o in phases]
    for i, lay in enumerate(self.layers):
      W = self.params.views[2 * i]
      ci, co = lay.cin, lay.cout
      ck = _ck_for(lay.cinp, 2, self.k, lay.lout)
      self.w_fwd.append(
          PackedOperand(W, [(0, 1, ci * co, co, 1)], ci, co, lay.cinp, ck,
                        self.k, parity_major=True))
      ck = _ck_for(lay.coutp, 1, self.k // 2, lay.lin // 2)
      self.w_dgrad.append(
          PackedOperand(W, [(t0, -2, ci * co, 1, co) for t0, _ in phases], co,
                        ci, lay.coutp, ck, self.k // 2))
    self._pack_plan = PackPlan(self.w_fwd + self.w_dgrad, device)
    self.repack()
    self._ws = {}

  # -- parameters ---------------------------------------------------------
  def repack(self):
    self._pack_plan.run()

  @property
  def dense_w(self):
    return self.params.views[-2]

  @property
  def dense_b(self):
    return self.params.views[-1]

  # -- workspace ----------------------------------------------------------
  def workspace(self, nB):
    ws = self._ws.get(nB)
    if ws is None:
      ws = _DisWorkspace(self, nB)
      self._ws[nB] = ws
    return ws


class _DisWorkspace(object):
  """Activations / gradients of the discriminator for a batch of nB samples and
  the launch descriptors over them (also for sub-batches that start at sample
  0 or at a segment boundary)."""

  def __init__(self, net, nB):
    dev = net.device
    self.net = net
    self.nB = nB
    L0 = net.layers[0].lin
    cp0 = net.layers[0].cinp
    z = lambda *s, dt=None: torch.zeros(*s, dtype=dt or act_dtype(), device=dev)
    self.act = [z(nB, L0, cp0)] + [z(nB, l.lout, l.coutp) for l in net.layers]
    self.e = [None] + [z(nB, l.lout, l.coutp) for l in net.layers[:-1]]
    self.delta = [None] + [z(nB, l.lout, l.coutp) for l in net.layers]
    self.d_out = z(nB, dt=torch.float32)
    self._plans = {}

  def x0(self, k):
    """Input buffer X0 of critic update k of a step: update 0 uses act[0]; the
    others get their own buffer, so that ONE launch at the start of the step
    (cg_dense_rows_interp) can leave every update's [real | fake_k | x^_k] in
    place (WGAN_GP._critic_generate_all).  201 MB each at cfg2."""
    if k == 0:
      return self.act[0]
    alt = self.__dict__.setdefault('_x0_alt', {})
    if k not in alt:
      alt[k] = torch.zeros_like(self.act[0])
    return alt[k]

  def plan(self, nB, seg_size, input_grad_from, want_norm=True, x0_index=0,
           shifts=None):
    """Descriptors for a run over the first nB samples with shift segments of
    seg_size samples; the layer-1 input gradient is computed for samples
    [input_grad_from, nB) (None = not at all).  want_norm: its per-sample sum of
    squares (the penalty norm) is taken in the same launch -- the generator
    update's pass does not need it.  x0_index: which of the step's input
    buffers (x0(k)) the layer-1 launches read.  shifts (first use only): the int32
    (4, segments) tensor the plan's launches read their PhaseShuffle draws from
    (a view of the caller's staging buffer: no copy in front of a replay)."""
    key = (nB, seg_size, input_grad_from, bool(want_norm), int(x0_index))
    pl = self._plans.get(key)
    if pl is None:
      pl = _DisPlan(self, nB, seg_size, input_grad_from, want_norm,
                    x0=self.x0(x0_index), shifts=shifts)
      self._plans[key] = pl
    return pl


class _DisPlan(object):

  def __init__(self, ws, nB, seg_size, input_grad_from, want_norm=True, x0=None,
               shifts=None):
    net = ws.net
    dev = net.device
    self.ws, self.nB, self.seg_size = ws, nB, seg_size
    # the layer-1 input of this plan's launches (ws.act[0] unless the step keeps
    # one input buffer per critic update: _DisWorkspace.x0)
    self.x0 = ws.act[0] if x0 is None else x0
    src = lambda i: self.x0 if i == 0 else ws.act[i]
    self.nseg = (nB + seg_size - 1) // seg_size
    k, pl = net.k, net.pl
    # shifts[l][seg]: PhaseShuffle draw applied after layer l+1 (l = 0..3)
    if shifts is not None:
      assert (tuple(shifts.shape) == (4, self.nseg) and shifts.is_contiguous() and
              shifts.dtype == torch.int32)
    self.shifts = (shifts if shifts is not None else
                   torch.zeros(4, self.nseg, dtype=torch.int32, device=dev))
    self.coef = torch.zeros(self.nseg, dtype=torch.float32, device=dev)
    self.bias_coef = torch.zeros(self.nseg, dtype=torch.float32, device=dev)
    self.fwd, self.dgrad, self.jvp, self.wgrad = [], [], [], []
    for i, lay in enumerate(net.layers):
      sh = self.shifts[i - 1] if i > 0 else None
      bias = net.params.views[2 * i + 1]
      op = net.w_fwd[i]
      self.fwd.append(
          _conv_desc(src(i), op.buf, ws.act[i + 1], nB, lay.lin, lay.cinp, k,
                     2, -pl, lay.lout, lay.cout, lay.lout, lay.coutp, op.CK,
                     bias=bias, shifts=sh, seg_size=seg_size,
                     epilogue=_lib.EPI_LRELU, w_parity_major=op.parity_major,
                     w_narrow_last=op.narrow_last, alpha=net.alpha))
      self.wgrad.append(
          _wgrad_desc(src(i), ws.delta[i + 1], net.params.grad_views[2 * i],
                      nB, lay.lin, lay.cinp, lay.lout, lay.coutp, k, 2, -pl,
                      lay.cin, lay.cout, shifts=sh, seg_size=seg_size,
                      dbias=net.params.grad_views[2 * i + 1], slot=i))
    # input-gradient chain: layer i (1-based l = i+1) maps delta[l] -> e[l-1]
    # PhaseShuffle adjoint + LeakyReLU' mask in the input-gradient launch's
    # epilogue (rows stored at their source positions, masked there) when the
    # reflected and the empty rows of a sample cannot coincide; the at most m
    # reflected rows per sample go through `side` and cg_unshuffle_fixup
    m = max(1, int(net.hp.m))
    self.side = {}
    for i in range(len(net.layers) - 1, 0, -1):
      lay = net.layers[i]
      op = net.w_dgrad[i]
      fused = _FUSE_UNSHUFFLE and 2 * m + 1 <= lay.lin
      extra = {}
      if fused:
        self.side[i] = torch.zeros(nB, m, lay.cinp, dtype=act_dtype(),
                                   device=dev)
        extra = dict(mask_src=ws.act[i], epilogue=_lib.EPI_MASK,
                     out_shifts=(self.shifts[i - 1], seg_size, self.side[i], m),
                     alpha=net.alpha)
      self.dgrad.append((i,
                         _conv_desc(ws.delta[i + 1], op.buf,
                                    ws.delta[i] if fused else ws.e[i], nB,
                                    lay.lout, lay.coutp, k // 2, 1,
                                    net.dgrad_offs[0], lay.lin // 2, lay.cin,
                                    lay.lin, lay.cinp, op.CK, y_stride=2,
                                    y_off=0, nphase=2,
                                    w_phase_stride=op.elems,
                                    off_phase_step=net.dgrad_offs[1] -
                                    net.dgrad_offs[0], yoff_phase_step=1,
                                    **extra)))
    # layer 1 over [real | fake] only + cg_lrelu_mix for the x^ segment (_L1_LINEAR)
    self.fwd_l1_pair = None
    if (_L1_LINEAR and self.nseg == 3 and nB == 3 * seg_size and
        input_grad_from == 2 * seg_size and 0.0 < net.alpha <= 1.0 and
        seg_size * net.layers[0].lout >= _L1_LINEAR_MIN_ROWS):
      lay, op = net.layers[0], net.w_fwd[0]
      self.fwd_l1_pair = _conv_desc(
          self.x0[:2 * seg_size], op.buf, ws.act[1][:2 * seg_size], 2 * seg_size,
          lay.lin, lay.cinp, k, 2, -pl, lay.lout, lay.cout, lay.lout, lay.coutp,
          op.CK, bias=net.params.views[1], seg_size=seg_size,
          epilogue=_lib.EPI_LRELU, w_parity_major=op.parity_major,
          w_narrow_last=op.narrow_last, alpha=net.alpha)
    self.input_grad = None
    self.gin = None
    if input_grad_from is not None:
      lay = net.layers[0]
      op = net.w_dgrad[0]
      nG = nB - input_grad_from
      self.nG = nG
      # bf16 like every other activation gradient (its f32 sum of squares,
      # the penalty norm, is taken in the producing launch's epilogue).  When a
      # tangent chain follows (the critic's x^ segment) it is written straight
      # over that segment of X0 -- x^ is dead once layer 1 has run forward --
      # where the chain's first launch and the layer-1 weight gradient read it
      # (_FOLD_SCALE: no separate pass that scales it by the penalty's coef_b)
      self.gin_in_x0 = _FOLD_SCALE and input_grad_from > 0
      self.gin = (self.x0[input_grad_from:nB] if self.gin_in_x0 else
                  torch.zeros(nG, lay.lin, lay.cinp, dtype=act_dtype(),
                              device=dev))
      # penalty norm fused into this launch's epilogue when a 256-row tile
      # never spans two samples; else the standalone cg_rownorm is used
      self.sumsq = None
      if want_norm and lay.lin // 2 >= 256 and (lay.lin // 2) % 256 == 0:
        self.sumsq = torch.zeros(nG, dtype=torch.float32, device=dev)
      self.input_grad = _conv_desc(
          ws.delta[1][input_grad_from:], op.buf, self.gin, nG, lay.lout,
          lay.coutp, k // 2, 1, net.dgrad_offs[0], lay.lin // 2, lay.cin,
          lay.lin, lay.cinp, op.CK, y_stride=2, y_off=0, nphase=2,
          w_phase_stride=op.elems,
          off_phase_step=net.dgrad_offs[1] - net.dgrad_offs[0],
          yoff_phase_step=1, rowsumsq=self.sumsq)
    # (slots tensor, slots per sample) of the ordered penalty norm, or None
    self.ssq = getattr(self.input_grad, '_ssq', None)
    self.norm_deferred = False

  def defer_norm(self):
    """Leave the penalty norm's slots unsummed: the caller's cg_gp_loss_scale adds
    them (one launch for norm, penalty, loss and v's scale).  Returns whether the
    plan can (ordered mode with the norm fused into the input-gradient launch)."""
    if self.ssq is None:
      return False
    self.input_grad.rowsumsq_defer = 1
    self.norm_deferred = True
    return True

  def build_jvp(self, seg_index, coef=None):
    """Tangent-forward chain (gradient-penalty second backward) over segment
    seg_index, in place over that segment's activations.  coef (f32 per sample
    of the segment): the chain starts from v = coef_b * g; with the input
    gradient g sitting in X0 (gin_in_x0) the first launch reads g and applies
    coef_b in its epilogue (cg_conv_desc.row_scale)."""
    ws, net = self.ws, self.ws.net
    fold = coef is not None and getattr(self, 'gin_in_x0', False)
    s0 = seg_index * self.seg_size
    n = min(self.seg_size, self.nB - s0)
    k, pl = net.k, net.pl
    lib = _lib.load()

    def build(fold_first):
      descs = []
      for i, lay in enumerate(net.layers):
        sh = self.shifts[i - 1][seg_index:] if i > 0 else None
        op = net.w_fwd[i]
        seg_act = ws.act[i + 1][s0:s0 + n]
        descs.append(
            _conv_desc((self.x0 if i == 0 else ws.act[i])[s0:s0 + n], op.buf,
                       seg_act, n, lay.lin,
                       lay.cinp, k, 2, -pl, lay.lout, lay.cout, lay.lout,
                       lay.coutp, op.CK, mask_src=seg_act, shifts=sh,
                       seg_size=n, epilogue=_lib.EPI_MASK,
                       w_parity_major=op.parity_major,
                       w_narrow_last=op.narrow_last, alpha=net.alpha,
                       row_scale=coef if (fold_first and i == 0) else None))
      return descs

    self.jvp = build(fold)
    # (only the software-pipelined tiles carry the per-sample scale: 24-tap
    # kernels on the GPU; otherwise v is formed by its own pass, as before)
    self.jvp_folds = bool(fold and self.x0.is_cuda and
                          lib.cg_swconv_check(ctypes.byref(self.jvp[0])) == 0)
    if fold and not self.jvp_folds:
      self.jvp = build(False)

  # -- schedules ------------------------------------------------------------
  @property
  def mixes_layer1(self):
    """forward(mix=alpha) forms layer 1 of the x^ segment from the real and fake
    segments' (the plan's X0 need not hold x^ then)."""
    return self.fwd_l1_pair is not None

  def forward(self, seed_backward=False, mix=None):
    """act[0] (already filled) -> d_out[:nB].  seed_backward: the head's pass
    over h5 also writes delta_5 = coef * w_d * lrelu'(h5) (the seed does not
    depend on the head's output: one launch and one read of h5 instead of two);
    backward_chain(seeded=True) then starts from it.  mix (f32 per sample of a
    segment; plans with mixes_layer1): the third segment is x^ = mix * real +
    (1 - mix) * fake -- its layer 1 is cg_lrelu_mix of the first two segments'."""
    st = _stream()
    net, ws = self.ws.net, self.ws
    if mix is not None and self.fwd_l1_pair is not None:
      B, lay = self.seg_size, net.layers[0]
      a1 = ws.act[1]
      _run_conv(self.fwd_l1_pair, st)
      _lib.call('cg_lrelu_mix', _p(a1[:B]), _p(a1[B:2 * B]), _p(mix),
                _p(a1[2 * B:3 * B]), B, lay.lout * lay.coutp, net.alpha, st)
      rest = self.fwd[1:]
    else:
      rest = self.fwd
    for d in rest:
      _run_conv(d, st)
    last = net.layers[-1]
    if seed_backward:
      _lib.call('cg_dense1_fwd_bwd', _p(ws.act[-1]), _p(net.dense_w),
                _p(net.dense_b), _p(ws.d_out), _p(self.coef), _p(ws.delta[-1]),
                self.nB, last.lout, last.cout, last.coutp, self.seg_size,
                net.alpha, st)
      return
    _lib.call('cg_dense1_fwd', _p(ws.act[-1]), _p(net.dense_w), _p(net.dense_b),
              _p(ws.d_out), self.nB, last.lout, last.cout, last.coutp, st)

  def backward_chain(self, seeded=False):
    """delta[5] = coef * w_d * lrelu'(h5) (unless forward(seed_backward=True)
    wrote it); then down to delta[1]; optional layer-1 input gradient into
    self.gin (bf16)."""
    st = _stream()
    net, ws = self.ws.net, self.ws
    last = net.layers[-1]
    if not seeded:
      _lib.call('cg_dense1_bwd', _p(net.dense_w), _p(self.coef), _p(ws.act[-1]),
                _p(ws.delta[-1]), self.nB, last.lout, last.cout, last.coutp,
                self.seg_size, net.alpha, st)
    for i, d in self.dgrad:
      _run_conv(d, st)
      lay = net.layers[i - 1]
      if i in self.side:
        side = self.side[i]
        _lib.call('cg_unshuffle_fixup', _p(side), _p(ws.act[i]),
                  _p(ws.delta[i]), _p(self.shifts[i - 1]), self.nB, lay.lout,
                  lay.coutp, self.seg_size, side.shape[1], net.alpha, st)
      else:
        _lib.call('cg_unshuffle_mask', _p(ws.e[i]), _p(ws.act[i]),
                  _p(ws.delta[i]), _p(self.shifts[i - 1]), self.nB, lay.lout,
                  lay.coutp, self.seg_size, net.alpha, st)
    if self.input_grad is not None:
      if self.sumsq is not None and not self.input_grad.rowsumsq_ws:
        self.sumsq.zero_()  # (the atomics form adds into it)
      _run_conv(self.input_grad, st)

  def jvp_forward(self):
    st = _stream()
    for d in self.jvp:
      _run_conv(d, st)

  def weight_grads(self, bias_rows):
    """Accumulate dW (all nB samples), db (first bias_rows samples; taken
    inside the wgrad kernel from the delta tiles it stages) and the dense head
    gradients into params.grad (caller zeroed it)."""
    st = _stream()
    net, ws = self.ws.net, self.ws
    for i, d in enumerate(self.wgrad):
      d.bias_rows = bias_rows * net.layers[i].lout
    _run_wgrads(self.wgrad, st)
    last = net.layers[-1]
    _lib.call('cg_dense1_wgrad', _p(ws.act[-1]), _p(self.coef),
              _p(self.bias_coef), _p(net.params.grad_views[-2]),
              _p(net.params.grad_views[-1]), self.nB, last.lout, last.cout,
              last.coutp, self.seg_size, _p(reduce_ws(net.device)), st)


# ===========================================================================
# Generator
# ===========================================================================
class GeneratorNet(object):
  """Dense -> LeakyReLU -> reshape(w, nd) -> 5 x [Conv1DTranspose(k, s=2) ->
  LayerNorm -> LeakyReLU] -> Dense(C) -> sigmoid  (calciumgan.py:22-103)."""

  def __init__(self, hp, device, rng):
    _require_gpu(hp)
    self.precision = precision_of(hp)
    self.h_dtype = act_dtype()  # torch dtype of activations / operands
    self.w0 = geo.validate_hparams(hp)
    self.hp = hp
    self.alpha = geo.activation_alpha(hp)  # x -> max(x, alpha x)
    self.device = device
    self.k = hp.kernel_size
    self.pl = geo.same_padding_left(self.k, hp.strides)
    self.nd = hp.noise_dim
    self.layers = geo.generator_layers(hp)
    self.C = hp.num_channels
    self.Cp = geo.pitch(self.C)
    self.L = hp.signal_shape[0]
    self.layer_norm = bool(hp.layer_norm)
    # BatchNormalization before the (optional) LayerNormalization of every block
    # (calciumgan.py:42-45); single rank only, ordered reductions only
    self.batch_norm = bool(getattr(hp, 'batch_norm', False))
    if self.batch_norm:
      from . import parallel
      if parallel.world_size() > 1:
        raise ValueError('calciumgan_amd: batch_norm under data parallelism needs '
                         'cross-rank batch statistics (not implemented)')
      if not DETERMINISTIC:
        raise ValueError('calciumgan_amd: batch_norm needs the ordered reductions '
                         '(CALCIUMGAN_DETERMINISTIC=1, the default)')
    self.normalize = bool(hp.normalize)
    nflat = self.w0 * self.nd
    shapes = [(self.nd, nflat), (nflat,)]
    init = [
        glorot_uniform(rng, (self.nd, nflat), self.nd, nflat),
        np.zeros(nflat, np.float32)
    ]
    self.idx_conv, self.idx_bn, self.idx_ln = [], [], []
    frozen = []
    for lay in self.layers:
      self.idx_conv.append(len(shapes))
      shapes += [(self.k, 1, lay.cout, lay.cin), (lay.cout,)]
      init += [
          glorot_uniform(rng, (self.k, 1, lay.cout, lay.cin),
                         self.k * lay.cout, self.k * lay.cin),
          np.zeros(lay.cout, np.float32)
      ]
      if self.batch_norm:
        # Keras order: gamma, beta, moving_mean, moving_variance
        self.idx_bn.append(len(shapes))
        frozen += [len(shapes) + 2, len(shapes) + 3]
        shapes += [(lay.cout,)] * 4
        init += [np.ones(lay.cout, np.float32), np.zeros(lay.cout, np.float32),
                 np.zeros(lay.cout, np.float32), np.ones(lay.cout, np.float32)]
      else:
        self.idx_bn.append(None)
      self.idx_ln.append(len(shapes) if self.layer_norm else None)
      if self.layer_norm:
        shapes += [(lay.cout,), (lay.cout,)]
        init += [np.ones(lay.cout, np.float32), np.zeros(lay.cout, np.float32)]
    self.idx_out = len(shapes)
    shapes += [(self.C, self.C), (self.C,)]
    init += [
        glorot_uniform(rng, (self.C, self.C), self.C, self.C),
        np.zeros(self.C, np.float32)
    ]
    self.params = FlatParams(shapes, device, frozen=frozen)
    self.params.set_weights(init)
    V = self.params.views
    # packed operands
    self.w_in = PackedOperand(V[0], [(0, 1, 0, nflat, 1)], self.nd, nflat,
                              self.nd, self.nd, 1)
    phases = _transpose_phases(self.k, self.pl)
    self.fwd_offs = [o for _, o in phases]
    self.w_fwd, self.w_dgrad = [], []
    for lay, ic in zip(self.layers, self.idx_conv):
      W = V[ic]
      ci, co = lay.cin, lay.cout
      ck = _ck_for(lay.cinp, 1, self.k // 2, lay.lin)
      self.w_fwd.append(
          PackedOperand(W, [(t0, -2, co * ci, 1, ci) for t0, _ in phases], ci,
                        co, lay.cinp, ck, self.k // 2))
      ck = _ck_for(lay.coutp, 2, self.k, lay.lin)
      self.w_dgrad.append(
          PackedOperand(W, [(0, 1, co * ci, ci, 1)], co, ci, lay.coutp, ck,
                        self.k, parity_major=True))
    Wo = V[self.idx_out]
    ck = _ck_for(self.Cp, 1, 1, self.L)
    self.w_out = PackedOperand(Wo, [(0, 1, 0, self.C, 1)], self.C, self.C,
                               self.Cp, ck, 1)
    self.w_out_t = PackedOperand(Wo, [(0, 1, 0, 1, self.C)], self.C, self.C,
                                 self.Cp, ck, 1)
    # the f32 output (B, L, Cf): the streaming Dense writes rows of C rounded
    # up to 8 channels (102 -> 104: 19 % fewer bytes than the bf16 pitch 128 for
    # it and for every pass that reads it); the swconv fallback keeps Cp
    self.streaming_out = geo.dense_streams(self.Cp) and ck == 32
    # its input gradient dh = dz W^T streams too (the LDS-panel form: pitches of
    # 128 and up)
    self.streaming_out_dgrad = self.streaming_out and self.Cp >= 128
    self.Cf = (self.C + 7) // 8 * 8 if self.streaming_out else self.Cp
    self._pack_plan = PackPlan(
        [self.w_in, self.w_out, self.w_out_t] + self.w_fwd + self.w_dgrad,
        device)
    self.repack()
    self._ws = {}

  def repack(self):
    self._pack_plan.run()

  def workspace(self, B, forward_only=False):
    """forward_only: no backward buffers / descriptors (the batched G(z) of
    all critic updates of a step)."""
    ws = self._ws.get((B, forward_only))
    if ws is None:
      ws = _GenWorkspace(self, B, forward_only)
      self._ws[(B, forward_only)] = ws
    return ws


class _GenWorkspace(object):

  def __init__(self, net, B, forward_only=False):
    dev = net.device
    self.net, self.B = net, B
    self.forward_only = forward_only
    z = lambda *s, dt=None: torch.zeros(*s, dtype=dt or act_dtype(), device=dev)
    nd, w0 = net.nd, net.w0
    V = net.params.views
    self.z = z(B, 1, nd)
    self.h = [z(B, w0, nd)] + [z(B, l.lout, l.coutp) for l in net.layers]
    self.ypre = [None] + [z(B, l.lout, l.coutp) for l in net.layers]
    self.mean = [None] + [z(B * l.lout, dt=torch.float32) for l in net.layers]
    self.rstd = [None] + [z(B * l.lout, dt=torch.float32) for l in net.layers]
    self.fake = z(B, net.L, net.Cf, dt=torch.float32)
    if net.batch_norm:
      # batch statistics of the last training-mode forward (the backward reads
      # them) and, with a LayerNormalization behind it, the tensor between the two
      self.bn_mean = [None] + [z(l.cout, dt=torch.float32) for l in net.layers]
      self.bn_var = [None] + [z(l.cout, dt=torch.float32) for l in net.layers]
      self.ybn = [None] + [z(B, l.lout, l.coutp) if net.layer_norm else None
                           for l in net.layers]
      if not forward_only and net.layer_norm:
        self.dybn = [None] + [z(B, l.lout, l.coutp) for l in net.layers]
    if not forward_only:  # backward buffers
      self.dz = z(B, net.L, net.Cp)
      self.dh = [z(B, w0, nd)] + [z(B, l.lout, l.coutp) for l in net.layers]
      self.dy = [z(B, 1, w0 * nd)] + [z(B, l.lout, l.coutp) for l in net.layers]
    k = net.k
    # ---- forward descriptors
    self.f_in = _conv_desc(self.z, net.w_in.buf, self.h[0], B, 1, nd, 1, 1, 0, 1,
                           w0 * nd, 1, w0 * nd, net.w_in.CK, bias=V[1],
                           epilogue=_lib.EPI_LRELU, alpha=net.alpha)
    self.f_conv = []
    self.f_conv_fwd_only = []  # G(z) of a critic update: nothing kept for backward
    self.ln_fused = []
    for i, (lay, ic) in enumerate(zip(net.layers, net.idx_conv)):
      op = net.w_fwd[i]
      il = net.idx_ln[i]
      normed = net.layer_norm or net.batch_norm
      dst = self.ypre[i + 1] if normed else self.h[i + 1]
      fuse = (net.layer_norm and not net.batch_norm and
              _ln_fusable(lay, op.CK, k // 2))
      self.ln_fused.append(fuse)
      self.f_conv.append(
          _conv_desc(self.h[i], op.buf, dst, B, lay.lin, lay.cinp, k // 2, 1,
                     net.fwd_offs[0], lay.lin, lay.cout, lay.lout, lay.coutp,
                     op.CK, y_stride=2, y_off=0, bias=V[ic + 1],
                     epilogue=_lib.EPI_NONE
                     if normed else _lib.EPI_LRELU, nphase=2,
                     w_phase_stride=op.elems,
                     off_phase_step=net.fwd_offs[1] - net.fwd_offs[0],
                     yoff_phase_step=1,
                     ln=(V[il], V[il + 1], self.h[i + 1], self.mean[i + 1],
                         self.rstd[i + 1]) if fuse else None, alpha=net.alpha))
      self.f_conv_fwd_only.append(
          _conv_desc(self.h[i], op.buf, dst, B, lay.lin, lay.cinp, k // 2, 1,
                     net.fwd_offs[0], lay.lin, lay.cout, lay.lout, lay.coutp,
                     op.CK, y_stride=2, y_off=0, bias=V[ic + 1], nphase=2,
                     w_phase_stride=op.elems,
                     off_phase_step=net.fwd_offs[1] - net.fwd_offs[0],
                     yoff_phase_step=1,
                     ln=(V[il], V[il + 1], self.h[i + 1], None, None),
                     alpha=net.alpha)
          if fuse else self.f_conv[-1])
    # (fallback for outputs wider than 128 channels; the streaming Dense of
    # forward() otherwise -- self.fake then has the narrower pitch Cf)
    self.f_out = None if net.streaming_out else _conv_desc(
        self.h[-1], net.w_out.buf, self.fake, B, net.L, net.Cp, 1, 1, 0, net.L,
        net.C, net.L, net.Cp, net.w_out.CK, bias=V[net.idx_out + 1],
        epilogue=_lib.EPI_SIGMOID if net.normalize else _lib.EPI_NONE,
        out_f32=True)
    if forward_only:
      return
    # ---- backward descriptors
    G = net.params.grad_views
    self.b_out_dgrad = None if net.streaming_out_dgrad else _conv_desc(
        self.dz, net.w_out_t.buf, self.dh[-1], B, net.L, net.Cp, 1, 1, 0, net.L,
        net.C, net.L, net.Cp, net.w_out_t.CK)
    # (partial tiles of the output Dense's weight gradient: cg_dense_wgrad)
    # (always for the ordered form; else only when dW has at most four 128 x 128
    # tiles, where a thousand workgroups' atomics on the same addresses serialise)
    need = _lib.load().cg_dense_wgrad_ws_elems(B * net.L, net.C, net.C)
    few_tiles = ((net.C + 127) // 128)**2 <= 4
    self.out_wgrad_ws = (torch.empty(need, dtype=torch.float32, device=dev)
                         if need > 0 and (DETERMINISTIC or few_tiles) else None)
    self.b_dgrad, self.b_wgrad = [], []
    for i, (lay, ic) in enumerate(zip(net.layers, net.idx_conv)):
      op = net.w_dgrad[i]
      self.b_dgrad.append(
          _conv_desc(self.dy[i + 1], op.buf, self.dh[i], B, lay.lout, lay.coutp,
                     k, 2, -net.pl, lay.lin, lay.cin, lay.lin, lay.cinp, op.CK,
                     w_parity_major=op.parity_major,
                     w_narrow_last=op.narrow_last))
      self.b_wgrad.append(
          _wgrad_desc(self.dy[i + 1], self.h[i], G[ic], B, lay.lout, lay.coutp,
                      lay.lin, lay.cinp, k, 2, -net.pl, lay.cout, lay.cin,
                      slot=i))
    self.b_in_wgrad = _wgrad_desc(self.z, self.dy[0], G[0], B, 1, nd, 1,
                                  w0 * nd, 1, 1, 0, nd, w0 * nd, slot='in')

  def can_interp(self, n):
    """Whether forward(interp=...) exists for this model: the streaming output
    Dense -- its register form at a 128-channel pitch (whole 16-row blocks), its
    LDS-panel form beyond (32-row blocks inside one update: B * L % 32 == 0)."""
    net = self.net
    if not (net.streaming_out and net.L % 16 == 0 and 1 <= n <= 8 and
            self.B % n == 0):
      return False
    if net.Cp == 128:
      return True
    return (net.Cp // 32 in (4, 8, 12, 16) and
            ((self.B // n) * net.L) % 32 == 0)

  def forward(self, z_f32, keep=True, training=True, interp=None):
    """z (B, nd) f32 device -> self.fake (B, L, Cf) f32 (first C channels).
    interp = (real f32 (B / n, L, C), alpha f32 (B), [x0_0 .. x0_{n-1}]): this
    workspace holds the n fake batches of a step's critic updates; instead of
    self.fake the critic's input buffers are written (returns None).
    keep=False: forward only (the fake batch of a critic update) -- the fused
    LayerNorm launches then skip the pre-activations and row statistics that
    only backward() reads.  training (BatchNormalization only): batch
    statistics + moving-average update, as every training=True call of the
    Keras model does; False: the moving statistics (validate / generate)."""
    net = self.net
    convs = self.f_conv if keep else self.f_conv_fwd_only
    st = _stream()
    _lib.call('cg_cast_pad', _p(z_f32), _p(self.z), self.B, net.nd, net.nd,
              net.nd, st)
    _run_conv(self.f_in, st)
    V = net.params.views
    for i, (lay, ic) in enumerate(zip(net.layers, net.idx_conv)):
      _run_conv(convs[i], st)
      rows = self.B * lay.lout
      ln_in = self.ypre[i + 1]
      if net.batch_norm:
        ib = net.idx_bn[i]
        if training:
          _lib.call('cg_bn_stats', _p(self.ypre[i + 1]), rows, lay.cout,
                    lay.coutp, _p(self.bn_mean[i + 1]), _p(self.bn_var[i + 1]),
                    _p(V[ib + 2]), _p(V[ib + 3]), BN_MOMENTUM,
                    _p(reduce_ws(net.device)), st)
          mean, var = self.bn_mean[i + 1], self.bn_var[i + 1]
        else:
          mean, var = V[ib + 2], V[ib + 3]
        out = self.ybn[i + 1] if net.layer_norm else self.h[i + 1]
        _lib.call('cg_bn_apply', _p(self.ypre[i + 1]), _p(mean), _p(var),
                  _p(V[ib]), _p(V[ib + 1]), _p(out), rows, lay.cout, lay.coutp,
                  BN_EPS, 1.0 if net.layer_norm else net.alpha, st)
        ln_in = out
      if net.layer_norm and not self.ln_fused[i]:
        il = net.idx_ln[i]
        _lib.call('cg_ln_lrelu_fwd', _p(ln_in), _p(V[il]),
                  _p(V[il + 1]), _p(self.h[i + 1]), _p(self.mean[i + 1]),
                  _p(self.rstd[i + 1]), rows, lay.cout, lay.coutp,
                  LN_EPS, net.alpha, st)
    if interp is not None:
      # the fake batches of all critic updates of a step: Dense + sigmoid with the
      # interpolation and the packing of the critic's inputs in its epilogue --
      # x0s[k] <- [real | fake_k | x^_k] (cg_dense_rows_interp); no f32 fake batch
      real, alpha, x0s = interp
      n = len(x0s)
      Bu = self.B // n
      ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in x0s])
      _lib.call('cg_dense_rows_interp', _p(self.h[-1]), _p(net.w_out.buf),
                _p(V[net.idx_out + 1]), _p(real), _p(alpha), ptrs, n, Bu, net.L,
                net.Cp, net.C, net.C, net.Cp,
                _lib.EPI_SIGMOID if net.normalize else _lib.EPI_NONE, st)
      return None
    if net.streaming_out:
      # HBM-bound per-timestep Dense (+ sigmoid): the streaming kernel
      _lib.call('cg_dense_rows', _p(self.h[-1]), _p(net.w_out.buf),
                _p(V[net.idx_out + 1]), _p(self.fake), self.B * net.L, net.Cp,
                net.C, net.Cf,
                _lib.EPI_SIGMOID if net.normalize else _lib.EPI_NONE, st)
    else:
      _run_conv(self.f_out, st)
    return self.fake

  def backward(self, dfake):
    """dfake (B, L, Cp) bf16 = d loss / d fake -> accumulates every generator
    gradient into params.grad (caller zeroed it)."""
    net = self.net
    st = _stream()
    # the bias / gamma / beta column sums below are read by nobody before the
    # optimizer: their seven finishing launches become one (cg_finish_defer),
    # each reduction with its own workspace region.  (BatchNormalization's
    # backward reads its sums at once: not deferred.)
    defer = (_DEFER_FINISH and DETERMINISTIC and net.layer_norm and
             not net.batch_norm and self.h[0].is_cuda)
    if defer:
      regions = reduce_ws_regions(net.device, len(net.layers) + 2)
      _lib.load().cg_finish_defer(1)
      try:
        self._backward(dfake, st, lambda k: _p(regions[k]))
      finally:
        _lib.call('cg_finish_flush', st)
    else:
      rws = _p(reduce_ws(net.device))
      self._backward(dfake, st, lambda k: rws)

  def _backward(self, dfake, st, rws_of):
    net = self.net
    G = net.params.grad_views
    V = net.params.views
    rows = self.B * net.L
    if net.normalize:
      _lib.call('cg_sigmoid_bwd', _p(dfake), _p(self.fake), _p(self.dz),
                rows, net.C, net.Cf, net.Cp, st)
    else:
      self.dz.copy_(dfake.view_as(self.dz))  # linear output: dz = dfake
    # dW of the per-timestep Dense: the streaming kernel (the generic cg_wgrad
    # path, taps = 1, reads its operands in 64-byte slivers: 92 -> 31 us at cfg2)
    _lib.call('cg_dense_wgrad', _p(self.h[-1]), _p(self.dz), _p(G[net.idx_out]),
              rows, net.Cp, net.Cp, net.C, net.C, _p(self.out_wgrad_ws),
              0 if self.out_wgrad_ws is None else self.out_wgrad_ws.numel(), st)
    nl = len(net.layers)
    _lib.call('cg_colsum', _p(self.dz), _p(G[net.idx_out + 1]), rows, net.C,
              net.Cp, rws_of(nl), st)
    if net.streaming_out_dgrad:
      _lib.call('cg_dense_rows_act', _p(self.dz), _p(net.w_out_t.buf),
                _p(self.dh[-1]), rows, net.Cp, net.C, net.Cp, st)
    else:
      _run_conv(self.b_out_dgrad, st)
    for i in range(len(net.layers) - 1, -1, -1):
      lay, ic = net.layers[i], net.idx_conv[i]
      n = self.B * lay.lout
      if net.batch_norm:
        ib = net.idx_bn[i]
        dout, hmask, act = self.dh[i + 1], self.h[i + 1], 1
        if net.layer_norm:
          il = net.idx_ln[i]
          _lib.call('cg_ln_lrelu_bwd', _p(self.dh[i + 1]), _p(self.h[i + 1]),
                    _p(self.ybn[i + 1]), _p(self.mean[i + 1]),
                    _p(self.rstd[i + 1]), _p(V[il]), _p(self.dybn[i + 1]),
                    _p(G[il]), _p(G[il + 1]), None, n, lay.cout, lay.coutp,
                    net.alpha, rws_of(i), st)
          dout, hmask, act = self.dybn[i + 1], None, 0
        _lib.call('cg_bn_bwd', _p(dout), _p(hmask), _p(self.ypre[i + 1]),
                  _p(self.bn_mean[i + 1]), _p(self.bn_var[i + 1]), _p(V[ib]),
                  _p(self.dy[i + 1]), _p(G[ib]), _p(G[ib + 1]), n, lay.cout,
                  lay.coutp, BN_EPS, net.alpha, act, rws_of(i), st)
        # (the conv bias gradient: BatchNormalization removes the column mean, so
        # this sum is zero up to rounding -- as the reference's autodiff gives it)
        _lib.call('cg_colsum', _p(self.dy[i + 1]), _p(G[ic + 1]), n, lay.cout,
                  lay.coutp, rws_of(i), st)
      elif net.layer_norm:
        il = net.idx_ln[i]
        _lib.call('cg_ln_lrelu_bwd', _p(self.dh[i + 1]), _p(self.h[i + 1]),
                  _p(self.ypre[i + 1]), _p(self.mean[i + 1]),
                  _p(self.rstd[i + 1]), _p(V[il]), _p(self.dy[i + 1]),
                  _p(G[il]), _p(G[il + 1]), _p(G[ic + 1]), n, lay.cout,
                  lay.coutp, net.alpha, rws_of(i), st)
      else:
        _lib.call('cg_lrelu_bwd', _p(self.dh[i + 1]), _p(self.h[i + 1]),
                  _p(self.dy[i + 1]), n * lay.coutp, net.alpha, st)
        _lib.call('cg_colsum', _p(self.dy[i + 1]), _p(G[ic + 1]), n, lay.cout,
                  lay.coutp, rws_of(i), st)
      _run_conv(self.b_dgrad[i], st)
    nflat = net.w0 * net.nd
    _lib.call('cg_lrelu_bwd', _p(self.dh[0]), _p(self.h[0]), _p(self.dy[0]),
              self.B * nflat, net.alpha, st)
    _run_wgrad(self.b_in_wgrad, st)
    _lib.call('cg_colsum', _p(self.dy[0]), _p(G[1]), self.B, nflat, nflat,
              rws_of(nl + 1), st)
    # the conv-transpose weight gradients read h[i] / dy[i+1], which the chain
    # above only produced: all of them together, at the end
    _run_wgrads(self.b_wgrad, st)


def adam_lr_t(step, lr, beta1=0.9, beta2=0.999):
  """Keras Adam's bias-corrected step size at 1-based iteration `step`."""
  return lr * math.sqrt(1.0 - beta2**step) / (1.0 - beta1**step)


LOSS_SCALE_INIT = 2.0**15     # tf DynamicLossScale defaults
LOSS_SCALE_INTERVAL = 2000


def new_loss_scale_state(device):
  """Device state of one dynamic loss scaler: [S, consecutive finite updates,
  applied Adam steps, gradients-finite flag] (include/calciumgan_hip.h)."""
  return torch.tensor([LOSS_SCALE_INIT, 0.0, 0.0, 1.0], dtype=torch.float32,
                      device=device)


def adam_update_scaled(params, lr, ls, grad_scale=1.0, beta1=0.9, beta2=0.999,
                       eps=1e-7, interval=LOSS_SCALE_INTERVAL):
  """LossScaleOptimizer(Adam).apply_gradients on gradients that carry the
  loss scale ls[0] (optimizer.py:23-34): finite check, unscale + Keras Adam
  (skipped when a gradient is inf / nan), then the scale's own update.  All on
  the device: nothing here depends on a host-side step count."""
  st = _stream()
  _lib.call('cg_grad_finite', _p(params.grad), params.numel, _p(ls), st)
  _lib.call('cg_adam_scaled', _p(params.data), _p(params.grad), _p(params.m),
            _p(params.v), params.numel, lr, beta1, beta2, eps, grad_scale,
            _p(ls), st)
  _lib.call('cg_loss_scale_update', _p(ls), interval, st)


def adam_update(params, step, lr, grad_scale=1.0, beta1=0.9, beta2=0.999,
                eps=1e-7, lr_t_dev=None):
  """tf.keras.optimizers.Adam dense update (gan/algorithms/optimizer.py:9,
  :31-34); `step` is the 1-based iteration count.  lr_t_dev (device scalar)
  overrides the host-computed step size (captured-graph replay)."""
  lr_t = adam_lr_t(step, lr, beta1, beta2)
  _lib.call('cg_adam', _p(params.data), _p(params.grad), _p(params.m),
            _p(params.v), params.numel, lr_t, beta1, beta2, eps, grad_scale,
            _p(lr_t_dev), _stream())
