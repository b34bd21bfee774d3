"""ctypes binding of the C ABI in include/calciumgan_hip.h.

The HIP library is the ONLY compute path of this package: if it is missing or
fails to load, every op raises (there is no CPU / eager fallback).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CALCIUMGAN_HIP_LIB points at another build of the same C ABI (development)
LIB_PATH = os.environ.get('CALCIUMGAN_HIP_LIB') or os.path.join(
    _HERE, 'csrc', 'libcalciumgan_hip.so')
# the same sources with fp16 activations: the reference's mixed_float16 mode
LIB_PATH_F16 = os.environ.get('CALCIUMGAN_HIP_LIB_F16') or os.path.join(
    _HERE, 'csrc', 'libcalciumgan_hip_f16.so')
DTYPE_BF16, DTYPE_F16 = 0, 1

CG_EINVAL = 100001
EPI_NONE, EPI_LRELU, EPI_MASK, EPI_SIGMOID, EPI_LN_LRELU = 0, 1, 2, 3, 4
# CG_TILE_*: value -> (rows, cols, mfma rows)
TILES = {0: (256, 64, 16), 1: (64, 64, 16), 2: (128, 64, 16),
         3: (256, 64, 32), 4: (128, 64, 32), 5: (256, 128, 32),
         6: (128, 128, 32), 7: (256, 128, 16),
         8: (128, 128, 16)}
# CG_TILE_SWP_*: value -> (rows, cols); swconv_swp.hip, two waves per SIMD
SWP_TILES = {9: (512, 64), 10: (256, 64), 11: (256, 128), 12: (128, 128),
             13: (128, 64), 14: (256, 64), 15: (128, 128)}


def tile_shape(tile):
  """(rows, cols) of any CG_TILE_* value."""
  return TILES[tile][:2] if tile in TILES else SWP_TILES[tile]


c_vp = C.c_void_p
c_i = C.c_int
c_ll = C.c_longlong
c_f = C.c_float


class ConvDesc(C.Structure):
  """struct cg_conv_desc."""
  _fields_ = [
      ('x', c_vp), ('w', c_vp), ('y', c_vp), ('bias', c_vp), ('mask_src', c_vp),
      ('shifts', c_vp),
      ('nB', c_i), ('Lx', c_i), ('Cx', c_i), ('seg_size', c_i),
      ('taps', c_i), ('stride', c_i), ('off', c_i), ('Lu', c_i),
      ('N', c_i), ('Ly', c_i), ('Cy', c_i), ('y_stride', c_i), ('y_off', c_i),
      ('CK', c_i),
      ('epilogue', c_i), ('out_f32', c_i),
      ('alpha', c_f),
      ('nphase', c_i),
      ('w_phase_stride', c_ll),
      ('off_phase_step', c_i), ('yoff_phase_step', c_i),
      ('tile', c_i),
      ('stage_ksteps', c_i),
      ('rowsumsq', c_vp),
      ('w_parity_major', c_i),
      ('split_parity', c_i),
      ('ln_gamma', c_vp), ('ln_beta', c_vp), ('ln_h', c_vp),
      ('ln_mean', c_vp), ('ln_rstd', c_vp), ('ln_eps', c_f),
      ('w_narrow_last', c_i),
      ('out_shifts', c_vp), ('out_seg_size', c_i), ('side', c_vp),
      ('side_rows', c_i),
      ('ksplit', c_i), ('split_ws', c_vp), ('split_ws_elems', c_ll),
      ('row_scale', c_vp),
      ('rowsumsq_ws', c_vp), ('rowsumsq_ws_elems', c_ll),
      ('rowsumsq_defer', c_i),
  ]


class PackDesc(C.Structure):
  """struct cg_pack_desc."""
  _fields_ = [
      ('src', c_vp), ('dst', c_vp),
      ('taps', c_i), ('tap0', c_i), ('tap_step', c_i),
      ('s_tap', c_ll), ('s_c', c_ll), ('s_n', c_ll),
      ('C_real', c_i), ('N_real', c_i), ('Cx', c_i), ('CK', c_i),
      ('parity_major', c_i),
      ('narrow_last', c_i),
  ]


class WgradDesc(C.Structure):
  """struct cg_wgrad_desc."""
  _fields_ = [
      ('x', c_vp), ('g', c_vp), ('dw', c_vp), ('shifts', c_vp),
      ('nB', c_i), ('Lx', c_i), ('Cx', c_i), ('seg_size', c_i),
      ('Lu', c_i), ('Cg', c_i),
      ('taps', c_i), ('stride', c_i), ('off', c_i),
      ('Cx_real', c_i), ('Cg_real', c_i),
      ('nsplit', c_i),
      ('tile_rows', c_i),
      ('no_xcd_group', c_i),
      ('classic_staging', c_i),
      ('dbias', c_vp),
      ('bias_rows', c_ll),
      ('partials', c_vp),
      ('partials_elems', c_ll),
      ('store', c_i),
  ]


# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    'cg_abi_version': [],
    'cg_act_dtype': [],
    'cg_struct_size': [c_i],
    'cg_tile_shape': [c_i, C.POINTER(c_i), C.POINTER(c_i)],
    'cg_debug_lean_epilogue': [c_i],
    'cg_debug_wgrad_flex': [c_i],
    'cg_wgrad_flex_plan': [C.POINTER(WgradDesc), c_i, c_i, C.POINTER(c_i), c_ll,
                           C.POINTER(c_i)],
    'cg_profile_enable': [c_i],
    'cg_profile_collect': [C.POINTER(c_f), C.POINTER(c_i), c_i],
    'cg_swconv': [C.POINTER(ConvDesc), c_vp],
    'cg_swconv_check': [C.POINTER(ConvDesc)],
    'cg_rowsumsq_ws_elems': [C.POINTER(ConvDesc)],
    'cg_reduce_ws_elems': [],
    'cg_finish_defer': [c_i],
    'cg_finish_flush': [c_vp],
    'cg_dense_rows': [c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_i, c_i, c_vp],
    'cg_dense_rows_interp': [c_vp, c_vp, c_vp, c_vp, c_vp, C.POINTER(c_vp), c_i, c_i,
                             c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    'cg_dense_rows_act': [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_i, c_vp],
    'cg_dense_wgrad': [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_i, c_i, c_vp, c_ll,
                       c_vp],
    'cg_dense_wgrad_ws_elems': [c_ll, c_i, c_i],
    'cg_packed_elems': [c_i, c_i, c_i, c_i],
    'cg_pack_weights': [C.POINTER(PackDesc), c_vp],
    'cg_pack_plan_bytes': [c_i, c_ll],
    'cg_pack_plan_build': [C.POINTER(PackDesc), c_i, c_vp, c_ll],
    'cg_pack_batched': [c_vp, c_i, c_ll, c_vp],
    'cg_wgrad': [C.POINTER(WgradDesc), c_vp],
    'cg_wgrad_batched': [C.POINTER(WgradDesc), c_i, c_vp],
    'cg_wgrad_partials_elems': [C.POINTER(WgradDesc)],
    'cg_ln_lrelu_fwd': [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_f,
                        c_f, c_vp],
    'cg_ln_lrelu_bwd': [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                        c_vp, c_ll, c_i, c_i, c_f, c_vp, c_vp],
    'cg_bn_stats': [c_vp, c_ll, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_f, c_vp, c_vp],
    'cg_bn_apply': [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_f, c_f,
                    c_vp],
    'cg_bn_bwd': [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_i,
                  c_i, c_f, c_f, c_i, c_vp, c_vp],
    'cg_dense1_fwd': [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    'cg_dense1_fwd_bwd': [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i,
                          c_i, c_f, c_vp],
    'cg_gp_critic_loss': [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_f, c_i, c_f, c_vp],
    'cg_gp_loss_scale': [c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_f, c_f,
                         c_vp, c_vp, c_ll, c_vp],
    'cg_dense1_bwd': [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f,
                      c_vp],
    'cg_dense1_wgrad': [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i,
                        c_vp, c_vp],
    'cg_unshuffle_mask': [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f,
                          c_vp],
    'cg_unshuffle_fixup': [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f,
                           c_vp],
    'cg_interp_pack': [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i,
                       c_i, c_vp],
    'cg_cast_pad': [c_vp, c_vp, c_ll, c_i, c_i, c_i, c_vp],
    'cg_rownorm': [c_vp, c_vp, c_i, c_ll, c_vp, c_vp],
    'cg_gp_finalize': [c_vp, c_vp, c_vp, c_i, c_f, c_i, c_vp],
    'cg_scale_rows': [c_vp, c_vp, c_vp, c_i, c_ll, c_vp],
    'cg_critic_loss': [c_vp, c_vp, c_f, c_vp, c_i, c_vp],
    'cg_neg_mean': [c_vp, c_vp, c_i, c_vp],
    'cg_colsum': [c_vp, c_vp, c_ll, c_i, c_i, c_vp, c_vp],
    'cg_sigmoid_bwd': [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_i, c_vp],
    'cg_lrelu_bwd': [c_vp, c_vp, c_vp, c_ll, c_f, c_vp],
    'cg_lrelu_mix': [c_vp, c_vp, c_vp, c_vp, c_i, c_ll, c_f, c_vp],
    'cg_adam': [c_vp, c_vp, c_vp, c_vp, c_ll, c_f, c_f, c_f, c_f, c_f, c_vp,
                c_vp],
    'cg_grad_finite': [c_vp, c_ll, c_vp, c_vp],
    'cg_adam_scaled': [c_vp, c_vp, c_vp, c_vp, c_ll, c_f, c_f, c_f, c_f, c_f,
                       c_vp, c_vp],
    'cg_loss_scale_update': [c_vp, c_i, c_vp],
    'cg_signal_metrics': [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_i, c_f, c_f,
                          c_vp, c_vp],
    'cg_step_outputs': [c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_vp],
}
_RESTYPES = {'cg_packed_elems': c_ll, 'cg_pack_plan_bytes': c_ll,
             'cg_pack_plan_build': c_ll, 'cg_wgrad_partials_elems': c_ll,
             'cg_dense_wgrad_ws_elems': c_ll, 'cg_rowsumsq_ws_elems': c_ll,
             'cg_reduce_ws_elems': c_ll, 'cg_wgrad_flex_plan': c_ll}

_libs = {}       # precision -> ctypes handle
_active = 'bf16'  # precision of the library `call` / `load()` address


class HipLibraryError(RuntimeError):
  pass


def use(precision):
  """Select the build every later `call` / `load()` goes to: 'bf16' (default)
  or 'f16' (mixed_float16).  The two builds keep separate kernels and tuning
  tables; objects created under one precision must be driven under it (the
  algorithm object checks)."""
  global _active
  if precision not in ('bf16', 'f16'):
    raise ValueError("precision must be 'bf16' or 'f16'")
  load(precision)
  _active = precision


def active():
  return _active


def load(precision=None):
  """Load (once per precision) and return the ctypes handle.  Raises
  HipLibraryError when the library has not been built -- there is deliberately
  no fallback path."""
  precision = precision or _active
  if precision in _libs:
    return _libs[precision]
  # torch first: its wheel bundles a HIP runtime, and the library must bind to
  # THAT copy -- loaded before torch, it pulled in /opt/rocm's libamdhip64 and
  # the process ended up with two runtimes (first launch: hipErrorNoDevice)
  import torch  # noqa: F401
  path = LIB_PATH_F16 if precision == 'f16' else LIB_PATH
  if not os.path.exists(path):
    raise HipLibraryError(
        'calciumgan_amd: {} not found. Build it with `python -m '
        'calciumgan_amd.build` (hipcc --offload-arch=gfx950); there is no '
        'CPU fallback.'.format(path))
  try:
    lib = C.CDLL(path)
  except OSError as e:
    raise HipLibraryError('calciumgan_amd: cannot load {}: {}'.format(path, e))
  for name, argtypes in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError if the symbol is missing
    fn.argtypes = argtypes
    fn.restype = _RESTYPES.get(name, c_i)
  # the ctypes mirrors of the descriptor structs must match the compiled header
  for which, cls in enumerate((ConvDesc, PackDesc, WgradDesc)):
    if lib.cg_struct_size(which) != C.sizeof(cls):
      raise HipLibraryError(
          'calciumgan_amd: {} is {} bytes here, {} in {} -- stale build or '
          'binding (rebuild with `python -m calciumgan_amd.build`)'.format(
              cls.__name__, C.sizeof(cls), lib.cg_struct_size(which), path))
  want = DTYPE_F16 if precision == 'f16' else DTYPE_BF16
  if lib.cg_act_dtype() != want:
    raise HipLibraryError(
        'calciumgan_amd: {} computes with dtype {} (wanted {})'.format(
            path, lib.cg_act_dtype(), want))
  _libs[precision] = lib
  return lib


def check(rc, what):
  if rc != 0:
    if rc == CG_EINVAL:
      raise ValueError('{}: unsupported shape/arguments (CG_EINVAL)'.format(what))
    raise RuntimeError('{}: HIP error {}'.format(what, rc))


def call(name, *args):
  """Invoke an int-returning entry point of the active build and raise on a
  non-zero code."""
  rc = getattr(load(), name)(*args)
  check(rc, name)
