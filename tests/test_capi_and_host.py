"""CPU tests: the C-ABI library builds for gfx950, loads, and exports every
symbol include/calciumgan_hip.h declares; host-side shape logic; the registry /
error surface; loud failure without a device (no CPU fallback)."""
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle as O
from calciumgan_amd import _lib
from calciumgan_amd import build as cg_build
from calciumgan_amd import geometry as geo
from calciumgan_amd import nets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'calciumgan_hip.h')


def _declared():
  src = open(HEADER).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(?:int|long long)\s+(cg_\w+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
  cg_build.build(verbose=False)
  lib = _lib.load()
  names = _declared()
  assert len(names) >= 20
  for n in names:
    assert hasattr(lib, n), 'missing export ' + n
    assert n in _lib.SIGNATURES, 'no ctypes signature for ' + n
  assert sorted(_lib.SIGNATURES) == names
  want = int(re.search(r'#define CG_ABI_VERSION (\d+)', open(HEADER).read()).group(1))
  assert lib.cg_abi_version() == want


def test_graft_entry_build_runs():
  """The driver's build check: compiles (or finds) every extension, loads the
  C ABI and checks its version against the header."""
  import __graft_entry__ as entry
  entry.build()


def test_descriptor_layouts_match_library():
  """ctypes mirrors vs the compiled structs (also enforced by _lib.load)."""
  import ctypes
  lib = _lib.load()
  for which, cls in enumerate((_lib.ConvDesc, _lib.PackDesc, _lib.WgradDesc)):
    assert lib.cg_struct_size(which) == ctypes.sizeof(cls), cls.__name__
  assert lib.cg_struct_size(3) == -1


def test_packed_elems_host_formula():
  lib = _lib.load()
  # N rows padded to 128, K padded to 16 groups of 8 per channel chunk
  assert lib.cg_packed_elems(64, 24, 104, 104) == 128 * 320 * 8
  assert lib.cg_packed_elems(102, 1, 104, 104) == 128 * 16 * 8
  assert lib.cg_packed_elems(320, 12, 256, 64) == 384 * 4 * 96 * 8
  assert lib.cg_packed_elems(64, 24, 100, 50) == -1  # CK not multiple of 8


def test_tile_table_matches_library():
  lib = _lib.load()
  import ctypes
  tiles = sorted(list(_lib.TILES) + list(_lib.SWP_TILES))
  assert tiles == list(range(len(tiles)))
  for tile in tiles:
    r, c = ctypes.c_int(), ctypes.c_int()
    assert lib.cg_tile_shape(tile, ctypes.byref(r), ctypes.byref(c)) == 0
    assert (r.value, c.value) == _lib.tile_shape(tile)
  assert lib.cg_tile_shape(len(tiles), None, None) == _lib.CG_EINVAL


def _dry_conv_desc(stride, taps, nB, Lx, Cx, N, tile, epilogue=0):
  """A descriptor for cg_swconv_check only (no pointer is dereferenced)."""
  import ctypes
  d = _lib.ConvDesc()
  d.x = d.w = d.y = 0x1000
  d.nB, d.Lx, d.Cx, d.seg_size = nB, Lx, Cx, nB
  d.taps, d.stride = taps, stride
  d.Lu = Lx // 2 if stride == 2 else Lx
  d.off = -geo.same_padding_left(taps, 2) if stride == 2 else -(taps // 2 - 1)
  d.N = N
  d.nphase = 1 if stride == 2 else 2
  d.Ly, d.Cy = d.Lu * d.nphase, geo.pitch(N)
  d.y_stride, d.y_off, d.CK = d.nphase, 0, 32
  d.epilogue, d.alpha = epilogue, 0.3
  if epilogue == _lib.EPI_LN_LRELU:
    d.ln_gamma = d.ln_beta = d.ln_h = d.ln_mean = d.ln_rstd = 0x1000
    d.ln_eps = 1e-3
  d.w_phase_stride, d.off_phase_step, d.yoff_phase_step = 1 << 20, 1, 1
  d.tile, d.stage_ksteps = tile, 2
  d.w_parity_major = int(stride == 2)
  return d, ctypes.byref(d)


def test_swconv_check_admits_the_software_pipelined_tiles():
  """cg_swconv_check (the dry run nets.py validates tile choices with) on the
  CPU: the 32-row wave tiles are admitted at the cfg2 layer shapes and refused
  where a pass is not 12 taps or a wave's rows would straddle samples."""
  lib = _lib.load()
  ok = lambda *a, **k: lib.cg_swconv_check(_dry_conv_desc(*a, **k)[1])
  for tile in (13, 14, 15):
    assert ok(2, 24, 384, 2048, 128, 192, tile) == 0      # critic forward
    assert ok(1, 12, 128, 128, 320, 256, tile) == 0       # input gradient / convT
  assert ok(1, 12, 640, 1024, 128, 102, 15, epilogue=_lib.EPI_LN_LRELU) == 0
  assert ok(1, 12, 640, 1024, 128, 102, 14, epilogue=_lib.EPI_LN_LRELU) == _lib.CG_EINVAL
  assert ok(1, 12, 640, 256, 256, 192, 15, epilogue=_lib.EPI_LN_LRELU) == _lib.CG_EINVAL
  assert ok(2, 8, 4, 256, 64, 64, 14) == _lib.CG_EINVAL      # 4 taps per parity
  assert ok(2, 24, 4, 32, 64, 64, 14) == _lib.CG_EINVAL      # 16-row samples
  rows, cols = ctypes_int(), ctypes_int()
  for tile, shape in _lib.SWP_TILES.items():
    assert lib.cg_tile_shape(tile, rows.ref, cols.ref) == 0
    assert (rows.value, cols.value) == shape


class ctypes_int(object):
  def __init__(self):
    import ctypes
    self._v = ctypes.c_int(0)
    self.ref = ctypes.byref(self._v)

  @property
  def value(self):
    return self._v.value


def test_no_cpu_fallback_without_device():
  if torch.cuda.is_available():
    pytest.skip('device present')
  hp = O.make_hparams(64, 6, 8)
  with pytest.raises(RuntimeError):
    nets.DiscriminatorNet(hp, torch.device('cpu'), np.random.RandomState(0))
  from calciumgan_amd.gan.models import get_models
  hp.verbose = 0
  with pytest.raises(RuntimeError):
    get_models(hp, None)


def test_missing_library_fails_loudly(monkeypatch):
  monkeypatch.setattr(_lib, '_libs', {})
  monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libcalciumgan_hip.so')
  monkeypatch.setattr(_lib, 'LIB_PATH_F16', '/nonexistent/libcalciumgan_hip_f16.so')
  with pytest.raises(_lib.HipLibraryError):
    _lib.load()
  with pytest.raises(_lib.HipLibraryError):
    _lib.use('f16')
  assert _lib.active() == 'bf16'


def test_both_precision_builds_load_and_export_the_abi():
  """libcalciumgan_hip.so (bf16 activations) and libcalciumgan_hip_f16.so
  (-DCG_ACT_F16=1, mixed_float16) export every symbol of the header and say
  which storage type they compute with."""
  bf = _lib.load('bf16')
  hf = _lib.load('f16')
  assert bf.cg_act_dtype() == _lib.DTYPE_BF16
  assert hf.cg_act_dtype() == _lib.DTYPE_F16
  assert bf.cg_abi_version() == hf.cg_abi_version()
  for name in _lib.SIGNATURES:
    assert hasattr(hf, name) and hasattr(bf, name)
  with pytest.raises(ValueError):
    _lib.use('fp8')


def test_registries_mirror_reference_contract(capsys):
  from calciumgan_amd.gan.algorithms import registry as areg
  from calciumgan_amd.gan.models import registry as mreg
  import calciumgan_amd.gan.algorithms  # noqa: F401  (registers)
  import calciumgan_amd.gan.models  # noqa: F401
  assert 'calciumgan' in mreg._MODELS
  assert set(areg._ALGORITHMS) >= {'gan', 'wgan-gp'}
  # unknown names print and exit (gan/models/registry.py:17-19)
  with pytest.raises(SystemExit):
    mreg.get_models(SimpleNamespace(model='wavegan'), None)
  assert 'not found' in capsys.readouterr().out
  with pytest.raises(SystemExit):
    areg.get_algorithm(SimpleNamespace(algorithm='lswgan'), None, None, None)


def test_layer_tables_match_survey_appendix_b():
  hp = O.make_hparams(2048, 102, 64)
  d = geo.discriminator_layers(hp)
  assert [(l.cin, l.cout, l.lin, l.lout) for l in d] == [
      (102, 64, 2048, 1024), (64, 128, 1024, 512), (128, 192, 512, 256),
      (192, 256, 256, 128), (256, 320, 128, 64)]
  g = geo.generator_layers(hp)
  assert [(l.cin, l.cout, l.lin, l.lout) for l in g] == [
      (32, 320, 64, 128), (320, 256, 128, 256), (256, 192, 256, 512),
      (192, 128, 512, 1024), (128, 102, 1024, 2048)]
  assert d[0].cinp == 128 and g[-1].coutp == 128
  assert geo.pitch(6) == 32 and geo.pitch(320) == 320 and geo.pitch(130) == 160


def test_validate_hparams_errors():
  with pytest.raises(ValueError):  # calciumgan.py:17-18
    geo.validate_hparams(O.make_hparams(100, 4, 8))
  with pytest.raises(ValueError):
    geo.validate_hparams(O.make_hparams(96, 4, 8))  # L/32 = 3 unsupported
  with pytest.raises(ValueError):
    geo.validate_hparams(O.make_hparams(64, 4, 8, strides=3))
  with pytest.raises(ValueError):
    geo.validate_hparams(O.make_hparams(64, 4, 8, m=4))  # reflect pad >= len
  assert geo.validate_hparams(O.make_hparams(2048, 102, 64, m=10)) == 64


@pytest.mark.parametrize('k', [24, 8, 2])
def test_transpose_phase_walk_equals_conv_transpose(k):
  """Host-side phase/tap/offset spec used for dgrad + Conv1DTranspose, checked
  in numpy against the oracle: out[2u+p] = sum_jj x[u+off_p+jj] W[tap0_p-2jj]."""
  rng = np.random.RandomState(0)
  L, Ci, Co = 8, 3, 4
  x = rng.randn(1, L, Ci)
  Wt = rng.randn(k, 1, Co, Ci)
  ref = O.conv1d_transpose_same(
      torch.tensor(x), torch.tensor(Wt), None, 2).numpy()[0]
  pl = geo.same_padding_left(k, 2)
  out = np.zeros((2 * L, Co))
  for p, (tap0, off) in enumerate(nets._transpose_phases(k, pl)):
    for u in range(L):
      for jj in range(k // 2):
        i = u + off + jj
        if 0 <= i < L:
          out[2 * u + p] += Wt[tap0 - 2 * jj, 0] @ x[0, i]
  np.testing.assert_allclose(out, ref, atol=1e-12)


def test_ck_choice_fits_lds():
  for cx, stride, taps, lu in [(104, 2, 24, 1024), (64, 2, 24, 512),
                               (320, 1, 12, 64), (256, 2, 24, 64),
                               (32, 1, 1, 1), (104, 1, 1, 2048)]:
    ck = nets._ck_for(cx, stride, taps, lu)
    assert cx % ck == 0 and ck % 8 == 0 and ck >= 32


def test_smooth_activations_are_rejected_with_the_reason():
  """activation_fn (gan/models/utils.py:6-8) accepts any Keras name; the HIP
  schedule covers the piecewise-linear ones and says why it refuses the rest."""
  import pytest
  from calciumgan_amd import geometry as geo
  import oracle as O
  hp = O.make_hparams(256, 16, 8)
  for name, alpha in (('leakyrelu', 0.3), ('relu', 0.0), ('linear', 1.0)):
    hp.activation = name
    assert geo.activation_alpha(hp) == alpha
    geo.validate_hparams(hp)
  hp.activation = 'tanh'
  with pytest.raises(ValueError, match='piecewise-linear'):
    geo.validate_hparams(hp)
  # the oracle itself follows activation_fn for every name
  import torch
  x = torch.tensor([-1.0, 0.0, 2.0])
  assert torch.equal(O.activation_fn('relu')(x), torch.tensor([0.0, 0.0, 2.0]))
  assert torch.allclose(O.activation_fn('leakyrelu')(x), torch.tensor([-0.3, 0.0, 2.0]))
  assert torch.allclose(O.activation_fn('tanh')(x), torch.tanh(x))
