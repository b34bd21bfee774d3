"""Host-side shape logic of the CalciumGAN stack (pure python, CPU-testable).

Restates the shape rules of gan/models/calciumgan.py (generator :22-103,
discriminator :141-192, calculate_noise_shape :15-19) and adds the MI355X
layout decisions: channel pitch, channel-chunk (CK) of the packed MFMA operand
and the tile choice of cg_swconv.
"""
from collections import namedtuple

NUM_CONVS = 5
LDS_BYTES = 160 * 1024
_LDS_B = 3 * 64 * 64 * 2  # smallest (64 x 64) 3-deep LDS-DMA weight ring slot set
_SCRATCH = 4 * 16 * 68 * 4


def round_up(v, m):
  return (v + m - 1) // m * m


def pitch(channels):
  """Channel pitch of a bf16 activation: a multiple of 32, so the 32-channel
  chunk (CK = 32: 16-byte rows, uniform K walk, three workgroups per CU)
  always divides it.  102 -> 128: the 23 % extra (zero) K on the C = 102 layers
  is cheaper than the single-chunk CK = 104 path (1 workgroup per CU, measured
  0.49 vs 0.9 PFLOP/s)."""
  return max(32, round_up(channels, 32))


def calculate_noise_shape(output_shape, noise_dim, num_convolutions, strides):
  """calciumgan.py:15-19."""
  w = output_shape[0] / (strides**num_convolutions)
  if not float(w).is_integer():
    raise ValueError('Conv1D: w {} is not an integer.'.format(w))
  return (int(w), noise_dim)


def same_padding_left(kernel_size, strides):
  """TF 'same' left pad for even input lengths: (k - s) // 2."""
  return max(kernel_size - strides, 0) // 2


def tile_rows(Lu, M, n_tiles_n):
  """Row tile of cg_swconv: 256 when the per-sample length allows it and the
  launch still has >= 256 workgroups, else 64.  Returns (CG_TILE_* value, TM)."""
  def ok(tm):
    return (Lu % tm == 0) if Lu >= tm else (tm % Lu == 0)
  if ok(256) and (M // 256) * n_tiles_n >= 256:
    return 0, 256
  if ok(64):
    return 1, 64
  if ok(256):
    return 0, 256
  raise ValueError(
      'unsupported per-sample length {} (must be a power of two <= 64 or a '
      'multiple of 64)'.format(Lu))


def lds_bytes(CK, stride, taps, Lu, TM, TN=64, mfma_rows=16,
              split_parity=False):
  """Smallest dynamic LDS of a cg_swconv launch (64-deep weight stages)."""
  c8 = CK // 8
  if mfma_rows == 16:
    pitch_a = CK + 8 * ((6 - (c8 & 3)) & 3)  # 16-byte slots == 2 (mod 4)
  else:
    pitch_a = CK + 8                         # odd slot count (4 | c8)
  S = min(Lu, TM)
  nseg = TM // S
  WR = S + taps // stride - 1
  regions = 1 if (split_parity and stride == 2) else stride
  a = max(regions * nseg * WR * pitch_a * 2, _SCRATCH)
  return round_up(a, 16) + _LDS_B * (TN // 64)


def dense_streams(cp):
  """Channel pitches whose per-timestep Dense runs on the streaming kernels
  (cg_dense_rows / cg_dense_rows_act) instead of an LDS-staged cg_swconv
  launch: W in registers up to 128, one 128-column panel of W in LDS for 256,
  384, 512."""
  return cp <= 128 or cp in (256, 384, 512)


ConvLayer = namedtuple('ConvLayer', 'cin cout lin lout cinp coutp')


def discriminator_layers(hp):
  """Conv1D(k, s, 'same') x5 with filters U..5U (calciumgan.py:145-185)."""
  u = hp.num_units
  filters = [u, 2 * u, 3 * u, 4 * u, 5 * u]
  layers = []
  cin, length = hp.num_channels, hp.signal_shape[0]
  for cout in filters:
    if length % hp.strides:
      raise ValueError('sequence length not divisible by strides**5')
    lout = length // hp.strides
    layers.append(ConvLayer(cin, cout, length, lout, pitch(cin), pitch(cout)))
    cin, length = cout, lout
  return layers


def generator_layers(hp):
  """Conv1DTranspose x5 with filters 5U..2U, C (calciumgan.py:37-87)."""
  u = hp.num_units
  filters = [5 * u, 4 * u, 3 * u, 2 * u, hp.num_channels]
  w, nd = calculate_noise_shape(hp.signal_shape, hp.noise_dim, NUM_CONVS,
                                hp.strides)
  layers = []
  cin, length = nd, w
  for cout in filters:
    lout = length * hp.strides
    layers.append(ConvLayer(cin, cout, length, lout, pitch(cin), pitch(cout)))
    cin, length = cout, lout
  return layers


# activation_fn (gan/models/utils.py:6-8): 'leakyrelu' -> LeakyReLU() (Keras
# default alpha 0.3), any other name -> layers.Activation(name).  The kernels
# implement the PIECEWISE-LINEAR ones as x -> max(x, alpha x): the fused
# epilogues, the LeakyReLU' masks of every backward chain and -- the reason for
# the restriction -- the hand-derived second backward of the gradient penalty,
# which relies on the activation's second derivative being zero.
PIECEWISE_LINEAR_ALPHA = {'leakyrelu': 0.3, 'relu': 0.0, 'linear': 1.0}


def activation_alpha(hp):
  """Slope for negative inputs of hparams.activation, or ValueError."""
  name = getattr(hp, 'activation', 'leakyrelu')
  if name not in PIECEWISE_LINEAR_ALPHA:
    raise ValueError(
        "calciumgan_amd: activation '{}' is not implemented (piecewise-linear "
        'activations only: {}; a smooth activation adds a second-derivative '
        'term to the gradient penalty that the kernel schedule does not '
        'carry)'.format(name, ', '.join(sorted(PIECEWISE_LINEAR_ALPHA))))
  return PIECEWISE_LINEAR_ALPHA[name]


def validate_hparams(hp):
  """Shapes the HIP path supports; raises ValueError like the reference does
  for a non-integer noise width (calciumgan.py:17-18)."""
  if hp.strides != 2:
    raise ValueError('calciumgan_amd: only strides=2 is implemented in HIP')
  if hp.kernel_size % 2 or hp.kernel_size > 24 or hp.kernel_size < 2:
    raise ValueError('calciumgan_amd: kernel_size must be even and <= 24')
  if hp.noise_dim % 8 or hp.noise_dim < 32:
    raise ValueError('calciumgan_amd: noise_dim must be a multiple of 8, >= 32')
  # (batch_norm: single rank only -- GeneratorNet raises under data parallelism,
  # where the batch statistics would need a cross-rank reduction)
  activation_alpha(hp)  # raises for activations the kernels do not cover
  w, _ = calculate_noise_shape(hp.signal_shape, hp.noise_dim, NUM_CONVS,
                               hp.strides)
  for lay in discriminator_layers(hp):
    lu = lay.lout
    if not ((lu % 64 == 0) or (64 % lu == 0)):
      raise ValueError(
          'calciumgan_amd: layer length {} unsupported (sequence_length/32 must '
          'be a power of two or a multiple of 64)'.format(lu))
  # tf.pad(mode='reflect') needs pad < length: PhaseShuffle on the layer-4
  # output (length L/16) bounds m (calciumgan.py:126-135)
  if hp.m >= hp.signal_shape[0] // 16:
    raise ValueError('phase shuffle m={} too large for sequence length'.format(
        hp.m))
  return w
