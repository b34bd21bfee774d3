#!/usr/bin/env python
"""Compile-time ablations of swconv.hip (development tool): builds variants of
the library with one cost removed (results become WRONG; timing only) into
tools/probe/_abl/, to be timed with
  CALCIUMGAN_HIP_LIB=tools/probe/_abl/lib_<name>.so python tools/bench_conv.py ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'calciumgan_amd', 'csrc')
OUT = os.path.join(ROOT, 'tools', 'probe', '_abl')


def sub(s, old, new):
  assert old in s, old
  return s.replace(old, new, 1)


def no_epilogue(s):
  return sub(
      s, '  // ---- epilogue: accumulators -> LDS -> whole-line row-contiguous stores ----\n',
      '  {\n    float sacc = 0.f;\n    for (int mt = 0; mt < MT; ++mt)\n'
      '      for (int nt = 0; nt < NT; ++nt) sacc += acc[mt][nt][0];\n'
      '    if (sacc == 12345.f) reinterpret_cast<float*>(a.y)[0] = sacc;\n'
      '    return;\n  }\n')


def no_a_staging(s):
  return sub(s, '    if (a.nseg == 1) {\n      // fast path',
             '    if (cc > 0) {\n    } else if (a.nseg == 1) {\n      // fast path')


def no_b_dma(s):
  return sub(s, '        if (gs + kNBufB - 1 < total_stages) issue_dma(gs + kNBufB - 1);\n', '')


def no_barrier(s):
  s = no_b_dma(s)
  return sub(s, '        __builtin_amdgcn_s_barrier();\n        const uint16_t* curB',
             '        const uint16_t* curB')


def direct_b(s):
  """Weights straight from global memory (L1/L2) into MFMA registers, one stage
  ahead; no LDS ring, no DMA, no per-stage barrier.  Results stay correct."""
  s = sub(s, '  issue_dma(0);\n  if (total_stages > 1) issue_dma(1);\n',
          '  const uint16_t* wlane = wp + (long long)(n0 + wn * 64 + rM) * a.Kpack + g * 8;\n'
          '  bf16x8 bnext[KS][KH][NT];\n'
          '  auto load_b = [&](int gs) {\n'
          '#pragma unroll\n    for (int ks = 0; ks < KS; ++ks)\n'
          '#pragma unroll\n      for (int kh = 0; kh < KH; ++kh)\n'
          '#pragma unroll\n        for (int nt = 0; nt < NT; ++nt)\n'
          '          bnext[ks][kh][nt] = *reinterpret_cast<const bf16x8*>(\n'
          '              wlane + (long long)nt * MF * a.Kpack +\n'
          '              ((long long)gs * FS + 4 * ks + 2 * kh) * 8);\n'
          '  };\n  load_b(0);\n')
  i0 = s.index('        if (gs + 1 < total_stages)\n          asm volatile("s_waitcnt vmcnt(%0)"')
  i1 = s.index('        const uint16_t* curB = ldsB + (gs % kNBufB) * kBufB;\n')
  s = s[:i0] + (
      '        bf16x8 bcur[KS][KH][NT];\n'
      '#pragma unroll\n        for (int ks = 0; ks < KS; ++ks)\n'
      '#pragma unroll\n          for (int kh = 0; kh < KH; ++kh)\n'
      '#pragma unroll\n            for (int nt = 0; nt < NT; ++nt) bcur[ks][kh][nt] = bnext[ks][kh][nt];\n'
      '        if (gs + 1 < total_stages) load_b(gs + 1);\n') + s[i1:]
  s = sub(s, '        const uint16_t* curB = ldsB + (gs % kNBufB) * kBufB;\n', '')
  s = sub(s, '              bfrag[buf][kh][nt] = *reinterpret_cast<const bf16x8*>(\n'
             '                  curB + nt * MF * kRowB + boff[ks][kh]);\n',
          '              bfrag[buf][kh][nt] = bcur[ks][kh][nt];\n')
  return s


def stagger(n):
  """Phase-shift the co-resident workgroups of the first dispatch round (their
  epilogue store bursts / staging bursts then do not coincide)."""
  def f(s):
    return sub(
        s, '  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n'
        '  constexpr int WGM',
        '  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n'
        '  if (blockIdx.x < 768) {\n'
        '    const int slot = (blockIdx.x >> 8) %% 3;\n'
        '    for (int i = 0; i < slot * %d; ++i) __builtin_amdgcn_s_sleep(32);\n'
        '  }\n'
        '  constexpr int WGM' % n)
  return f


VARIANTS = {
    'base': lambda s: s,
    'ring4': lambda s: sub(s, 'constexpr int kNBufB = 3;', 'constexpr int kNBufB = 4;'),
    'stag3': stagger(3),
    'stag6': stagger(6),
    'stag10': stagger(10),
    'directb': direct_b,
    'noepi': no_epilogue,
    'noa': no_a_staging,
    'nob': no_b_dma,
    'nobar': no_barrier,
    'noab': lambda s: no_a_staging(no_b_dma(s)),
    'loop': lambda s: no_epilogue(no_a_staging(no_barrier(s))),
}


def main():
  os.makedirs(OUT, exist_ok=True)
  src = open(os.path.join(SRC, 'swconv.hip')).read()
  names = sys.argv[1:] or list(VARIANTS)
  for name in names:
    path = os.path.join(OUT, 'swconv_%s.hip' % name)
    open(path, 'w').write(VARIANTS[name](src))
    obj = path[:-4] + '.o'
    subprocess.check_call([
        '/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
        '-fPIC', '-I' + SRC, '-I' + os.path.join(ROOT, 'include'), '-c', path,
        '-o', obj])
    subprocess.check_call([
        '/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC',
        '-o', os.path.join(OUT, 'lib_%s.so' % name), obj,
        os.path.join(SRC, 'wgrad.o'), os.path.join(SRC, 'pointwise.o'),
        os.path.join(SRC, 'dense_rows.o')])
    print('built', name)


if __name__ == '__main__':
  main()
