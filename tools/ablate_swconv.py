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
      s, '  // ---- epilogue: accumulators -> LDS -> row-contiguous 16-byte stores ----\n',
      '  {\n    float sacc = 0.f;\n    for (int mt = 0; mt < MT; ++mt)\n'
      '      for (int nt = 0; nt < NT; ++nt) sacc += acc[mt][nt][0];\n'
      '    if (sacc == 12345.f) reinterpret_cast<float*>(a.y)[0] = sacc;\n'
      '    return;\n  }\n')


def no_a_staging(s):
  return sub(s, '    if (a.nseg == 1) {\n      // fast path',
             '    if (cc > 0) {\n    } else if (a.nseg == 1) {\n      // fast path')


def no_b_dma(s):
  return sub(s, '        if (gs + 2 < total_stages) issue_dma(gs + 2);\n', '')


def no_barrier(s):
  s = no_b_dma(s)
  return sub(s, '        __builtin_amdgcn_s_barrier();\n        const uint16_t* curB',
             '        const uint16_t* curB')


VARIANTS = {
    'base': lambda s: s,
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
        os.path.join(SRC, 'wgrad.o'), os.path.join(SRC, 'pointwise.o')])
    print('built', name)


if __name__ == '__main__':
  main()
