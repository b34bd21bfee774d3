#!/usr/bin/env python
"""Compile-time ablations of swconv_swp.hip (development tool): builds variants
of the library with one cost removed (results become WRONG; timing only) into
tools/probe/_abl/, to be timed with
  CALCIUMGAN_HIP_LIB=tools/probe/_abl/lib_swp_<name>.so python tools/bench_conv.py ...
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'calciumgan_amd', 'csrc')
OUT = os.path.join(ROOT, 'tools', 'probe', '_abl')


def loop_region(s):
  i0 = s.index('  int gs = 0;\n  int na_prev = 0;')
  i1 = s.index('  // ---- epilogue: accumulators')
  return i0, i1


def in_loop(s, fn):
  i0, i1 = loop_region(s)
  return s[:i0] + fn(s[i0:i1]) + s[i1:]


def no_vmwait(s):
  return in_loop(s, lambda t: re.sub(r'asm volatile\("s_waitcnt vmcnt[^;]*;', ';', t))


def no_barrier(s):
  return in_loop(s, lambda t: t.replace('__builtin_amdgcn_s_barrier();', ''))


def no_dma(s):
  def f(t):
    t = t.replace('if (gs + 3 < pa.total_stages) issue_b(gs + 3);', '')
    t = t.replace('na_prev = issue_a_range(p + 1, s * pa.apw, (s + 1) * pa.apw);',
                  'na_prev = 0;')
    return t
  return in_loop(no_vmwait(s), f)


def no_reads(s):
  def f(t):
    t = t.replace('read_frags(af1, bf1, p, s, gs, 1);', '')
    return re.sub(r'read_frags\(af0, bf0, last_of_pass[^;]*;', '', t, flags=re.S)
  return in_loop(s, f)


def no_mfma(s):
  def f(t):
    t = t.replace('mfma_step(af0, bf0);', 'acc[0][0][0] += af0[0][0] + bf0[0][0];')
    t = t.replace('mfma_step(af1, bf1);', 'acc[0][0][1] += af1[0][0] + bf1[0][0];')
    return t
  return in_loop(s, f)


def no_epilogue(s):
  return s.replace(
      '  // ---- epilogue: accumulators -> LDS -> whole-line row-contiguous stores ------\n',
      '  {\n    float sacc = 0.f;\n    for (int mt = 0; mt < MT; ++mt)\n'
      '      for (int nt = 0; nt < NT; ++nt) sacc += acc[mt][nt][0];\n'
      '    if (sacc == 12345.f) reinterpret_cast<float*>(a.y)[0] = sacc;\n'
      '    return;\n  }\n', 1)


VARIANTS = {
    'base': lambda s: s,
    'novm': no_vmwait,
    'nobar': lambda s: no_barrier(no_vmwait(s)),
    'nodma': no_dma,
    'noreads': no_reads,
    'nomfma': no_mfma,
    'noepi': no_epilogue,
    'loop': lambda s: no_epilogue(no_barrier(no_dma(s))),
}


def main():
  os.makedirs(OUT, exist_ok=True)
  src = open(os.path.join(SRC, 'swconv_swp.hip')).read()
  names = sys.argv[1:] or list(VARIANTS)
  for name in names:
    path = os.path.join(OUT, 'swconv_swp_%s.hip' % name)
    open(path, 'w').write(VARIANTS[name](src))
    obj = path[:-4] + '.o'
    subprocess.check_call([
        '/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
        '-fPIC', '-I' + SRC, '-I' + os.path.join(ROOT, 'include'), '-c', path,
        '-o', obj])
    subprocess.check_call([
        '/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC',
        '-o', os.path.join(OUT, 'lib_swp_%s.so' % name), obj] +
        [os.path.join(SRC, o) for o in ('swconv.o', 'wgrad.o', 'pointwise.o',
                                        'dense_rows.o')])
    print('built', name, flush=True)


if __name__ == '__main__':
  main()
