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
# (ABL_OUT=tools/_ab/abl: a directory that travels to the GPU box)
OUT = os.environ.get('ABL_OUT') or os.path.join(ROOT, 'tools', 'probe', '_abl')


def no_loop(s):  # prologue + epilogue only
  a = """  for (int p = 0; p < full_passes; ++p) run_pass(False{}, p);
  if (a.narrow) run_pass(True{}, full_passes);
"""
  assert a in s
  return s.replace(a, '')


def no_dma(s):  # no DMA issue inside the passes (stale LDS is read)
  a = "      if (!ends || s + 3 < NSTG) issue_b(s % kSwpRing);\n"
  b = "        if (!last) {\n          // apw (1 or 2) slots of this wave per issuing stage"
  assert a in s and b in s
  s = s.replace(a, '')
  return s.replace(b, "        if (false) {\n          // apw (1 or 2) slots of this wave per issuing stage")


def no_dma_b(s):  # weight stages only
  a = "      if (!ends || s + 3 < NSTG) issue_b(s % kSwpRing);\n"
  assert a in s
  return s.replace(a, '')


def no_dma_a(s):  # window pieces only
  b = "        if (!last) {\n          // apw (1 or 2) slots of this wave per issuing stage"
  assert b in s
  return s.replace(b, "        if (false) {\n          // apw (1 or 2) slots of this wave per issuing stage")


def no_reads(s):  # no fragment reads at all (uninitialised operands)
  n = len(re.findall(r'lds_read128<[^;]*;', s))
  assert n == 2, n
  return re.sub(r'lds_read128<[^;]*;', ';', s)


def no_barrier(s):
  a = "      __builtin_amdgcn_s_barrier();\n      CG_TR(2);  // barrier\n"
  assert a in s
  return s.replace(a, "")


def no_epilogue(s):  # (anchor of the round-2 kernel: a no-op on the persistent form)
  return s.replace(
      '  // ---- epilogue: accumulators -> LDS -> whole-line row-contiguous stores ------\n',
      '  {\n    float sacc = 0.f;\n    for (int mt = 0; mt < MT; ++mt)\n'
      '      for (int nt = 0; nt < NT; ++nt) sacc += acc[mt][nt][0];\n'
      '    if (sacc == 12345.f) reinterpret_cast<float*>(a.y)[0] = sacc;\n'
      '    return;\n  }\n', 1)


def stagger(kind, n):
  def f(s):
    a = "  const int lin = blockIdx.x;\n"
    assert a in s
    cond = {'slot': "((lin >> 3) >> 5) & 1", 'adj': "(lin >> 3) & 1"}[kind]
    return s.replace(a, a + "  if (lin < 512 && (%s)) {\n    for (int i = 0; i < %d; ++i) __builtin_amdgcn_s_sleep(127);\n  }\n" % (cond, n))
  return f


def setprio(s):  # MFMA blocks at priority 1 (the reads / DMA issue of the other waves yield)
  for a in ("      mfma_step(af0, bf0);\n", "      mfma_step(af1, bf1);\n"):
    assert a in s
    s = s.replace(a, "      __builtin_amdgcn_s_setprio(1);\n" + a + "      __builtin_amdgcn_s_setprio(0);\n")
  return s


VARIANTS = {
    'base': lambda s: s,
    'noepi': no_epilogue,
    'noloop': no_loop,
    'nodma': no_dma,
    'nodma_a': no_dma_a,
    'nodma_b': no_dma_b,
    'noreads': no_reads,
    'nobar': no_barrier,
    'mfmaonly': lambda s: no_barrier(no_reads(no_dma(s))),
    'setprio': setprio,
    'stag_slot2': stagger('slot', 2),
    'stag_slot4': stagger('slot', 4),
    'stag_adj2': stagger('adj', 2),
    'stag_adj4': stagger('adj', 4),
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
