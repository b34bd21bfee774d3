#!/usr/bin/env python
"""Compile-time ablations of the ring-staged loop of wgrad.hip (development tool;
results become WRONG, timing only).  Variants land in tools/probe/_abl/lib_wg_<name>.so; time with
  CALCIUMGAN_HIP_LIB=tools/probe/_abl/lib_wg_<name>.so python tools/bench_conv.py wgrad ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'calciumgan_amd', 'csrc')
OUT = os.environ.get('ABL_OUT') or os.path.join(ROOT, 'tools', 'probe', '_abl')


def rep(s, a, b, count=1):
  assert a in s, a
  return s.replace(a, b, count)


def ring_noreads(s):  # fragment reads inside the tile loop are dropped
  return rep(s, """        read_one(NSET{}, ksn_tag, integral_constant<int, 2 * j>{});
        read_one(NSET{}, ksn_tag, integral_constant<int, 2 * j + 1>{});
""", "")


def ring_nomfma(s):
  return rep(s, """      mfma_acc(acc[g >> 1][g & 1][nt], join(f.al[SET][g], f.ah[SET][g]),
               join(f.bl[SET][nt], f.bh[SET][nt]));""",
             "      acc[g >> 1][g & 1][nt][0] += (float)f.al[SET][g][0] + (float)f.ah[SET][g][0] + (float)f.bl[SET][nt][0] + (float)f.bh[SET][nt][0];")


def ring_nodma(s):
  return rep(s, "            if (i + G::NS < n_i) issue_tile(soff);\n", "")


def ring_nobar(s):
  return rep(s, """            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();""", """            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");""")


def ring_noloop(s):  # prologue + accumulator flush (+ the reduce launch) only
  return rep(s, "    for (int i = 0; i < n_i; ++i) {\n      const int tile = bz + i * gz;", "    for (int i = 0; i < 0; ++i) {\n      const int tile = bz + i * gz;")


VARIANTS = {
    'base': lambda s: s,
    'noreads': ring_noreads,
    'nomfma': ring_nomfma,
    'nodma': ring_nodma,
    'nobar': ring_nobar,
    'nodmabar': lambda s: ring_nobar(ring_nodma(s)),
    'noloop': ring_noloop,
    'mfmaonly': lambda s: ring_nodma(ring_nobar(ring_noreads(s))),
}


def main():
  os.makedirs(OUT, exist_ok=True)
  src = open(os.path.join(SRC, 'wgrad.hip')).read()
  for name in sys.argv[1:] or list(VARIANTS):
    path = os.path.join(OUT, 'wgrad_%s.hip' % name)
    open(path, 'w').write(VARIANTS[name](src))
    obj = path[:-4] + '.o'
    subprocess.check_call([
        '/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
        '-fPIC', '-I' + SRC, '-I' + os.path.join(ROOT, 'include'), '-c', path,
        '-o', obj])
    subprocess.check_call([
        '/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC',
        '-o', os.path.join(OUT, 'lib_wg_%s.so' % name), obj] +
        [os.path.join(SRC, o) for o in ('swconv.o', 'swconv_swp.o', 'pointwise.o',
                                        'dense_rows.o')])
    print('built', name, flush=True)


if __name__ == '__main__':
  main()
