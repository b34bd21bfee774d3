"""Build the gfx950 kernel library (C ABI: include/calciumgan_hip.h) in-tree.

``python -m calciumgan_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles for gfx950 without a GPU present.  The .so is git-ignored but
travels to the GPU box with the working tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libcalciumgan_hip.so')
SOURCES = ['swconv.hip', 'swconv_swp.hip', 'wgrad.hip', 'pointwise.hip',
           'dense_rows.hip']
HEADERS = ['cg_common.h', 'swconv_args.h', os.path.join('..', '..', 'include',
                                       'calciumgan_hip.h')]


def _hipcc():
  for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
    if cand and (os.path.isabs(cand) and os.path.exists(cand) or
                 not os.path.isabs(cand)):
      return cand
  raise RuntimeError('hipcc not found')


def needs_build():
  if not os.path.exists(LIB):
    return True
  t = os.path.getmtime(LIB)
  deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
  if not force and not needs_build():
    return LIB
  objs = []
  for src in SOURCES:
    obj = os.path.join(CSRC, src.replace('.hip', '.o'))
    cmd = [
        _hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
        '-Wall', '-Wno-unused-function', '-c',
        os.path.join(CSRC, src), '-o', obj
    ]
    if verbose:
      print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    objs.append(obj)
  cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB
        ] + objs
  if verbose:
    print(' '.join(cmd), flush=True)
  subprocess.check_call(cmd)
  return LIB


HOST_LIB = os.path.join(CSRC, 'libcalciumgan_host.so')
HOST_SOURCES = ['oasis_ar1.c', 'crc32c.c']


def build_host(force=False, verbose=True):
  """gcc build of the host-side C helpers (OASIS AR(1) deconvolution for the
  post-hoc spike statistics, CRC-32C for TFRecord framing; not part of the GPU
  hot path)."""
  srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
  if (not force and os.path.exists(HOST_LIB) and
      all(os.path.getmtime(s) <= os.path.getmtime(HOST_LIB) for s in srcs)):
    return HOST_LIB
  cmd = ['gcc', '-O2', '-shared', '-fPIC', '-o', HOST_LIB] + srcs + ['-lm']
  if verbose:
    print(' '.join(cmd), flush=True)
  subprocess.check_call(cmd)
  return HOST_LIB


if __name__ == '__main__':
  build(force='--force' in sys.argv)
  build_host(force='--force' in sys.argv)
