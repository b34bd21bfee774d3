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
# the same sources with fp16 activations (-DCG_ACT_F16=1): mixed_float16 mode
LIB_F16 = os.path.join(CSRC, 'libcalciumgan_hip_f16.so')
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


def needs_build(lib=LIB):
  if not os.path.exists(lib):
    return True
  t = os.path.getmtime(lib)
  deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
  return any(os.path.getmtime(d) > t for d in deps)


def _build_one(lib, defines, suffix, verbose):
  from concurrent.futures import ThreadPoolExecutor
  jobs = []
  for src in SOURCES:
    obj = os.path.join(CSRC, src.replace('.hip', suffix + '.o'))
    jobs.append(([
        _hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
        '-Wall', '-Wno-unused-function'
    ] + defines + ['-c', os.path.join(CSRC, src), '-o', obj], obj))

  def run(job):
    if verbose:
      print(' '.join(job[0]), flush=True)
    subprocess.check_call(job[0])
    return job[1]

  # (a few translation units, the largest a minute of hipcc: compile together)
  with ThreadPoolExecutor(max_workers=4) as pool:
    objs = list(pool.map(run, jobs))
  cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib
        ] + objs
  if verbose:
    print(' '.join(cmd), flush=True)
  subprocess.check_call(cmd)


def build(force=False, verbose=True):
  """Both builds of the kernel library: bf16 activations (LIB) and fp16
  activations (LIB_F16, -DCG_ACT_F16=1)."""
  if force or needs_build(LIB):
    _build_one(LIB, [], '', verbose)
  if force or needs_build(LIB_F16):
    _build_one(LIB_F16, ['-DCG_ACT_F16=1'], '.f16', verbose)
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
