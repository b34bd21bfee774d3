"""2-rank data-parallel run of the REAL train() path on the GPU box: both
ranks drive the one available GPU, collectives go over gloo (the production
backend 'nccl' = RCCL needs one GPU per rank; the code path -- asynchronous
all-reduce of the flat gradient buffer between hipGraph segments, 1/N folded
into Adam -- is the same).  Checks what data parallelism must guarantee: the
replicas hold bit-identical weights after every rank applied the reduced
gradients, the step replays as 2*n_critic+3 graphs cut around the collectives,
and the per-rank noise streams differ."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _run_workers(tmp_path, backend):
  env = dict(os.environ)
  env['DP_WORKER_OUT'] = str(tmp_path)
  env['DP_BACKEND'] = backend
  env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()),
         os.path.join(ROOT, 'tests', 'dp_worker.py')]
  out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  recs = [json.load(open(os.path.join(str(tmp_path), 'rank%d.json' % r)))
          for r in (0, 1)]
  assert sorted(r['rank'] for r in recs) == [0, 1]
  for r in recs:
    assert r['world'] == 2 and r['backend'] == backend
    assert r['weights_identical'], r
    assert r['finite'], r
    assert r['graphed'] and r['segments'] == 2 * 5 + 3, r
    assert r['noise_differs'], r
    # losses / metrics are all-reduced once per step: every rank logs the
    # global mean; rank 0's tuned tiles are broadcast: identical launches
    assert r['logged_identical'], r
    assert r['tiles_identical'], r


def test_two_rank_train_keeps_replicas_identical(tmp_path):
  _run_workers(tmp_path, 'gloo')


def test_two_rank_train_over_rccl(tmp_path):
  """The same two ranks over backend 'nccl' (= RCCL), one GPU each: runs
  wherever the box has at least two devices (the driver's 8-GPU node), skipped
  on the single-GPU test box."""
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip('needs one GPU per rank (found {})'.format(
        torch.cuda.device_count()))
  _run_workers(tmp_path, 'nccl')
