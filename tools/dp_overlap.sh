# Two data-parallel ranks of the real train() on ONE GPU (gloo collectives
# through the host): step time with the gradient all-reduces overlapped with
# the next segment (production schedule) vs waited for at once.  A rehearsal of
# the mechanism, not a scaling number (no xGMI, both ranks share the GPU).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 DP_TIMED_STEPS=20
for ov in 1 0; do
  mkdir -p gpurun_out/dp_ov$ov
  CALCIUMGAN_DP_OVERLAP=$ov DP_WORKER_OUT=gpurun_out/dp_ov$ov python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + ov)) tests/dp_worker.py > gpurun_out/dp_ov$ov.log 2>&1 || exit 1
done
python3 - <<'PY'
import json
out = {}
for ov in (1, 0):
    r = json.load(open('gpurun_out/dp_ov%d/rank0.json' % ov))
    out['overlap' if ov else 'wait_at_once'] = dict(ms_per_step=r['ms_per_step'], segments=r['segments'], weights_identical=r['weights_identical'])
out['note'] = '2 ranks, one MI355X, gloo (host) all-reduce of the flat gradient buffers (4.0 MB critic x5, 4.4 MB generator per step at cfg1 shapes, num_units 32); rehearsal of the overlap schedule only'
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/r02_dp2_overlap.json', 'w'), indent=1)
PY
