#!/bin/bash
# Second pass of the stale-graph hunt: (1) standalone hipGraph memset-node probe,
# (2) the scenario with the library fix (cg_rownorm zeroes with a kernel),
# (3) the in-scenario memset-node probe.
set -u
out=gpurun_out/hunt2
mkdir -p $out
timeout -k 10 120 tools/probe/graph_memset > $out/graph_memset.log 2>&1
tail -8 $out/graph_memset.log | tee -a $out/summary.txt
run() {
  local n=$1 v=$2
  for i in $(seq 1 $n); do
    timeout -k 10 120 python tools/stale_graph_hunt.py $v ${v}_$i > $out/${v}_$i.log 2>&1
    grep -h "^HUNT\|^GUARD\|^MEMSETPROBE" $out/${v}_$i.log | cut -c1-300 | tee -a $out/summary.txt
  done
}
run 12 plain
run 6 noval
run 6 noeager
run 6 memsetprobe
echo done
