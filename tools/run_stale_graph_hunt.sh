#!/bin/bash
# One gpurun call: the stale-graph scenario in several variants, a process each
# (tools/stale_graph_hunt.py).  Logs under gpurun_out/hunt/.
set -u
out=gpurun_out/hunt
mkdir -p $out
run() {  # run <n> <variant> [ENV=VAL ...]
  local n=$1 v=$2; shift 2
  for i in $(seq 1 $n); do
    local tag="${v}_$(echo "$*" | tr ' =' '__')_$i"
    env "$@" CALCIUMGAN_TILE_CACHE=$out/tiles_$tag.json \
      timeout -k 10 120 python tools/stale_graph_hunt.py $v $tag \
      > $out/$tag.log 2>&1
    grep -h "^HUNT\|^GUARD" $out/$tag.log | cut -c1-400 | tee -a $out/summary.txt
  done
}
run ${N_PLAIN:-10} plain X=1
run 5 guard X=1
run 5 noval X=1
run 5 noeager X=1
run 5 val8 X=1
run 5 poison X=1
run 5 plain CALCIUMGAN_SPLIT_K=0
run 5 plain CALCIUMGAN_SWP_TILES=0
# every BAD plain table again WITHOUT tuning launches (table loaded from file)
for f in $(grep -l "BAD" $out/plain_X_1_*.log 2>/dev/null); do
  t=$(basename $f .log)
  for i in 1 2 3; do
    cp $out/tiles_$t.json $out/tiles_re_${t}_$i.json
    CALCIUMGAN_TILE_CACHE=$out/tiles_re_${t}_$i.json timeout -k 10 120 \
      python tools/stale_graph_hunt.py plain re_${t}_$i > $out/re_${t}_$i.log 2>&1
    grep -h "^HUNT" $out/re_${t}_$i.log | cut -c1-400 | tee -a $out/summary.txt
  done
done
echo done
