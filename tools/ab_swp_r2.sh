set -u
o=gpurun_out/ab1; mkdir -p $o
R2=${R2:-$PWD/calciumgan_amd/csrc/libcalciumgan_hip_r2swp.so}  # build: git show <old>:calciumgan_amd/csrc/swconv_swp.hip, hipcc -c, link with the current objects
for rep in 1 2; do
  cp profiles/r02_tuned_tiles.json $o/tA.json; cp profiles/r02_tuned_tiles.json $o/tB.json
  CALCIUMGAN_TILE_CACHE=$o/tA.json CALCIUMGAN_HIP_LIB=$R2 python bench.py --no_cpu_baseline --steps 40 > $o/benchA_$rep.log 2>&1
  CALCIUMGAN_TILE_CACHE=$o/tB.json python bench.py --no_cpu_baseline --steps 40 > $o/benchB_$rep.log 2>&1
  for x in A B; do python - <<P
import json
l=[l for l in open('$o/bench${x}_$rep.log') if l.startswith('{')][-1]; d=json.loads(l)
print('$x$rep', round(d['ms_per_step'],3), 'swconv us', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],4), 'wgrad frac', round(d['roofline']['wgrad_kernel']['frac'],4))
P
  done
done
cp profiles/r02_tuned_tiles.json $o/tA.json; cp profiles/r02_tuned_tiles.json $o/tB.json
CALCIUMGAN_TILE_CACHE=$o/tA.json CALCIUMGAN_HIP_LIB=$R2 python tools/layer_times.py > $o/layersA.log 2>&1
CALCIUMGAN_TILE_CACHE=$o/tB.json python tools/layer_times.py > $o/layersB.log 2>&1
CALCIUMGAN_TILE_CACHE=$o/tC.json CALCIUMGAN_TUNE_LOG=$o/tuneC.jsonl python bench.py --no_cpu_baseline --steps 40 > $o/benchC.log 2>&1
tail -1 $o/benchC.log | cut -c1-120
python tools/layer_times.py > $o/layersC.log 2>&1 
