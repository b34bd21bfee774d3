# HBM traffic of the MFMA kernels from rocprofv3 PMC counters (separate passes
# for FETCH_SIZE and WRITE_SIZE: they do not fit one pass on gfx950).
export TMPDIR=/tmp
export CALCIUMGAN_GRAPH=0
# tune once (no profiler), then profile exactly the tuned launches
# BENCH_ARGS: extra bench.py arguments (another workload); PREFIX: output name prefix
export CALCIUMGAN_TILE_CACHE=/tmp/cg_tiles${PREFIX:-}.json
export PREFIX=${PREFIX:-}
python3 bench.py $BENCH_ARGS --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_timing > gpurun_out/${PREFIX}traffic_tune.log 2>&1
export CALCIUMGAN_AUTOTUNE=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${PREFIX}traffic_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${PREFIX}traffic_$c -- python3 bench.py $BENCH_ARGS --steps ${PMC_STEPS:-2} --warmup 1 --no_cpu_baseline --no_kernel_timing > gpurun_out/${PREFIX}traffic_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json, os
PREFIX = os.environ.get('PREFIX', '')
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    # (the newest run only: gpurun_out/ may hold earlier passes)
    fs = sorted(glob.glob('gpurun_out/%straffic_%s/*/*counter_collection.csv' % (PREFIX, c)), key=os.path.getmtime)
    for r in csv.DictReader(open(fs[-1])):
        n = r['Kernel_Name']
        fam = ('swconv' if ('swconv_kernel' in n or 'swconv_swp_kernel' in n) else
               'wgrad_reduce' if ('wgrad_reduce' in n or 'wgrad_flex_reduce' in n) else
               'wgrad_batched' if ('wgrad_multi' in n or 'wgrad_flex_kernel' in n) else
               'wgrad_single' if ('wgrad_kernel' in n and 'dense1' not in n) else
               # the HBM-bound kernels, one family per kernel
               (__import__('re').search(r'(\w+_kernel)', n).group(1) if '_kernel' in n and 'at::' not in n and 'rocclr' not in n else None))
        if fam is None or r['Counter_Name'] != c:
            continue
        a = res[fam][c]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
out = {}
for fam, d in res.items():
    n = d['FETCH_SIZE'][0]
    fetch_kb = d['FETCH_SIZE'][1] / n
    write_kb = d['WRITE_SIZE'][1] / d['WRITE_SIZE'][0]
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE reports exactly half the bytes of a
    # wide (16 B/lane) coalesced streaming read on gfx950 -> doubled; WRITE_SIZE
    # is exact for 16-B-per-lane stores and float atomics.  Units: KiB.
    out[fam] = dict(launches=n, fetch_kib_raw=fetch_kb, write_kib=write_kb,
                    hbm_bytes_per_launch=(2 * fetch_kb + write_kb) * 1024)
# cg_wgrad_batched: the conv layers of one backward pass in one launch (5 critic
# passes at 3 x 128 samples + 1 generator pass at 128 per step); algorithmic
# bytes = x + g of every layer once + the partial sums written
if not PREFIX:
    out['wgrad_batched']['note'] = ('average over the 5 critic-pass launches (nB 384) and the generator-pass '
                                    'launch (nB 128) of a step; x + g once + partial sums: 793 / 264 MB')
out['bench_args'] = os.environ.get('BENCH_ARGS', '')
try:
    out['commit'] = open('profiles/.head_commit').read().strip()
except OSError:
    out['commit'] = None
print(json.dumps(out, indent=1)[:4000])
json.dump(out, open('gpurun_out/%spmc_traffic.json' % PREFIX, 'w'), indent=1)
PY
