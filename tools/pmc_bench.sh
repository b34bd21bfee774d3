# SQ counters of the MFMA kernel families over the default benchmark command
# (eager launches, tuned tiles): MFMA pipe utilisation, where the waves' cycles
# go, LDS bank conflicts.  Separate rocprofv3 --pmc passes (8 SQ slots each);
# python3 bench.py ... directly after `--`.  Run on the GPU box from the repo
# root; writes gpurun_out/pmc_sq.json (copied to profiles/ by hand).
export TMPDIR=/tmp
export CALCIUMGAN_GRAPH=0
export CALCIUMGAN_TILE_CACHE=/tmp/cg_tiles.json
python3 bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_timing > gpurun_out/pmc_tune.log 2>&1 || exit 1
export CALCIUMGAN_AUTOTUNE=0
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_sq_a -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > gpurun_out/pmc_sq_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq_b -- python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > gpurun_out/pmc_sq_b.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, json
fam_of = lambda n: ('swconv_swp' if 'swconv_swp_kernel' in n else 'swconv' if 'swconv_kernel' in n else
                    'wgrad' if ('wgrad_multi' in n or 'wgrad_kernel' in n or 'wgrad_flex_kernel' in n) else None)
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for d in ('pmc_sq_a', 'pmc_sq_b'):
    for f in glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d):
        for r in csv.DictReader(open(f)):
            fam = fam_of(r['Kernel_Name'])
            if fam is None: continue
            a = res[fam][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for f in glob.glob('gpurun_out/%s/*/*kernel_trace.csv' % d):
        for r in csv.DictReader(open(f)):
            fam = fam_of(r['Kernel_Name'])
            if fam is None: continue
            dur[fam][0] += 1; dur[fam][1] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3
out = {}
for fam, c in res.items():
    m = {k: v[1] / v[0] for k, v in c.items()}
    wc = m['SQ_WAVE_CYCLES']
    out[fam] = dict(
        launches_profiled=c['SQ_WAVE_CYCLES'][0],
        avg_launch_us_under_profiler=dur[fam][1] / dur[fam][0],
        # SQ_VALU_MFMA_BUSY_CYCLES counts per-SIMD cycles, SQ_BUSY_CU_CYCLES per CU: 4 SIMDs.
        # "busy_cu": of the cycles in which a CU held waves; "wall": of the launch's
        # wall clock on all 1024 SIMDs (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        mfma_pipe_busy_frac=m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * m['SQ_BUSY_CU_CYCLES']),
        mfma_pipe_busy_frac_wall=m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0),
        cu_busy_frac_wall=m['SQ_BUSY_CU_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8.0 * 256.0),
        wave_cycles_frac=dict(issuing=m['SQ_ACTIVE_INST_ANY'] / wc, waitcnt_or_barrier=m['SQ_WAIT_ANY'] / wc,
                              issue_stalled=m['SQ_WAIT_INST_ANY'] / wc),
        per_mfma=dict(valu=(m['SQ_INSTS_VALU'] - m['SQ_INSTS_MFMA']) / m['SQ_INSTS_MFMA'], salu=m['SQ_INSTS_SALU'] / m['SQ_INSTS_MFMA'],
                      lds=m['SQ_INSTS_LDS'] / m['SQ_INSTS_MFMA'], vmem=m['SQ_INSTS_VMEM'] / m['SQ_INSTS_MFMA']),
        lds_bank_conflict_frac=m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'],
        gfx_clock_ghz_est=m['GRBM_GUI_ACTIVE'] / 8.0 / (dur[fam][1] / dur[fam][0] * 1e3),
        raw={k: round(v, 1) for k, v in sorted(m.items())})
try:
    out['commit'] = open('profiles/.head_commit').read().strip()
except OSError:
    out['commit'] = None
out['command'] = 'rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 (CALCIUMGAN_GRAPH=0, tuned tiles)'
print(json.dumps({k: v for k, v in out.items()}, indent=1)[:3000])
json.dump(out, open('gpurun_out/pmc_sq.json', 'w'), indent=1)
PY
