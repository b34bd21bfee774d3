export TMPDIR=/tmp
ARGS="conv 2 24 384 512 128 192 32 0 1 0 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -- python3 tools/bench_conv.py $ARGS > gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc2 -- python3 tools/bench_conv.py $ARGS > gpurun_out/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc1','pmc2'):
    for f in glob.glob('gpurun_out/%s/*/*counter_collection.csv'%d):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            if 'swconv' not in r['Kernel_Name']: continue
            a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
        for k,(n,v) in sorted(agg.items()):
            print(d,k,n,v/n)
PY
