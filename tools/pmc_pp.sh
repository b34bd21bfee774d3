# PMC comparison of swconv tiles on one geometry (development tool; GPU box)
export TMPDIR=/tmp
export CALCIUMGAN_AUTOTUNE=0
run() {
tag=$1; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/bench_conv.py "$@" > gpurun_out/pmc_$tag.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcb_$tag -- python3 tools/bench_conv.py "$@" >> gpurun_out/pmc_$tag.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcc_$tag -- python3 tools/bench_conv.py "$@" >> gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
res={}
for d in ('pmc_$tag','pmcb_$tag','pmcc_$tag'):
    for f in glob.glob('gpurun_out/%s/*/*counter_collection.csv'%d):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            if 'swconv' not in r['Kernel_Name']: continue
            a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
        for k,(n,v) in agg.items(): res[k]=v/n
    for f in glob.glob('gpurun_out/%s/*/*kernel_trace.csv'%d):
        ds=[float(r['End_Timestamp'])-float(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'swconv' in r['Kernel_Name']]
        res['dur_us_'+d[:4]]=sum(ds)/len(ds)/1e3
print('$tag', {k:round(v,1) for k,v in sorted(res.items())})
wc=res['SQ_WAVE_CYCLES']
print('  per wave-cycle: active %.2f wait_any %.2f wait_inst %.2f | mfma_quad %.2f conflicts/lds %.3f valu/mfma %.2f lds/mfma %.2f salu/mfma %.2f vmem/mfma %.3f' % (res['SQ_ACTIVE_INST_ANY']/wc, res['SQ_WAIT_ANY']/wc, res['SQ_WAIT_INST_ANY']/wc, res['SQ_INSTS_MFMA']*4/wc, res['SQ_LDS_BANK_CONFLICT']/res['SQ_LDS_IDX_ACTIVE'], (res['SQ_INSTS_VALU']-res['SQ_INSTS_MFMA'])/res['SQ_INSTS_MFMA'], res['SQ_INSTS_LDS']/res['SQ_INSTS_MFMA'], res['SQ_INSTS_SALU']/res['SQ_INSTS_MFMA'], res['SQ_INSTS_VMEM']/res['SQ_INSTS_MFMA']))
print('  mfma busy / busy_cu_cycles: %.3f ; GUI_ACTIVE %.0f' % (res.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(res.get('SQ_BUSY_CU_CYCLES',1),1), res.get('GRBM_GUI_ACTIVE',0)))
PY
}
# D layer-3 forward: R2 t24 nB384 Lx512 Cx128 N192 (epi 1)
run old0sp conv 2 24 384 512 128 192 32 0 1 0 2 0 1
run pp10 conv 2 24 384 512 128 192 32 10 1 0 2 0 0
run pp11 conv 2 24 384 512 128 192 32 11 1 0 2 0 0
run pp9 conv 2 24 384 512 128 192 32 9 1 0 2 0 0
