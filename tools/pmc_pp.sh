# PMC comparison of swconv tiles on one geometry (development tool; GPU box)
export TMPDIR=/tmp
export CALCIUMGAN_AUTOTUNE=0
run() {
tag=$1; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/bench_conv.py "$@" > gpurun_out/pmc_$tag.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcb_$tag -- python3 tools/bench_conv.py "$@" >> gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
res={}
for d in ('pmc_$tag','pmcb_$tag'):
    for f in glob.glob('gpurun_out/%s/*/*counter_collection.csv'%d):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            if 'swconv' not in r['Kernel_Name']: continue
            a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
        for k,(n,v) in agg.items(): res[k]=v/n
    for f in glob.glob('gpurun_out/%s/*/*kernel_trace.csv'%d):
        ds=[float(r['End_Timestamp'])-float(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'swconv' in r['Kernel_Name']]
        res['dur_us']=sum(ds)/len(ds)/1e3
wc=res['SQ_WAVE_CYCLES']
print('$tag: %.1f us | mfma pipe busy %.2f | wave cycles: issuing %.2f waiting %.2f issue-stalled %.2f | per mfma: valu %.2f salu %.2f lds %.2f vmem %.3f | lds conflict %.3f lds active/cu-cycle %.2f | clock %.2f GHz' % (res['dur_us'], res['SQ_VALU_MFMA_BUSY_CYCLES']/(4*res['SQ_BUSY_CU_CYCLES']), res['SQ_ACTIVE_INST_ANY']/wc, res['SQ_WAIT_ANY']/wc, res['SQ_WAIT_INST_ANY']/wc, (res['SQ_INSTS_VALU']-res['SQ_INSTS_MFMA'])/res['SQ_INSTS_MFMA'], res['SQ_INSTS_SALU']/res['SQ_INSTS_MFMA'], res['SQ_INSTS_LDS']/res['SQ_INSTS_MFMA'], res['SQ_INSTS_VMEM']/res['SQ_INSTS_MFMA'], res['SQ_LDS_BANK_CONFLICT']/res['SQ_LDS_IDX_ACTIVE'], res['SQ_LDS_IDX_ACTIVE']/res['SQ_BUSY_CU_CYCLES'], res['GRBM_GUI_ACTIVE']/8/(res['dur_us']*1e3)))
PY
}
# D layer-3 forward: R2 t24 nB384 Lx512 Cx128 N192 (epi 1); D layer-1 forward; dgrad D4
run d3_old0sp conv 2 24 384 512 128 192 32 0 1 0 2 0 1
run d3_swp10 conv 2 24 384 512 128 192 32 10 1 0 2 0 0
run d3_swp11 conv 2 24 384 512 128 192 32 11 1 0 2 0 0
run d1_swp10 conv 2 24 384 2048 128 64 32 10 1 0 2 0 0
run dg4_swp10 conv 1 12 384 128 256 192 32 10 2 0 2 0 0
run dg4_old conv 1 12 384 128 256 192 32 0 2 0 2 0 0
