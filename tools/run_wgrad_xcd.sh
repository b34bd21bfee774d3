# wgrad: XCD-grouped vs plain block order, the five critic layers at cfg2 (GPU box)
export TMPDIR=/tmp
for geo in "2 24 384 2048 128 64" "2 24 384 1024 64 128" "2 24 384 512 128 192" "2 24 384 256 192 256" "2 24 384 128 256 320"; do
  python tools/bench_conv.py wgrad $geo 0 0 0 2>&1 | grep -v amdgpu
  python tools/bench_conv.py wgrad $geo 0 0 1 2>&1 | grep -v amdgpu
done
for plain in 0 1; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/wgx_${c}_$plain -- python3 tools/bench_conv.py wgrad 2 24 384 512 128 192 0 0 $plain > /dev/null 2>&1
done
done
python3 - <<'PY'
import csv, glob
for plain in (0,1):
    out={}
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        vals=[]
        for f in glob.glob('gpurun_out/wgx_%s_%d/*/*counter_collection.csv'%(c,plain)):
            for r in csv.DictReader(open(f)):
                if 'wgrad_kernel' in r['Kernel_Name'] and r['Counter_Name']==c:
                    vals.append(float(r['Counter_Value']))
        out[c]=sum(vals)/max(len(vals),1)
    print('layer3 plain=%d: fetch KiB raw %.0f write KiB %.0f -> HBM MB per launch %.1f' % (plain, out['FETCH_SIZE'], out['WRITE_SIZE'], (2*out['FETCH_SIZE']+out['WRITE_SIZE'])*1024/1e6))
PY
