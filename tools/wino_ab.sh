cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wino.py -q -x 2>&1 | tail -3
for a in "--plain" "" "--b8"; do echo "args: $a"; timeout 120 python tools/conv_single.py --batch 64 --iters 10 $a 2>&1 | grep -v amdgpu; done
( cd /tmp && timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/wq -o w -- python /root/repo/tools/conv_single.py --b8 --batch 64 --iters 2 > /tmp/wq.log 2>&1; tail -1 /tmp/wq.log )
python - <<PY
import csv,glob,collections
for f in glob.glob('/tmp/wq/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_wino' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print({k: sum(v)/len(v) for k,v in agg.items()})
PY
