cd /root/repo
export TMPDIR=/tmp
for v in "0 0" "0 8" "0 16" "0 32"; do set -- $v; echo "sched=$1 ablate=$2"; timeout 120 python tools/conv_single.py --plain --batch 64 --iters 10 --wino-sched $1 --wino-ablate $2 2>&1 | grep -v amdgpu
 ( cd /tmp && timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d /tmp/wq_$2 -o w -- python /root/repo/tools/conv_single.py --plain --batch 64 --iters 2 --wino-ablate $2 > /tmp/wq_$2.log 2>&1; tail -1 /tmp/wq_$2.log )
 python - <<PY
import csv,glob,collections
for f in glob.glob('/tmp/wq_$2/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_wino' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print({k: sum(v)/len(v) for k,v in agg.items()})
PY
done
