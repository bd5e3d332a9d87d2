#!/bin/bash
# Round-4 visit B: GPU suite again (autograd bridge fixed, register-resident LSA, batched P2P post-processing, TTA merge), then the
# P2PNet lines with the probe (roofline + cpu_baseline) and kernel-trace stats of the P2P forward + loss step (LSA share).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4b}
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --maxfail=25 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -120 > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
for m in "--config cfg3" "--config cfg3 --mode infer"; do
  n=$(echo $m | tr -d ' -'); timeout 600 python bench.py $m --steps 10 --warmup 3 --batch-sweep '' --small-batch 0 --train-steps 0 2>gpurun_out/${TAG}_bench_$n.err | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-300 gpurun_out/${TAG}_bench_$n.json; echo; tail -2 gpurun_out/${TAG}_bench_$n.err
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_p2p -- python $OLDPWD/bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_p2p_infer -- python $OLDPWD/bench.py --config cfg3 --mode infer --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}_i.log 2>&1 )
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
ls gpurun_out | grep ${TAG}
