#!/bin/bash
# Round-5 visit D: staggered request slots of the weights-direct bf16 instance (measurement build, hook bit 12) against the product
# schedule; per-tensor report of the FC-option gradients; fp32 training step on ONE stream under rocprofv3 (attribution); B=2 layer table.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r5d}
{
for shape in "--batch 64 --hw 160" "--batch 8 --hw 128" "--batch 8 --hw 256" "--batch 64 --hw 80 --cin 512 --cout 512" "--batch 8 --hw 256 --cin 512 --cout 256 --k 1 --plain --res"; do
  for w in 1 4097; do
    echo -n "word $w $shape: "; timeout 120 python tools/conv_single.py --bf16 $shape --iters 20 --bf16-dma $w --check-against 1 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
  done
done
} > gpurun_out/${TAG}_bf16_stag_ab.txt 2>&1
cat gpurun_out/${TAG}_bf16_stag_ab.txt
timeout 200 python tools/diag/option_grads_report.py ins_tower_fc 2>&1 | grep -v amdgpu | tail -20 | tee gpurun_out/${TAG}_fc_grads.txt
timeout 200 python tools/diag/option_grads_report.py cpr_fc 2>&1 | tail -3
( cd /tmp && CPR_TRAIN_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_train1s -- python $OLDPWD/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/prof_${TAG}_train.log 2>&1; tail -2 /tmp/prof_${TAG}_train.log | cut -c1-200 )
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
head -30 gpurun_out/${TAG}_train1s_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
timeout 300 python tools/conv_bench.py --batch 2 > gpurun_out/${TAG}_convbench_b2.txt 2>&1; tail -1 gpurun_out/${TAG}_convbench_b2.txt
