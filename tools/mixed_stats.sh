#!/bin/bash
# Round 4: rocprofv3 kernel stats of the mixed-precision training step (bf16 compute mode + CprTrainer), R50 640^2 B=64.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mixed -o mixed -- python $OLDPWD/bench.py --mode train --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/prof_mixed.log 2>&1 )
tail -2 /tmp/prof_mixed.log | cut -c1-300
find /tmp/prof_mixed -name "*kernel_stats*" -exec cp {} gpurun_out/r4_mixed_train_kernel_stats.csv \;
head -40 gpurun_out/r4_mixed_train_kernel_stats.csv | cut -c1-170
