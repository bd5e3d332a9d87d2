#!/bin/bash
# Round-3 visit O: mixed-precision training step (bf16 recorded forward, fp32 backward over the widened maps).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3o}
timeout 900 python -m pytest tests/test_gpu_train_step.py -q -m gpu --tb=short -p no:cacheprovider -x -k "mixed" 2>&1 | tail -15
for m in "--mode train --batch 64 --dtype bf16" "--mode train --batch 64" "--mode train --depth 101 --size 1024 --batch 8 --dtype bf16" "--mode train --depth 101 --size 1024 --batch 8"; do
  n=$(echo $m | tr -d ' -'); timeout 600 python bench.py $m --steps 6 --warmup 2 --no-cpu-baseline --no-probe 2>gpurun_out/${TAG}_$n.err | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-260 gpurun_out/${TAG}_bench_$n.json; tail -2 gpurun_out/${TAG}_$n.err; echo
done
