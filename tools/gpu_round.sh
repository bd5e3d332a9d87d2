#!/bin/bash
# One GPU-box visit: parity tests (all, not fail-fast), a short bench, and a rocprofv3 kernel trace.
# Everything lands in gpurun_out/ (merged back by gpurun).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r1}
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/${TAG}_gpu.txt
nproc >> gpurun_out/${TAG}_gpu.txt; lscpu | grep -i "model name" >> gpurun_out/${TAG}_gpu.txt
if [ "${TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/${TAG}_pytest.log
  echo "pytest exit: $?" >> gpurun_out/${TAG}_pytest.log
  tail -5 gpurun_out/${TAG}_pytest.log
fi
timeout 600 python bench.py --steps 5 --warmup 2 --batch ${BATCH:-64} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cat gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
if [ "${CONVBENCH:-1}" = "1" ]; then
  timeout 300 python tools/conv_bench.py --batch ${BATCH:-64} --out gpurun_out/${TAG}_convbench.json > gpurun_out/${TAG}_convbench.txt 2>&1
  tail -45 gpurun_out/${TAG}_convbench.txt
  if [ "${AB:-0}" = "1" ]; then
    timeout 300 python tools/conv_bench.py --batch ${BATCH:-64} --pipeline 0 > gpurun_out/${TAG}_convbench_pipe0.txt 2>&1
    tail -3 gpurun_out/${TAG}_convbench_pipe0.txt
  fi
fi
if [ "${PMC:-0}" = "1" ]; then bash tools/gpu_pmc.sh ${TAG}; fi
if [ "${PROFILE:-1}" = "1" ]; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --steps 3 --warmup 1 --batch ${BATCH:-64} --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}.log 2>&1 )
  find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
  find /tmp/prof_${TAG} -type f | head -20
  find /tmp/prof_${TAG} -name "*stats*.csv" | head; tail -3 /tmp/prof_${TAG}.log
fi
ls -la gpurun_out | tail -20
