#!/bin/bash
# Training-step visit: bench in train mode + rocprofv3 kernel stats of the same command.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-train}
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --batch ${BATCH:-16} --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cat gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err
if [ "${PROFILE:-1}" = "1" ]; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --mode train --steps 3 --warmup 1 --batch ${BATCH:-16} --no-cpu-baseline --no-probe > /tmp/prof_${TAG}.log 2>&1 )
  find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
  tail -3 /tmp/prof_${TAG}.log
  head -25 gpurun_out/${TAG}_kernel_stats.csv
fi
