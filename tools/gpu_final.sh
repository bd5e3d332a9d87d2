#!/bin/bash
# Round-end style visit: full GPU suite, default bench, rocprofv3 kernel stats of the default bench and of the train step.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-final}
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cat gpurun_out/${TAG}_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --train-steps 0 > /tmp/prof_${TAG}.log 2>&1 )
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}t -o ${TAG}_train -- python $OLDPWD/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/prof_${TAG}t.log 2>&1 )
find /tmp/prof_${TAG}t -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-probe > gpurun_out/${TAG}_train_bench.json 2>/dev/null
cat gpurun_out/${TAG}_train_bench.json
python tools/bwd_bench.py --batch 16 --top 30 > gpurun_out/${TAG}_bwd_ops.txt 2>&1
ls gpurun_out | grep ${TAG}
