#!/bin/bash
# Round-6 bf16 visit: the bf16 GPU tests, the ping-pong instance's check + ablation table, and the bf16 forward lines
# (configs[4] and R50 640^2 B=64) without the per-launch probe.  Usage: gpurun -- 'bash tools/gpu_round6_bf16.sh TAG'
TAG=${1:-r6b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu > $OUT/pytest_bf16.log 2>&1; echo "pytest bf16 rc=$?" >> $OUT/pytest_bf16.log
timeout 600 python tools/bf16_pp_check.py > $OUT/pp_check.txt 2>&1
timeout 600 python bench.py --config cfg4 --no-probe --no-cpu-baseline --steps 20 > $OUT/bench_cfg4_noprobe.json 2> $OUT/bench_cfg4_noprobe.err
timeout 600 python bench.py --dtype bf16 --no-probe --no-cpu-baseline --steps 20 > $OUT/bench_bf16_noprobe.json 2> $OUT/bench_bf16_noprobe.err
timeout 600 python bench.py --config cfg4 --mode train --no-probe --no-cpu-baseline > $OUT/bench_cfg4_train.json 2> $OUT/bench_cfg4_train.err
tail -3 $OUT/pytest_bf16.log
