#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "" "CPR_EXPERIMENT_NO_REPACK=1" "CPR_TRAIN_STREAMS=1" "CPR_TRAIN_STREAMS=1 CPR_EXPERIMENT_NO_REPACK=1"; do
  echo -n "[$v] "; env $v timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'img/s', round(d['ms_per_step'],1),'ms')"
done
