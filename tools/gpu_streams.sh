#!/bin/bash
# A/B: sub-batch streams and batch size; plus a PMC pass on the interleaved conv kernel.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TAG=${1:-st}
for S in 1 2 4; do
  for B in 16 32; do
    CPR_STREAMS=$S timeout 300 python bench.py --steps 5 --warmup 2 --batch $B --no-cpu-baseline > gpurun_out/${TAG}_s${S}_b${B}.json 2>/dev/null
    python -c "import json; d=json.load(open('gpurun_out/${TAG}_s${S}_b${B}.json')); print('streams $S batch $B: %.1f img/s  conv128 %.1f TF  e2e %.1f TF' % (d['value'], d['roofline']['achieved'], d['end_to_end_effective_tflops']))"
  done
done
timeout 200 python bench.py --steps 3 --warmup 1 --batch 16 > gpurun_out/${TAG}_full.json 2>/dev/null; cat gpurun_out/${TAG}_full.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['cpu_baseline'])"
CONV_ARGS="--batch 16" bash tools/gpu_pmc.sh ${TAG} 2>&1 | tail -12
