#!/bin/bash
# Round-3 visit E: Winograd weight gradient (conflict-free transform reads, slices across images) + product Winograd forward with
# LDS-DMA weights: parity tests, single-kernel timings, training bench line.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3e}
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for a in "" "--xf"; do timeout 120 python tools/wgrad_single.py --batch 64 --iters 5 $a 2>&1 | grep -v amdgpu; done
timeout 600 python bench.py --mode train --steps 5 --warmup 2 --batch 64 --no-cpu-baseline --no-probe 2>/dev/null | tail -1 > gpurun_out/${TAG}_train_bench.json
python -c "import json;d=json.load(open('gpurun_out/${TAG}_train_bench.json'));print('train', d['value'], d['ms_per_step'])"
