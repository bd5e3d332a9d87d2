#!/bin/bash
# Round-6 development visit: the mask-mode data gradient and the restructured block backward (tests, then the training A/B).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_autograd.py -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r6m/pytest.log
tail -5 gpurun_out/r6m/pytest.log
for mm in 1 0; do
  echo "== CPR_MIXED_MASK_MODE=$mm" >> gpurun_out/r6m/train_ab.txt
  CPR_MIXED_MASK_MODE=$mm timeout 600 python tools/bf16_ab.py --train 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/r6m/train_ab.txt
  CPR_MIXED_MASK_MODE=$mm timeout 600 python tools/bf16_ab.py --train --depth 50 --size 640 --batch 64 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/r6m/train_ab.txt
done
cat gpurun_out/r6m/train_ab.txt
timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-probe --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
