#!/bin/bash
# Round-3 visit J: bf16 DMA kernel: tests again, PMC passes on the head layer.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3j}
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -3
PMC_SCRIPT=conv_single.py CONV_ARGS="--bf16 --batch 64 --iters 3" bash tools/gpu_pmc.sh ${TAG} > /dev/null 2>&1
python tools/pmc_to_json.py ${TAG} 64 conv_bf16_dma_kernel ${TAG}_pmc_bf16_dma_kernel.json 2>&1 | tail -3
cat profiles/${TAG}_pmc_bf16_dma_kernel.json 2>/dev/null | head -30
for d in 1 0; do echo "bf16 head layer, dma=$d"; timeout 120 python tools/conv_single.py --bf16 --batch 64 --iters 10 --bf16-dma $d 2>&1 | grep -v amdgpu; done
