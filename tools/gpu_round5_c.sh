#!/bin/bash
# Round-5 visit C: backward of the CPRHead options (num_refine > 1, instance tower, FC layers) + the bridge / trainer files they touch;
# loop ablations of the weights-direct-to-registers bf16 instance on the head layer; its PMC passes (three separate --pmc runs).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r5c}
timeout 900 python -m pytest tests/test_gpu_options.py tests/test_gpu_autograd.py tests/test_gpu_train_step.py tests/test_gpu_backward.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -120 > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
{
echo "conv_bf16_dma_kernel<4, 2, 4, true> loop ablations, head layer 3x3 256->256 + GN statistics, tools/conv_single.py --bf16 --bf16-dma WORD (results of ablated runs are WRONG by design)"
for spec in "1:product" "7:no waits, no barrier" "9:no LDS-DMA requests (activations)" "17:no fragment reads" "2049:no weight-fragment loads" "2057:no DMA requests, no weight loads" "2073:no vector-memory requests, no fragment reads" "2079:MFMAs + prologue / epilogue only"; do
  w=${spec%%:*}; what=${spec#*:}
  for shape in "--batch 64 --hw 160" "--batch 8 --hw 128"; do
    echo -n "word $w ($what) $shape: "; timeout 120 python tools/conv_single.py --bf16 $shape --iters 20 --bf16-dma $w 2>&1 | grep -v amdgpu | tail -1
  done
done
echo "the same ablations on the LDS-staged instance (--wfrag 0)"
for spec in "1:product" "9:no LDS-DMA requests" "17:no fragment reads" "31:MFMAs + prologue / epilogue only"; do
  w=${spec%%:*}; what=${spec#*:}
  echo -n "word $w ($what) --batch 64 --hw 160 --wfrag 0: "; timeout 120 python tools/conv_single.py --bf16 --batch 64 --hw 160 --iters 20 --bf16-dma $w --wfrag 0 2>&1 | grep -v amdgpu | tail -1
done
} > gpurun_out/${TAG}_bf16_bd_ablation.txt 2>&1
cat gpurun_out/${TAG}_bf16_bd_ablation.txt
export ALGO_BYTES=$((2*131072*256*2 + 256*2304*2)) SHAPE_DESC="bf16 3x3 256->256 + GroupNorm statistics on (8,128,128,256): head layer at 1024^2, stride 8"
PMC_SCRIPT=conv_single.py CONV_ARGS="--bf16 --batch 8 --hw 128 --cin 256 --cout 256 --k 3 --iters 3" bash tools/gpu_pmc.sh ${TAG}b > /dev/null 2>&1
python tools/pmc_to_json.py ${TAG}b 8 "conv_bf16_dma_kernel<4, 2" round5_pmc_bf16_big_tile.json | tail -22
cp profiles/round5_pmc_bf16_big_tile.json gpurun_out/ 2>/dev/null
