#!/bin/bash
# Round 4: PMC passes (three separate --pmc runs) on the two bf16 LDS-DMA instances at configs[4] layer shapes.
cd "$(dirname "$0")/.."
# <2,1>: layer3 conv3, 1x1 256 -> 1024 + folded BN + residual + ReLU on 8 x 64 x 64: in 16.8 MB + residual 67.1 + out 67.1 + weights 0.5
export ALGO_BYTES=$((32768*256*2 + 2*32768*1024*2 + 256*1024*2)) SHAPE_DESC="bf16 1x1 256->1024 + BN + residual + ReLU on (8,64,64,256): R101 layer3 conv3 at 1024^2"
PMC_SCRIPT=conv_single.py CONV_ARGS="--bf16 --plain --res --batch 8 --hw 64 --cin 256 --cout 1024 --k 1 --iters 3" bash tools/gpu_pmc.sh r4s > /dev/null 2>&1
python tools/pmc_to_json.py r4s 8 "conv_bf16_dma_kernel<2, 1" round4_pmc_bf16_small_tile.json | tail -22
# <4,2>: head / FPN 3x3 256 -> 256 + GroupNorm statistics on 8 x 128 x 128
export ALGO_BYTES=$((2*131072*256*2 + 256*2304*2)) SHAPE_DESC="bf16 3x3 256->256 + GroupNorm statistics on (8,128,128,256): head layer at 1024^2, stride 8"
PMC_SCRIPT=conv_single.py CONV_ARGS="--bf16 --batch 8 --hw 128 --cin 256 --cout 256 --k 3 --iters 3" bash tools/gpu_pmc.sh r4b > /dev/null 2>&1
python tools/pmc_to_json.py r4b 8 "conv_bf16_dma_kernel<4, 2" round4_pmc_bf16_big_tile.json | tail -22
