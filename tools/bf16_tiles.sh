#!/bin/bash
# Round 4: the LDS-DMA half tiles (128 x 256, 256 x 128) against the 256 x 256 tile and the register-staged kernels on the
# R101 1024^2 B = 8 layer shapes that do not fill the chip with 256 x 256 tiles.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export CPR_BENCH_HOOKS=1
OUT=gpurun_out/${1:-r4}_bf16_tiles.txt
: > $OUT
run() { echo "## $*" >> $OUT; timeout 120 python tools/conv_single.py --bf16 --plain --iters 30 "$@" 2>&1 | tail -1 >> $OUT; }
for shape in "--batch 8 --hw 64 --cin 256 --cout 256 --k 3" "--batch 8 --hw 64 --cin 1024 --cout 256 --k 1" \
             "--batch 8 --hw 64 --cin 256 --cout 1024 --k 1" "--batch 8 --hw 128 --cin 128 --cout 128 --k 3" \
             "--batch 8 --hw 128 --cin 512 --cout 128 --k 1" "--batch 8 --hw 128 --cin 128 --cout 512 --k 1" \
             "--batch 8 --hw 32 --cin 512 --cout 512 --k 3" "--batch 8 --hw 32 --cin 2048 --cout 512 --k 1" \
             "--batch 8 --hw 128 --cin 256 --cout 256 --k 3" "--batch 8 --hw 256 --cin 64 --cout 256 --k 1" \
             "--batch 8 --hw 256 --cin 256 --cout 64 --k 1" "--batch 8 --hw 256 --cin 64 --cout 64 --k 3"; do
  for f in 0 129 33 65 97; do   # register-staged | round-3 rule | 256x256 | 128x256 | 256x128
    run $shape --bf16-dma $f
  done
done
cat $OUT
