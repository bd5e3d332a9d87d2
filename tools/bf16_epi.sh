#!/bin/bash
# Round 4: what the epilogue of the bf16 LDS-DMA kernel costs on the store-heavy 1x1 layers (ablations: results are WRONG).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export CPR_BENCH_HOOKS=1
OUT=gpurun_out/${1:-r4}_bf16_epi.txt
: > $OUT
run() { echo "## $*" >> $OUT; timeout 120 python tools/conv_single.py --bf16 --plain --iters 30 "$@" 2>&1 | tail -1 >> $OUT; }
for shape in "--batch 8 --hw 64 --cin 256 --cout 1024 --k 1" "--batch 8 --hw 128 --cin 128 --cout 512 --k 1" \
             "--batch 8 --hw 256 --cin 64 --cout 256 --k 1" "--batch 8 --hw 128 --cin 256 --cout 256 --k 3" \
             "--batch 8 --hw 64 --cin 256 --cout 256 --k 3"; do
  for f in 33 289 545 41; do   # 256x256 | stores dropped | no epilogue | no DMA requests (epilogue only)
    run $shape --bf16-dma $f
  done
done
cat $OUT
