#!/bin/bash
# Round 4: LDS-staged row-wise epilogue of the bf16 LDS-DMA kernel against the direct one (bit 10 of the hook word).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export CPR_BENCH_HOOKS=1
OUT=gpurun_out/${1:-r4}_bf16_epi2.txt
: > $OUT
run() { echo "## $*" >> $OUT; timeout 120 python tools/conv_single.py --bf16 --iters 30 "$@" 2>&1 | tail -1 >> $OUT; }
for f in 1 1025; do
  run --plain --res --batch 8 --hw 64 --cin 256 --cout 1024 --k 1 --bf16-dma $f
  run --plain --batch 8 --hw 64 --cin 256 --cout 1024 --k 1 --bf16-dma $f
  run --plain --res --batch 8 --hw 128 --cin 128 --cout 512 --k 1 --bf16-dma $f
  run --plain --res --batch 8 --hw 256 --cin 64 --cout 256 --k 1 --bf16-dma $f
  run --batch 8 --hw 128 --cin 256 --cout 256 --k 3 --bf16-dma $f
  run --batch 64 --hw 160 --cin 256 --cout 256 --k 3 --bf16-dma $f
  run --plain --batch 8 --hw 64 --cin 256 --cout 256 --k 3 --bf16-dma $f
  run --plain --batch 8 --hw 64 --cin 1024 --cout 256 --k 1 --bf16-dma $f
  run --plain --batch 8 --hw 128 --cin 128 --cout 128 --k 3 --bf16-dma $f
done
cat $OUT
