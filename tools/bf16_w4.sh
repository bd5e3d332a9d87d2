#!/bin/bash
# Round 4: the 256 x 256 bf16 LDS-DMA tile on FOUR waves (wave = 128 x 128, hook word 161) against the eight-wave instance (33).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export CPR_BENCH_HOOKS=1
OUT=gpurun_out/${1:-r4}_bf16_w4.txt
: > $OUT
for shape in "--batch 64 --hw 160 --cin 256 --cout 256 --k 3" "--batch 8 --hw 128 --cin 256 --cout 256 --k 3" "--plain --batch 8 --hw 128 --cin 256 --cout 256 --k 3" \
             "--plain --res --batch 8 --hw 128 --cin 512 --cout 256 --k 1" "--plain --batch 16 --hw 160 --cin 256 --cout 256 --k 3" "--plain --batch 64 --hw 80 --cin 512 --cout 512 --k 3"; do
  for f in 33 161; do echo "## $shape --bf16-dma $f" >> $OUT; timeout 120 python tools/conv_single.py --bf16 --iters 20 $shape --bf16-dma $f --check-against 33 2>&1 | tail -2 >> $OUT; done
done
cat $OUT
