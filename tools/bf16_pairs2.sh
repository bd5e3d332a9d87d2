#!/bin/bash
# Round 4: 256 x 256 (pair epilogue, word 33) against 128 x 128 (word 129) on the store-heavy layer shapes -- re-tuning the dispatch rule
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export CPR_BENCH_HOOKS=1
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_step.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | grep "passed\|failed\|error" | tail -3
OUT=gpurun_out/r4_bf16_pairs.txt; : > $OUT
for shape in "--res --batch 8 --hw 64 --cin 256 --cout 1024 --k 1" "--res --batch 8 --hw 128 --cin 128 --cout 512 --k 1" "--res --batch 8 --hw 32 --cin 512 --cout 2048 --k 1" \
             "--res --batch 8 --hw 256 --cin 64 --cout 256 --k 1" "--batch 8 --hw 128 --cin 512 --cout 256 --k 1" "--batch 64 --hw 40 --cin 256 --cout 1024 --k 1 --res" \
             "--batch 64 --hw 80 --cin 128 --cout 512 --k 1 --res" "--batch 64 --hw 160 --cin 64 --cout 256 --k 1 --res" "--batch 64 --hw 40 --cin 256 --cout 256 --k 3" "--batch 64 --hw 40 --cin 1024 --cout 256 --k 1"; do
  for f in 33 129; do echo -n "[$f] $shape: " >> $OUT; timeout 120 python tools/conv_single.py --bf16 --plain --iters 20 $shape --bf16-dma $f 2>&1 | tail -1 >> $OUT; done
done
cat $OUT
