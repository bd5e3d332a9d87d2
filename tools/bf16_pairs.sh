#!/bin/bash
# Round 4: the 256 x 256 bf16 instance after the interleaved-cout / pair-store epilogue (compare with profiles/round4_bf16_four_wave_tile.txt, word 33)
cd "$(dirname "$0")/.."
export CPR_BENCH_HOOKS=1
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_step.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for shape in "--batch 64 --hw 160 --cin 256 --cout 256 --k 3" "--batch 8 --hw 128 --cin 256 --cout 256 --k 3" "--plain --batch 8 --hw 128 --cin 256 --cout 256 --k 3" \
             "--plain --res --batch 8 --hw 128 --cin 512 --cout 256 --k 1" "--plain --batch 16 --hw 160 --cin 256 --cout 256 --k 3" "--plain --batch 64 --hw 80 --cin 512 --cout 512 --k 3" \
             "--plain --res --batch 8 --hw 64 --cin 256 --cout 1024 --k 1"; do
  echo -n "[33] $shape: "; timeout 120 python tools/conv_single.py --bf16 --iters 20 $shape --bf16-dma 33 2>&1 | tail -1
done
