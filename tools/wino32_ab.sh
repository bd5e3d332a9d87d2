#!/bin/bash
# A/B of the two fused Winograd forward kernels (CPR_WINO_TILE=64: csrc/conv_wino.hip, 32: csrc/conv_wino32.hip) per shape.
cd "$(dirname "$0")/.."
for args in "--b8 --batch 64" "--gn-stats --batch 64" "--b8 --batch 2" "--b8 --batch 16" "--plain --batch 64 --hw 40" "--plain --batch 2 --hw 40" "--plain --batch 64 --hw 80 --cin 128 --cout 128" "--plain --batch 64 --hw 160 --cin 64 --cout 64"; do
  for t in 64 32; do
    echo -n "tile $t  $args:  "; CPR_WINO_TILE=$t timeout 120 python tools/conv_single.py $args --iters 10 2>&1 | grep -v amdgpu | tail -1
  done
done
