#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for t in 0,0 128,128 128,64 64,128 64,64; do
  timeout 200 python tools/conv_bench.py --batch ${BATCH:-16} --tile $t --out gpurun_out/tiles_${t/,/x}.json > gpurun_out/tiles_${t/,/x}.txt 2>&1
  echo "tile $t: $(tail -1 gpurun_out/tiles_${t/,/x}.txt)"
done
