#!/bin/bash
# Round 4: the bf16-mode stem kernels (csrc/stem_bf16.hip) -- tests, microbenchmark, configs[4] with and without them.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | tail -6
python tools/stem_bench.py 2>&1 | grep "^stem" | tee gpurun_out/r4_stem_bench.txt
for v in "CPR_BF16_STEM=1" "CPR_BF16_STEM_POOL=0" "CPR_BF16_STEM=0"; do env $v timeout 300 python bench.py --config cfg4 --no-probe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v:', round(d['value'],1), d['ms_per_step'])" | tee -a gpurun_out/r4_stem_bench.txt; done
