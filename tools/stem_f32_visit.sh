#!/bin/bash
# Round 4: the fused fp32 stem (csrc/stem_f32.hip) -- kernel tests, parity files that see the stem, microbenchmark, headline A/B.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cpr_parity.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider --maxfail=5 2>&1 | tail -8
python tools/stem_bench.py --fp32 2>&1 | grep "^stem" | tee gpurun_out/r4_stem_f32.txt
for v in 1 0; do CPR_F32_STEM=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-probe --no-cpu-baseline --small-batch 2 --batch-sweep '' --train-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CPR_F32_STEM=$v:', round(d['value'],1), d['ms_per_step'], 'B=2:', round(d['small_batch']['img_per_s'],1))" | tee -a gpurun_out/r4_stem_f32.txt; done
