#!/bin/bash
# Round-3 visit I: bf16 LDS-DMA staged conv kernel: parity tests, head-layer A/B, bf16 bench lines.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3i}
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -25
for d in 1 0; do echo "bf16 head layer, dma=$d"; timeout 120 python tools/conv_single.py --bf16 --batch 64 --iters 10 --bf16-dma $d 2>&1 | grep -v amdgpu; done
for d in 1 0; do echo "bf16 1x1 256->256 160x160, dma=$d"; timeout 120 python tools/conv_single.py --bf16 --batch 64 --iters 10 --k 1 --bf16-dma $d 2>&1 | grep -v amdgpu; done
timeout 300 python bench.py --dtype bf16 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_bf16.json
python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_bf16.json'));print('bf16 R50 640 B=64', round(d['value'],1), round(d['ms_per_step'],2), d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))"
