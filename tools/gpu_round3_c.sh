#!/bin/bash
# Round-3 visit C: fused projection shortcut (dual-source conv) parity + A/B, Winograd schedule variants.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3c}
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cpr_parity.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log
tail -8 gpurun_out/${TAG}_pytest.log
timeout 600 python tools/wino_var.py --batch 64 --rounds 3 --iters 5 --vars 0,2,3,4 --tpx 0 --sched 0,1,2 > gpurun_out/${TAG}_wino_var.txt 2>&1; grep -v amdgpu.ids gpurun_out/${TAG}_wino_var.txt | tail -40
for f in 1 0; do
  CPR_FUSE_SHORTCUT=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_fuse$f.json
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_fuse$f.json'));print('fuse_shortcut=$f', d['value'], d['ms_per_step'])"
done
