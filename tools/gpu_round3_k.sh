#!/bin/bash
# Round-3 visit K: packed fp32 transform in the Winograd forward kernel: microbenchmark (packed vs scalar adds behind the MFMA),
# Winograd parity tests, head-layer timing, bench line.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3k}
timeout 300 tools/diag/mfma_shadow > gpurun_out/${TAG}_mfma_shadow.txt 2>&1; tail -8 gpurun_out/${TAG}_mfma_shadow.txt
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | tail -5
timeout 300 python tools/wino_var.py --batch 64 --vars 4 --tpx 0 --rounds 3 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/${TAG}_wino_var.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3k_bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline'])
PY
