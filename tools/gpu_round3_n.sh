#!/bin/bash
# Round-3 visit N: streamed 1x1 kernel: parity tests, per-layer A/B, bench line.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3n}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -x -k "stream or dual" 2>&1 | tail -15
for s in 1 0; do timeout 300 python tools/conv_bench.py --batch 64 --stream $s 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_convbench_stream$s.txt; grep "64, 256, 1, 1, True\|128, 512, 1, 1, True\|conv total" gpurun_out/${TAG}_convbench_stream$s.txt; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --small-batch 0 --batch-sweep '' 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3n_bench.json').read())
print(d['value'], d['ms_per_step']); r=d['roofline']; print(r.get('per_instance_executed_frac')); print(r.get('per_instance_share_of_conv_time'))
PY
