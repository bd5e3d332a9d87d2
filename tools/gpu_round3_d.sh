#!/bin/bash
# Round-3 visit D: 16-byte LDS-staged epilogue of the direct conv: parity (whole GPU suite) + per-layer A/B + bench line.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3d}
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
for n in 0 1; do timeout 300 python tools/conv_bench.py --batch 64 --narrow $n 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_convbench_narrow$n.txt; tail -1 gpurun_out/${TAG}_convbench_narrow$n.txt; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json
python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench.json'));print(d['value'], d['ms_per_step'])"
