#!/bin/bash
# Round-3 visit A: MFMA-shadow microbenchmark, full GPU suite (new B=64 / 2 GiB / logit-map / refine tests), short bench line.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3a}
timeout 120 tools/diag/mfma_shadow > gpurun_out/${TAG}_mfma_shadow.txt 2>&1; tail -30 gpurun_out/${TAG}_mfma_shadow.txt
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x --durations=15 2>&1 | tail -60 > gpurun_out/${TAG}_pytest.log
tail -40 gpurun_out/${TAG}_pytest.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --batch-sweep '' > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cut -c1-1500 gpurun_out/${TAG}_bench.json
