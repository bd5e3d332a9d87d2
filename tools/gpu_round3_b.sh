#!/bin/bash
# Round-3 visit B: Winograd staging variants (bit-equality + interleaved timing), MFMA-shadow microbenchmark, full GPU suite.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3b}
timeout 600 python tools/wino_var.py --batch 64 --rounds 3 --iters 5 > gpurun_out/${TAG}_wino_var.txt 2>&1; tail -60 gpurun_out/${TAG}_wino_var.txt
timeout 120 tools/diag/mfma_shadow > gpurun_out/${TAG}_mfma_shadow.txt 2>&1; tail -30 gpurun_out/${TAG}_mfma_shadow.txt
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --durations=8 2>&1 | tail -40 > gpurun_out/${TAG}_pytest.log
tail -25 gpurun_out/${TAG}_pytest.log
