#!/bin/bash
# Winograd kernels on a GPU box: parity tests, then forward / weight-gradient timings against the direct kernels on the
# head-layer shape (tools/conv_single.py, tools/wgrad_single.py).  Usage: gpurun -- 'bash tools/wino_ab.sh [batch]'
cd "$(dirname "$0")/.."
B=${1:-64}
timeout 300 python -m pytest tests/test_gpu_wino.py -q 2>&1 | grep -E "passed|failed"
for a in "--plain" "--plain --no-wino" "" "--b8"; do echo "conv_single $a"; timeout 120 python tools/conv_single.py --batch $B --iters 10 $a 2>&1 | grep -v amdgpu; done
for a in "" "--no-wino" "--xf" "--xf --no-wino"; do echo "wgrad_single $a"; timeout 120 python tools/wgrad_single.py --batch $B --iters 5 $a 2>&1 | grep -v amdgpu; done
