#!/bin/bash
# Round-3 visit F: whole GPU suite (new option cases, transposed Hungarian, product Winograd with LDS-DMA weights).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3f}
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${TAG}_pytest.log
grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" gpurun_out/${TAG}_pytest.log | tail -45
