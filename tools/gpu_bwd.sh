#!/bin/bash
# quick GPU check of the backward kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${TAG:-bwd}
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_backward.py} -m gpu -x -q 2>&1 | tail -40 > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
