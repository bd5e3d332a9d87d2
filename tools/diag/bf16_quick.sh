#!/bin/bash
# quick visit: bf16 kernel tests + A/B (--wfrag 1 / 0) on long-K and short-K shapes
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | tail -3
for shape in "--batch 64 --hw 160" "--batch 8 --hw 128 --cin 256 --cout 1024 --k 1 --plain --res" "--batch 8 --hw 256 --cin 64 --cout 256 --k 1 --plain --res" "--batch 8 --hw 256 --cin 512 --cout 256 --k 1 --plain --res" "--batch 8 --hw 128 --cin 128 --cout 512 --k 1 --plain --res"; do
  for wf in 1 0; do echo -n "wfrag $wf $shape: "; timeout 120 python tools/conv_single.py --bf16 $shape --iters 30 --wfrag $wf --check-against 1 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done
done
