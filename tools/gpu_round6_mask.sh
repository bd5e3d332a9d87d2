#!/bin/bash
# Round-6 development visit: the mixed-precision backward (mask mode, fused block-boundary gradient, transposing kernel, streaming pass,
# split-K reduce) and the advisor's seed sweep.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6m
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_options.py -q -m gpu -x -s --tb=short -p no:cacheprovider -k "seed_sweep" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl\|amdgpu.ids" | tail -30 > gpurun_out/r6m/seed_sweep.log
tail -12 gpurun_out/r6m/seed_sweep.log
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_autograd.py tests/test_gpu_p2p.py -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -12 > gpurun_out/r6m/pytest.log
tail -3 gpurun_out/r6m/pytest.log
rm -f gpurun_out/r6m/ab.txt
for v in "CPR_MIXED_MASK_MODE=1"; do
  echo "== $v" >> gpurun_out/r6m/ab.txt
  env $v timeout 600 python tools/bf16_ab.py --train --depth 50 --size 640 --batch 64 --rounds 2 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r6m/ab.txt
  env $v timeout 600 python tools/bf16_ab.py --train --rounds 2 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r6m/ab.txt
done
cat gpurun_out/r6m/ab.txt
( cd /tmp && CPR_TRAIN_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r6m -o cfg4mixed -- python $OLDPWD/bench.py --config cfg4 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/prof_r6m2.log 2>&1 )
find /tmp/prof_r6m -name "*kernel_stats*" -exec cp {} gpurun_out/r6m/ \;
grep "reduce\|transpose" gpurun_out/r6m/cfg4mixed_kernel_stats.csv | cut -c1-150
