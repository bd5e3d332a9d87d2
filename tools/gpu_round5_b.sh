#!/bin/bash
# Round-5 visit B: the 256 x 256 bf16 tile with the weights direct to registers (conv_bf16_dma_kernel<4, 2, 4, true>): parity +
# bit-equality tests, A/B per layer shape on the measurement build (--wfrag 1 / 0), configs[4] and R50 bf16 lines both ways.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r5b}
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_step.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -40 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
{
for shape in "--batch 8 --hw 128" "--batch 8 --hw 256" "--batch 64 --hw 160" "--batch 16 --hw 160" "--batch 64 --hw 80 --cin 512 --cout 512" "--batch 8 --hw 64 --cin 256 --cout 1024 --k 1 --plain --res" "--batch 8 --hw 256 --cin 512 --cout 256 --k 1 --plain --res"; do
  for wf in 1 0; do
    echo -n "wfrag $wf  $shape:  "; timeout 120 python tools/conv_single.py --bf16 $shape --iters 20 --wfrag $wf --check-against 1 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
  done
done
} > gpurun_out/${TAG}_bf16_wfrag_ab.txt 2>&1
cat gpurun_out/${TAG}_bf16_wfrag_ab.txt
for wf in 1 0; do
  CPR_BF16_WFRAG=$wf timeout 300 python bench.py --config cfg4 --no-probe --steps 20 --warmup 5 --no-cpu-baseline --batch-sweep '' --train-steps 0 --small-batch 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_cfg4_wfrag$wf.json; echo "cfg4 wfrag $wf: $(cut -c1-160 gpurun_out/${TAG}_cfg4_wfrag$wf.json)"
  CPR_BF16_WFRAG=$wf timeout 300 python bench.py --dtype bf16 --no-probe --steps 10 --warmup 3 --no-cpu-baseline --batch-sweep '' --train-steps 0 --small-batch 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_r50bf16_wfrag$wf.json; echo "r50 bf16 wfrag $wf: $(cut -c1-160 gpurun_out/${TAG}_r50bf16_wfrag$wf.json)"
done
