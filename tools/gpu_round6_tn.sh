#!/bin/bash
# Round-6 development visit: mixed-precision backward -- tests of the touched files, then a same-box training A/B of the switch in $AB.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6t
export TMPDIR=/tmp
AB=${AB:-CPR_REFRESH_IN_PLACE}
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_autograd.py tests/test_gpu_fullsize_grads.py tests/test_gpu_p2p.py tests/test_gpu_bf16.py -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -25 > gpurun_out/r6t/pytest.log
tail -8 gpurun_out/r6t/pytest.log
rm -f gpurun_out/r6t/ab.txt
for v in "$AB=0" "$AB=1" "$AB=0" "$AB=1"; do
  echo "== $v" >> gpurun_out/r6t/ab.txt
  env $v timeout 600 python tools/bf16_ab.py --train --depth 50 --size 640 --batch 64 --rounds 2 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r6t/ab.txt
  env $v timeout 600 python tools/bf16_ab.py --train --rounds 2 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r6t/ab.txt
done
cat gpurun_out/r6t/ab.txt
timeout 600 python bench.py --config cfg4 --mode train --steps 6 --warmup 2 --no-probe 2>/dev/null | tail -1 > gpurun_out/r6t/bench_train_cfg4.json
timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-probe 2>/dev/null | tail -1 > gpurun_out/r6t/bench_train_cfg1.json
python - <<P
import json
for f in ('bench_train_cfg4', 'bench_train_cfg1'):
    d=json.load(open('gpurun_out/r6t/%s.json' % f))
    g=d['train_step']['parity_gate']
    print(f, d['value'], g['passed'], {k:g.get(k) for k in ('per_block_max','backward_kernels','cosine','max_rel_l2')})
P
