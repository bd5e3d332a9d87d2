#!/bin/bash
# Round-4 visit: bf16 tests + the configs[4] bench line (+ R50 640^2 bf16) after the 128 x 128 LDS-DMA instance.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4h}
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_step.py -q -m gpu --tb=short -p no:cacheprovider --maxfail=10 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -60 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --config cfg4 > gpurun_out/${TAG}_cfg4.json 2> gpurun_out/${TAG}_cfg4.err
echo "bench exit: $?"
timeout 600 python bench.py --dtype bf16 --no-probe > gpurun_out/${TAG}_r50_bf16.json 2> gpurun_out/${TAG}_r50_bf16.err
TAG=$TAG python - <<'P'
import json, os
for n in ('cfg4', 'r50_bf16'):
    try:
        d=json.load(open('gpurun_out/%s_%s.json' % (os.environ['TAG'], n)))
        r=d.get('roofline') or {}
        print(n, round(d['value'],1), d['ms_per_step'], r.get('kernel'), r.get('frac'))
        for k in ('per_instance_effective_tflops','per_instance_share_of_conv_time'):
            print('  ', k, r.get(k))
    except Exception as e: print(n, 'failed', e)
P
