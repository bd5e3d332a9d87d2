#!/bin/bash
# Round-4 visit: P2P tests + the configs[3] bench line (output convs as tap projection).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4p}
timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_autograd.py -q -m gpu --tb=short -p no:cacheprovider --maxfail=10 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -60 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --config cfg3 > gpurun_out/${TAG}_cfg3.json 2> gpurun_out/${TAG}_cfg3.err
echo "bench exit: $?"; TAG=$TAG python - <<'P'
import json, os
d=json.load(open('gpurun_out/%s_cfg3.json' % os.environ['TAG']))
print(round(d['value'],1), d['roofline'], d.get('parity'))
print(json.dumps({k: v for k, v in d.items() if k in ('infer', 'inference', 'phases', 'lsa')})[:1500])
P
