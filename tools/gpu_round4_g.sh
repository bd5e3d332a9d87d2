#!/bin/bash
# Round-4 visit G: full GPU suite, then the default bench line (train_step now carries torch_autograd).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4g}
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --maxfail=25 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -100 > gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; python - <<'P'
import json
d=json.load(open('gpurun_out/r4g_bench.json'))
print(round(d['value'],1), d['roofline']['frac'], d.get('small_batch',{}).get('img_per_s'))
print(json.dumps(d.get('train_step'))[:1200])
P
