#!/bin/bash
# last visit of a round: the full GPU suite, smoke and the cfg4 line at the final commit (the evidence visit ran before the last fixture / naming fixes)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -12 > gpurun_out/r5final_pytest.log
tail -3 gpurun_out/r5final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --batch-sweep '' --train-steps 0 2>/dev/null | tail -1 > gpurun_out/r5final_bench_configcfg4.json; python -c "
import json; d=json.load(open('gpurun_out/r5final_bench_configcfg4.json')); r=d['roofline']; print(d['value'], r['kernel'], r['frac'], r['traffic'], d['cpu_baseline']['parity_gate']['passed'])"
