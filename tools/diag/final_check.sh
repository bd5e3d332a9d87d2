#!/bin/bash
# last visit of a round: the full GPU suite, smoke, the default line and the lines the last product changes touch (round 5: the
# mixed-precision training step after the fused GroupNorm-backward hand-offs -> default line's train_step.mixed_precision, the
# training lines of configs[1..4] under a 1-rank torchrun, the configs[4] forward line), all at the final commit
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -12 > gpurun_out/r5final_pytest.log
tail -3 gpurun_out/r5final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r5final_bench.json 2> gpurun_out/r5final_bench.err; echo "bench exit $?"
python -c "
import json; d=json.load(open('gpurun_out/r5final_bench.json')); r=d['roofline']; t=d.get('train_step',{})
print('default', d['value'], r['kernel'], round(r['frac'],3), d['cpu_baseline']['parity_gate']['passed'], 'B=2', d.get('small_batch',{}).get('img_per_s'),
      'train', t.get('value'), (t.get('parity_gate') or {}).get('passed'), 'autograd', (t.get('torch_autograd') or {}).get('value'),
      'mixed', (t.get('mixed_precision') or {}).get('value'), ((t.get('mixed_precision') or {}).get('parity_gate') or {}).get('passed'))"
timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --batch-sweep '' --train-steps 0 2>/dev/null | tail -1 > gpurun_out/r5final_bench_configcfg4.json; python -c "
import json; d=json.load(open('gpurun_out/r5final_bench_configcfg4.json')); r=d['roofline']; print('cfg4', d['value'], r['kernel'], r['frac'], d['cpu_baseline']['parity_gate']['passed'])"
for c in cfg1 cfg2 cfg3 cfg4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2965${c: -1} bench.py --gpus 1 --config $c --mode train --steps 6 --warmup 2 --no-probe 2>gpurun_out/r5final_train_$c.err | tail -1 > gpurun_out/r5final_bench_train_$c.json
  python -c "
import json; d=json.load(open('gpurun_out/r5final_bench_train_$c.json')); t=d.get('train_step',{}); print('train $c', d['value'], d['dtype'], (t.get('parity_gate') or {}).get('passed'), (t.get('reducer') or {}).get('exposed_ms'))"
done
