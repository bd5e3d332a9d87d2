#!/bin/bash
# Round-3 visit G: small batch (B=2) with 1 / 2 streams, kernel stats at B=2, torchrun 1-rank training line with the reducer timeline.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3g}
timeout 300 python -m pytest tests/test_gpu_train_step.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for s in 1 2; do
  CPR_STREAMS=$s timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 2 --batch-sweep '' 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_streams$s.json
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_bench_streams$s.json'));sb=d['small_batch'];print('streams=$s value', round(d['value'],1), 'small_batch', {k:(round(v['img_per_s'],1) if isinstance(v,dict) else v) for k,v in sb.items() if k in ('eager','hipgraph','img_per_s')})"
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_b2 -- python $OLDPWD/bench.py --batch 2 --steps 20 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}.log 2>&1 )
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
for r in all_reduce reduce_scatter; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --small-batch 0 --batch-sweep '' --train-steps 4 --reducer $r 2>/dev/null | tail -1 > gpurun_out/${TAG}_torchrun1_$r.json
  python -c "import json;d=json.load(open('gpurun_out/${TAG}_torchrun1_$r.json'));print('$r', round(d['value'],1), d.get('train_step'))"
done
