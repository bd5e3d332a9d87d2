#!/bin/bash
# Round-5 visit A: full GPU suite (incl. the new full-size gradient tests), the default bench line with its gates, one gated line
# per mode that gained a gate this round (bf16, inference, training steps of configs[2..4] under a 1-rank torchrun).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r5a}
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -150 > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cut -c1-260 gpurun_out/${TAG}_bench.json; echo; tail -3 gpurun_out/${TAG}_bench.err
for m in "--config cfg4" "--config cfg3 --mode infer"; do
  n=$(echo $m | tr -d ' -'); timeout 600 python bench.py $m --steps 10 --warmup 3 --batch-sweep '' --train-steps 0 2>gpurun_out/${TAG}_bench_$n.err | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-200 gpurun_out/${TAG}_bench_$n.json; echo; tail -2 gpurun_out/${TAG}_bench_$n.err
done
for c in cfg2 cfg3 cfg4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2963${c: -1} bench.py --gpus 1 --config $c --mode train --steps 5 --warmup 2 --no-probe 2>gpurun_out/${TAG}_train_$c.err | tail -1 > gpurun_out/${TAG}_bench_train_$c.json; cut -c1-200 gpurun_out/${TAG}_bench_train_$c.json; echo; tail -2 gpurun_out/${TAG}_train_$c.err
done
python - <<P
import json, glob
for f in sorted(glob.glob('gpurun_out/${TAG}_bench*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    g = (d.get('cpu_baseline') or {}).get('parity_gate')
    t = (d.get('train_step') or {})
    print(f.split('/')[-1], round(d['value'], 1), 'gate', None if g is None else {k: g.get(k) for k in ('passed', 'max_rel_err', 'cls_feat', 'images', 'error')},
          'train', {k: (t.get(k) if k != 'parity_gate' else {q: (t[k] or {}).get(q) for q in ('passed', 'max_rel_l2', 'global_norm_rel', 'cosine', 'loss_rel', 'worst', 'error', 'oracle_seconds')}) for k in t if k in ('value', 'parity_gate', 'grad_norm')},
          'mixed', ((t.get('mixed_precision') or {}).get('parity_gate') or {}).get('passed'), 'small', (d.get('small_batch') or {}).get('img_per_s'))
P
