#!/bin/bash
# Round-6 evidence visit (everything lands in gpurun_out/; the summaries are copied to profiles/round6_*):
#   full GPU suite + smoke | PMC passes (three separate --pmc runs each, never with tracing) of the dominant fp32 kernel and of the
#   bf16 256 x 256 instance | the default bench line | rocprofv3 --kernel-trace --stats of the same command (B=64, B=2, configs[2],
#   P2PNet, configs[4], training step on two streams and on one, the mixed-precision configs[4] step) | one gated line per BASELINE config
#   and mode: forward+loss, P2PNet inference, training steps of configs[1..4] under a 1-rank torchrun (reducer timeline + gradient gate) |
#   torchrun-1-rank vs plain agreement in both modes | the bf16 ping-pong instance's check / ablation table, LSA phases, layer tables.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r6fin}
STAGE=${2:-all}        # all | core (suite, smoke, default bench, PMC) | lines (config / training lines, profiles)
run_bench() { # name, args...
  local n=$1; shift
  timeout 600 python bench.py "$@" 2>gpurun_out/${TAG}_bench_$n.err | tail -1 > gpurun_out/${TAG}_bench_$n.json
  echo "$n: $(cut -c1-200 gpurun_out/${TAG}_bench_$n.json)"
}
run_torchrun() { # name, port, args...
  local n=$1 port=$2; shift 2
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 "$@" 2>gpurun_out/${TAG}_bench_$n.err | tail -1 > gpurun_out/${TAG}_bench_$n.json
  echo "$n: $(cut -c1-200 gpurun_out/${TAG}_bench_$n.json)"
}
prof() { # name, args...
  local n=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_$n -- python $OLDPWD/bench.py "$@" > /tmp/prof_${TAG}_$n.log 2>&1 )
}
if [ "$STAGE" != "lines" ]; then
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -60 > gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
PMC_SCRIPT=conv_single.py CONV_ARGS="--b8 --batch 64 --iters 3" bash tools/gpu_pmc.sh ${TAG} > /dev/null 2>&1
python tools/pmc_to_json.py ${TAG} 64 conv_wino_kernel pmc_dominant_kernel.json > /dev/null 2>&1; cp profiles/pmc_dominant_kernel.json gpurun_out/${TAG}_pmc_dominant_kernel.json; head -18 profiles/pmc_dominant_kernel.json
for g in sq fetch write; do mv gpurun_out/${TAG}_${g}_counters.csv gpurun_out/${TAG}_wino_${g}_counters.csv 2>/dev/null; done
( export ALGO_BYTES=$((2*131072*256*2 + 256*2304*2)) SHAPE_DESC="bf16 3x3 256->256 + GroupNorm statistics on (8,128,128,256): head layer at 1024^2, stride 8"
  PMC_SCRIPT=conv_single.py CONV_ARGS="--bf16 --batch 8 --hw 128 --cin 256 --cout 256 --k 3 --iters 3" bash tools/gpu_pmc.sh ${TAG}bf > /dev/null 2>&1
  python tools/pmc_to_json.py ${TAG}bf 8 "conv_bf16_pp_kernel" round6_pmc_bf16_big_tile.json | grep "mfma_busy\|duration_us\|wait_any\|traffic_over" )
cp profiles/round6_pmc_bf16_big_tile.json gpurun_out/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cut -c1-300 gpurun_out/${TAG}_bench.json; echo
fi
if [ "$STAGE" != "core" ]; then
prof b64 --steps 3 --warmup 1 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep ''
prof b2 --batch 2 --steps 20 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep ''
prof cfg2 --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep ''
prof p2p --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep ''
prof cfg4 --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep ''
prof train_b64 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe
CPR_TRAIN_STREAMS=1 prof train_b64_one_stream --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe
CPR_TRAIN_STREAMS=1 prof train_cfg4_one_stream --config cfg4 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
# one line per BASELINE config at its own shape: roofline + cpu_baseline + the mode's parity gate
run_bench configcfg0 --config cfg0 --steps 10 --warmup 3 --batch-sweep '' --train-steps 0
run_bench configcfg2 --config cfg2 --steps 10 --warmup 3 --batch-sweep '' --train-steps 0
run_bench configcfg3 --config cfg3 --steps 10 --warmup 3 --batch-sweep '' --train-steps 0
run_bench configcfg3modeinfer --config cfg3 --mode infer --steps 10 --warmup 3 --batch-sweep '' --train-steps 0
run_bench configcfg4 --config cfg4 --steps 10 --warmup 3 --batch-sweep '' --train-steps 0
run_bench r101fp32 --depth 101 --size 1024 --batch 8 --steps 10 --warmup 3 --batch-sweep '' --train-steps 0
run_bench configcfg4noprobe --config cfg4 --no-probe --steps 20 --warmup 5 --no-cpu-baseline --batch-sweep '' --train-steps 0 --small-batch 0
run_bench r50bf16noprobe --dtype bf16 --no-probe --steps 10 --warmup 3 --no-cpu-baseline --batch-sweep '' --train-steps 0 --small-batch 0
# training steps under a 1-rank torchrun (the RCCL path: bucket timeline, exposed_ms) with the gradient gate of the cpu_baseline leg
run_torchrun train_cfg1 29641 --mode train --steps 6 --warmup 2 --no-probe
run_torchrun train_cfg2 29642 --config cfg2 --mode train --steps 6 --warmup 2 --no-probe
run_torchrun train_cfg3 29643 --config cfg3 --mode train --steps 6 --warmup 2 --no-probe
run_torchrun train_cfg4 29644 --config cfg4 --mode train --steps 6 --warmup 2 --no-probe
# torchrun-1-rank vs plain, both modes, same flags
run_torchrun torchrun_fwd 29645 --steps 10 --warmup 3 --no-cpu-baseline --no-probe --small-batch 0 --batch-sweep '' --train-steps 0
run_bench plain_fwd --steps 10 --warmup 3 --no-cpu-baseline --no-probe --small-batch 0 --batch-sweep '' --train-steps 0
run_torchrun torchrun_train 29646 --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-probe
run_bench plain_train --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-probe
python - <<P
import json
out = {}
for mode in ('fwd', 'train'):
    a = json.load(open('gpurun_out/${TAG}_bench_torchrun_%s.json' % mode)); b = json.load(open('gpurun_out/${TAG}_bench_plain_%s.json' % mode))
    r = a['value'] / b['value']
    out[mode] = dict(torchrun_1rank_img_s=a['value'], plain_img_s=b['value'], ratio=r, within_2_percent=bool(abs(r - 1) <= 0.02),
                     n_ranks_seen=a.get('n_ranks_seen'), distinct_devices_seen=a.get('distinct_devices_seen'), rccl_version=a.get('rccl_version'),
                     reducer=(a.get('train_step') or {}).get('reducer'))
json.dump(out, open('gpurun_out/${TAG}_torchrun_vs_plain.json', 'w'), indent=1); print({k: (v['ratio'], v['within_2_percent']) for k, v in out.items()})
P
timeout 600 python tools/bf16_pp_check.py > gpurun_out/${TAG}_bf16_pp_check.txt 2>&1; grep -i 'equal' gpurun_out/${TAG}_bf16_pp_check.txt
timeout 300 python tools/conv_bench.py --depth 101 --size 1024 --dtype bf16 --batch 8 > gpurun_out/${TAG}_convbench_cfg4.txt 2>&1; tail -1 gpurun_out/${TAG}_convbench_cfg4.txt
timeout 120 python tools/lsa_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_lsa_bench.txt; head -3 gpurun_out/${TAG}_lsa_bench.txt
timeout 300 python tools/conv_bench.py --batch 2 > gpurun_out/${TAG}_convbench_b2.txt 2>&1; tail -1 gpurun_out/${TAG}_convbench_b2.txt
python - <<P
import json, glob
for f in sorted(glob.glob('gpurun_out/${TAG}_bench*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    g = (d.get('cpu_baseline') or {}).get('parity_gate') or {}
    t = (d.get('train_step') or {})
    tg = t.get('parity_gate') or {}
    print(f.split('/')[-1].replace('${TAG}_bench', ''), round(d['value'], 1), 'gate', g.get('passed'), g.get('max_rel_err'), '| train gate', tg.get('passed'), tg.get('max_rel_l2'), tg.get('global_norm_rel'),
          '| exposed_ms', (t.get('reducer') or {}).get('exposed_ms'), '| train', t.get('value'), (t.get('mixed_precision') or {}).get('value'), (t.get('torch_autograd') or {}).get('value'), '| small', (d.get('small_batch') or {}).get('img_per_s'))
P
fi
ls gpurun_out | grep ${TAG} | wc -l
