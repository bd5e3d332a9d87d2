#!/bin/bash
# Round-4 evidence visit: full GPU suite, smoke, PMC passes of the dominant kernel (regenerated), the default bench line, kernel-trace
# stats of the same command (B=64 and B=2), one parity-gated roofline-carrying line per BASELINE config at its own shape
# (cfg0 .. cfg4 + R101 fp32), the training line, the 1-rank torchrun line with the N=1 agreement check, P2P kernel stats, LSA phases.
# Everything lands in gpurun_out/ (copy the summaries to profiles/).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4final}
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -30 > gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
PMC_SCRIPT=conv_single.py CONV_ARGS="--b8 --batch 64 --iters 3" bash tools/gpu_pmc.sh ${TAG} > /dev/null 2>&1
python tools/pmc_to_json.py ${TAG} 64 conv_wino_kernel pmc_dominant_kernel.json > /dev/null 2>&1; cp profiles/pmc_dominant_kernel.json gpurun_out/${TAG}_pmc_dominant_kernel.json; head -16 profiles/pmc_dominant_kernel.json
for g in sq fetch write; do mv gpurun_out/${TAG}_${g}_counters.csv gpurun_out/${TAG}_wino_${g}_counters.csv 2>/dev/null; done
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cut -c1-300 gpurun_out/${TAG}_bench.json; echo
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_b64 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_b2 -- python $OLDPWD/bench.py --batch 2 --steps 20 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}_b2.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_cfg2 -- python $OLDPWD/bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}_cfg2.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_p2p -- python $OLDPWD/bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}_p2p.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_cfg4 -- python $OLDPWD/bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}_cfg4.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_train_b64 -- python $OLDPWD/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/prof_${TAG}_train.log 2>&1 )
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
# one line per BASELINE config at its own shape, each with roofline + cpu_baseline (+ parity gate where the mode is fp32 CPR)
for m in "--config cfg0" "--config cfg2" "--config cfg3" "--config cfg3 --mode infer" "--config cfg4" "--depth 101 --size 1024 --batch 8"; do
  n=$(echo $m | tr -d ' -'); timeout 600 python bench.py $m --steps 10 --warmup 3 --batch-sweep '' --train-steps 0 2>gpurun_out/${TAG}_bench_$n.err | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-220 gpurun_out/${TAG}_bench_$n.json; echo
done
for m in "--config cfg4 --no-probe" "--dtype bf16 --no-probe"; do
  n=$(echo $m | tr -d ' -'); timeout 600 python bench.py $m --steps 10 --warmup 3 --no-cpu-baseline --batch-sweep '' --train-steps 0 --small-batch 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-220 gpurun_out/${TAG}_bench_$n.json; echo
done
timeout 600 python bench.py --mode train --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-probe 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_modetrainbatch64.json; cut -c1-200 gpurun_out/${TAG}_bench_modetrainbatch64.json; echo
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-probe --small-batch 0 --batch-sweep '' --train-steps 4 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_torchrun_1rank.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --small-batch 0 --batch-sweep '' --train-steps 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_plain_noprobe.json
python - <<P
import json
a = json.load(open('gpurun_out/${TAG}_bench_torchrun_1rank.json')); b = json.load(open('gpurun_out/${TAG}_bench_plain_noprobe.json'))
r = a['value'] / b['value']
out = dict(torchrun_1rank_img_s=a['value'], plain_img_s=b['value'], ratio=r, within_2_percent=bool(abs(r - 1) <= 0.02),
           n_ranks_seen=a.get('n_ranks_seen'), distinct_devices_seen=a.get('distinct_devices_seen'), rccl_version=a.get('rccl_version'),
           reducer=(a.get('train_step') or {}).get('reducer'))
json.dump(out, open('gpurun_out/${TAG}_torchrun_vs_plain.json', 'w'), indent=1); print(out)
assert out['within_2_percent'], out
P
timeout 120 python tools/lsa_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_lsa_bench.txt; head -3 gpurun_out/${TAG}_lsa_bench.txt
timeout 300 python tools/conv_bench.py --batch 2 > gpurun_out/${TAG}_convbench_b2.txt 2>&1; tail -1 gpurun_out/${TAG}_convbench_b2.txt
ls gpurun_out | grep ${TAG} | wc -l
