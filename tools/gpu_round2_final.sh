#!/bin/bash
# Round-2 evidence visit: full GPU suite, default bench line, kernel-trace stats of the same command, per-layer conv table,
# PMC passes on the dominant (Winograd) kernel, P2PNet and bf16 bench lines.  Everything lands in gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r2final}
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cut -c1-700 gpurun_out/${TAG}_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}.log 2>&1 )
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
timeout 300 python tools/conv_bench.py --batch 64 > gpurun_out/${TAG}_convbench.txt 2>&1; tail -3 gpurun_out/${TAG}_convbench.txt
PMC_SCRIPT=conv_single.py CONV_ARGS="--b8 --batch 64 --iters 3" bash tools/gpu_pmc.sh ${TAG} > /dev/null 2>&1
python tools/pmc_to_json.py ${TAG} 64 conv_wino_kernel pmc_dominant_kernel.json > /dev/null 2>&1; cp profiles/pmc_dominant_kernel.json gpurun_out/${TAG}_pmc_dominant_kernel.json; cat profiles/pmc_dominant_kernel.json | head -20
for m in "--model p2p --batch 16" "--model p2p --mode infer --batch 16" "--dtype bf16 --batch 64" "--depth 101 --size 1024 --batch 8"; do
  n=$(echo $m | tr -d ' -'); timeout 300 python bench.py $m --steps 8 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-200 gpurun_out/${TAG}_bench_$n.json; echo
done
ls gpurun_out | grep ${TAG}
