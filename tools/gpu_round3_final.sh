#!/bin/bash
# Round-3 evidence visit: full GPU suite, smoke, default bench line, kernel-trace stats of the same command (B=64 and B=2), per-layer
# conv table, PMC passes REGENERATED for the dominant (Winograd) kernel and the bf16 DMA kernel, training / bf16 / P2P / R101 lines,
# the mixed-precision step, the bf16 weight gradient and the streamed 1x1 A/B.
# Everything lands in gpurun_out/ (copy the summaries to profiles/).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r3final}
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -30 > gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
# PMC first (the bench replays roofline.traffic from the JSON it finds): dominant kernel, then the bf16 DMA kernel
PMC_SCRIPT=conv_single.py CONV_ARGS="--b8 --batch 64 --iters 3" bash tools/gpu_pmc.sh ${TAG} > /dev/null 2>&1
python tools/pmc_to_json.py ${TAG} 64 conv_wino_kernel pmc_dominant_kernel.json > /dev/null 2>&1; cp profiles/pmc_dominant_kernel.json gpurun_out/${TAG}_pmc_dominant_kernel.json; head -16 profiles/pmc_dominant_kernel.json
for g in sq fetch write; do mv gpurun_out/${TAG}_${g}_counters.csv gpurun_out/${TAG}_wino_${g}_counters.csv 2>/dev/null; done
PMC_SCRIPT=conv_single.py CONV_ARGS="--bf16 --batch 64 --iters 3" bash tools/gpu_pmc.sh ${TAG} > /dev/null 2>&1
python tools/pmc_to_json.py ${TAG} 64 conv_bf16_dma_kernel round3_pmc_bf16_dma_kernel.json > /dev/null 2>&1; cp profiles/round3_pmc_bf16_dma_kernel.json gpurun_out/${TAG}_pmc_bf16_dma_kernel.json
for g in sq fetch write; do mv gpurun_out/${TAG}_${g}_counters.csv gpurun_out/${TAG}_bf16_${g}_counters.csv 2>/dev/null; done
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cut -c1-600 gpurun_out/${TAG}_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_b64 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_b2 -- python $OLDPWD/bench.py --batch 2 --steps 20 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_${TAG}_b2.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG}_train_b64 -- python $OLDPWD/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/prof_${TAG}_train.log 2>&1 )
find /tmp/prof_${TAG} -name "*kernel_stats*" -exec cp {} gpurun_out/ \; 2>/dev/null
timeout 300 python tools/conv_bench.py --batch 64 > gpurun_out/${TAG}_convbench.txt 2>&1; tail -2 gpurun_out/${TAG}_convbench.txt
for m in "--model p2p --batch 16" "--model p2p --mode infer --batch 16" "--dtype bf16 --batch 64" "--depth 101 --size 1024 --batch 8" "--depth 101 --size 1024 --batch 8 --dtype bf16" "--mode train --batch 64"; do
  n=$(echo $m | tr -d ' -'); timeout 400 python bench.py $m --steps 8 --warmup 3 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-200 gpurun_out/${TAG}_bench_$n.json; echo
done
# mixed-precision training step (bf16 compute mode in the trainer) next to the fp32 one, the bf16 weight gradient per layer, the
# streamed 1x1 kernels against the tiled one
for m in "--mode train --batch 64 --dtype bf16" "--mode train --depth 101 --size 1024 --batch 8 --dtype bf16" "--mode train --depth 101 --size 1024 --batch 8"; do
  n=$(echo $m | tr -d ' -'); timeout 600 python bench.py $m --steps 6 --warmup 2 --no-cpu-baseline --no-probe 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_$n.json; cut -c1-200 gpurun_out/${TAG}_bench_$n.json; echo
done
timeout 300 python tools/wgrad_bf16_bench.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_wgrad_bf16_bench.txt
timeout 300 python tests/report_mixed_precision_grads.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_mixed_precision_grad_check.txt
for s in 1 0; do timeout 300 python tools/conv_bench.py --batch 64 --stream $s 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_convbench_stream$s.txt; done
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-probe --small-batch 0 --batch-sweep '' --train-steps 4 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_torchrun_1rank.json
cut -c1-200 gpurun_out/${TAG}_bench_torchrun_1rank.json; echo
ls gpurun_out | grep ${TAG}
