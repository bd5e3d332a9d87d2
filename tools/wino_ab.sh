cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_cpr_parity.py -q -x 2>&1 | tail -2
PMC_SCRIPT=conv_single.py CONV_ARGS="--b8 --batch 64 --iters 3" bash tools/gpu_pmc.sh r2final2 > /dev/null 2>&1
python tools/pmc_to_json.py r2final2 64 conv_wino_kernel pmc_dominant_kernel.json > /dev/null 2>&1; cp profiles/pmc_dominant_kernel.json gpurun_out/r2final2_pmc_dominant_kernel.json
timeout 600 python bench.py > gpurun_out/r2final2_bench.json 2> gpurun_out/r2final2_bench.err; cut -c1-900 gpurun_out/r2final2_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f2 -o f2 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_f2.log 2>&1 )
find /tmp/prof_f2 -name "*kernel_stats*" -exec cp {} gpurun_out/r2final2_kernel_stats.csv \; 2>/dev/null
