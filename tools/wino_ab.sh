cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | head -10
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r2final3_bench.json 2> gpurun_out/r2final3_bench.err; cut -c1-200 gpurun_out/r2final3_bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python /root/repo/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/prof_t.log 2>&1 )
find /tmp/prof_t -name "*kernel_stats*" -exec cp {} gpurun_out/r2final3_train_kernel_stats.csv \; 2>/dev/null
head -8 gpurun_out/r2final3_train_kernel_stats.csv | cut -c1-150
