cd /root/repo
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b2 -o b2 -- python /root/repo/bench.py --steps 20 --warmup 3 --batch 2 --no-cpu-baseline --no-probe --train-steps 0 --small-batch 0 --batch-sweep '' > /tmp/prof_b2.log 2>&1; tail -1 /tmp/prof_b2.log | cut -c1-200 )
find /tmp/prof_b2 -name "*kernel_stats*" -exec cp {} gpurun_out/b2_kernel_stats.csv \;
head -25 gpurun_out/b2_kernel_stats.csv | cut -c1-170
