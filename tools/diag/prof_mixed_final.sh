cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r6t
( cd /tmp && CPR_TRAIN_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r6t -o cfg4final -- python /root/repo/bench.py --config cfg4 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/p1.log 2>&1 )
( cd /tmp && CPR_TRAIN_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r6t -o r50final -- python /root/repo/bench.py --mode train --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-probe > /tmp/p2.log 2>&1 )
find /tmp/prof_r6t -name "*kernel_stats*" -exec cp {} gpurun_out/r6t/ \;
ls gpurun_out/r6t | grep final
