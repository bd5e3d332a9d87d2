cd /root/repo
for rep in 1 2; do for a in "--b8" "--b8 --wino-sched 1" "--plain" "--plain --wino-sched 1"; do echo "args: $a"; timeout 120 python tools/conv_single.py --batch 64 --iters 15 $a 2>&1 | grep -v amdgpu; done; done
