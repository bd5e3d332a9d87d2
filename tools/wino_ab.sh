cd /root/repo
timeout 600 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_backward.py tests/test_gpu_wino.py -q -x 2>&1 | tail -4
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-probe 2>/dev/null | tail -1 | cut -c1-330
timeout 300 python bench.py --model p2p --mode train --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-probe 2>/dev/null | tail -1 | cut -c1-330
