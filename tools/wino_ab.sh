cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_wino.py -q -x 2>&1 | tail -3
for a in "--plain" "" "--b8"; do echo "args: $a"; timeout 120 python tools/conv_single.py --batch 64 --iters 10 $a 2>&1 | grep -v amdgpu; done
