cd /root/repo
for a in "--plain" "--plain --wino-ablate 16" "--plain --wino-ablate 32"; do echo "args: $a"; timeout 120 python tools/conv_single.py --batch 64 --iters 10 $a 2>&1 | grep -v amdgpu; done
