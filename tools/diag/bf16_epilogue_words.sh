cd /root/repo
for w in 1 257 513 2079 2591; do for shape in "--batch 64 --hw 160" "--batch 8 --hw 128 --cin 256 --cout 1024 --k 1 --plain --res"; do echo -n "word $w $shape: "; timeout 120 python tools/conv_single.py --bf16 $shape --iters 20 --bf16-dma $w 2>&1 | grep -v amdgpu | tail -1; done; done
