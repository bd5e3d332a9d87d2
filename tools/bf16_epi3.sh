#!/bin/bash
# Round 4: tile shapes x epilogue forms of the bf16 LDS-DMA kernel on the R101 1024^2 B = 8 layer shapes (bottleneck conv3 form
# = folded BN + residual + ReLU where --res).  Hook word: 1 + 32 (1 + shape) [+ 1024 = the direct epilogue].
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export CPR_BENCH_HOOKS=1
OUT=gpurun_out/${1:-r4}_bf16_epi3.txt
: > $OUT
run() { echo "## $*" >> $OUT; timeout 120 python tools/conv_single.py --bf16 --plain --iters 30 "$@" 2>&1 | tail -1 >> $OUT; }
for shape in "--res --batch 8 --hw 64 --cin 256 --cout 1024 --k 1" "--res --batch 8 --hw 128 --cin 128 --cout 512 --k 1" \
             "--res --batch 8 --hw 256 --cin 64 --cout 256 --k 1" "--batch 8 --hw 256 --cin 256 --cout 64 --k 1" \
             "--batch 8 --hw 64 --cin 256 --cout 256 --k 3" "--batch 8 --hw 64 --cin 1024 --cout 256 --k 1" \
             "--batch 8 --hw 128 --cin 128 --cout 128 --k 3" "--batch 8 --hw 128 --cin 512 --cout 128 --k 1" \
             "--res --batch 8 --hw 32 --cin 512 --cout 2048 --k 1" "--batch 8 --hw 32 --cin 512 --cout 512 --k 3"; do
  for f in 0 33 65 97 129 1057 1089 1121 1153; do
    run $shape --bf16-dma $f
  done
done
python - <<'P' >> $OUT
import re
L=open('gpurun_out/r4_bf16_epi3.txt').read().strip().split('\n')
rows={}
for i in range(0,len(L)-1,2):
    m=re.match(r'## (.*) --bf16-dma (\d+)',L[i]); ms=re.search(r'([\d.]+) ms',L[i+1])
    if m: rows.setdefault(m.group(1),{})[int(m.group(2))]=float(ms.group(1)) if ms else -1
cols=(0,33,65,97,129,1057,1089,1121,1153)
print('TABLE ms: %-52s' % 'shape' + ' '.join('%6s'%c for c in ('reg','256²','128x256','256x128','128²','d256²','d128x256','d256x128','d128²')))
for k,v in rows.items(): print('TABLE %-56s' % k + ' '.join('%7.3f'%v.get(c,-1) for c in cols))
P
grep TABLE $OUT
