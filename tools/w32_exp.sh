cd /root/repo
for w in "0,2,100" "0,1,100" "0,2,0" "0,2,200" "1,2,100" "2,2,100" "3,2,100" "4,2,100" "8,2,100" "15,2,100"; do
  echo -n "w32=$w: "; CPR_WINO_TILE=32 timeout 120 python tools/conv_single.py --b8 --batch 64 --iters 10 --w32 $w 2>&1 | grep -v amdgpu | tail -1
done
