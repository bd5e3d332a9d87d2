cd /root/repo
for rep in 1 2; do
for w in 512 1024 768; do
  echo "== CPR_WGRAD_TN_WGS=$w (pass $rep)"
  CPR_WGRAD_TN_WGS=$w timeout 600 python tools/bf16_ab.py --train --rounds 3 2>&1 | grep -v amdgpu.ids | head -1
done
done
