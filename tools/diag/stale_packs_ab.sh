cd /root/repo
for v in 0 1 0 1; do
  echo "== CPR_EXPERIMENT_STALE_PACKS=$v"
  CPR_EXPERIMENT_STALE_PACKS=$v timeout 600 python tools/bf16_ab.py --train --rounds 2 2>&1 | grep -v amdgpu.ids | head -1
  CPR_EXPERIMENT_STALE_PACKS=$v timeout 600 python tools/bf16_ab.py --train --depth 50 --size 640 --batch 64 --rounds 2 2>&1 | grep -v amdgpu.ids | head -1
done
