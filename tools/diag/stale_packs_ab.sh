cd /root/repo
# what the per-step re-packs / re-folds cost (CPR_EXPERIMENT_STALE_PACKS=1: stale packs, WRONG results, timing only)
for v in 0 1 0 1; do
  for c in cfg2 cfg3 cfg1; do
    r=$(CPR_EXPERIMENT_STALE_PACKS=$v timeout 600 python bench.py --config $c --mode train --steps 8 --warmup 3 --no-probe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "CPR_EXPERIMENT_STALE_PACKS=$v $c train: $r"
  done
done
