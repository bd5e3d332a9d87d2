cd /root/repo
# the fp32 training lines with the fp32 / Winograd packs refreshed in place (CPR_REFRESH_IN_PLACE=1) or lapsing (0)
for v in 0 1 0 1; do
  for c in cfg2 cfg3 cfg1; do
    r=$(CPR_REFRESH_IN_PLACE=$v timeout 600 python bench.py --config $c --mode train --steps 8 --warmup 3 --no-probe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "CPR_REFRESH_IN_PLACE=$v $c train: $r"
  done
done
