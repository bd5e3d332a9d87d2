cd /root/repo
p=29700
for rep in 1 2; do
for w in 768 1024 512; do
  p=$((p+1))
  v=$(CPR_WGRAD_TN_WGS=$w timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 1 --config cfg4 --mode train --steps 10 --warmup 3 --no-probe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "CPR_WGRAD_TN_WGS=$w (pass $rep): $v"
done
done
