#!/bin/bash
# Round-4 visit A: the whole GPU suite (new: autograd bridge, configs[0] / configs[2] full-size cases, deterministic loss backward),
# smoke, then the bench lines of configs[1] (default), configs[2] and configs[0] at their own shapes.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4a}
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --maxfail=20 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|Hostname\|Librccl" | tail -60 > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit: $?"; cut -c1-400 gpurun_out/${TAG}_bench.json
for c in cfg2 cfg0; do
  timeout 900 python bench.py --config $c --steps 10 --warmup 3 --batch-sweep '' > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  echo "bench $c exit: $?"; cut -c1-400 gpurun_out/${TAG}_bench_$c.json; tail -3 gpurun_out/${TAG}_bench_$c.err
done
ls gpurun_out | grep ${TAG}
