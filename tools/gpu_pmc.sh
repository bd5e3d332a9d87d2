#!/bin/bash
# rocprofv3 PMC passes on the dominant kernel (separate passes: SQ has 8 slots, TCC 4; FETCH_SIZE costs 3, WRITE_SIZE 2).
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG=${1:-pmc}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(gpu-agent|Name|SQ_|TCC_|GRBM_|FETCH|WRITE|Mfma|VALU|LDS)" | head -400 > $OUT/${TAG}_counters.txt 2>&1
run() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/${TAG}_$name -o $name -- python $REPO/tools/${PMC_SCRIPT:-conv_single.py} ${CONV_ARGS:-} > /tmp/${TAG}_$name.log 2>&1
  echo "pmc $name exit $?"; tail -2 /tmp/${TAG}_$name.log
  for f in $(find /tmp/${TAG}_$name -name "*counter_collection.csv"); do cp $f $OUT/${TAG}_${name}_counters.csv; done
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -la $OUT | grep ${TAG}
