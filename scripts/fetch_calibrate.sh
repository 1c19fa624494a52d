#!/bin/bash
# Usage (GPU box, repo root): bash scripts/fetch_calibrate.sh [tag]
# Calibrates FETCH_SIZE / WRITE_SIZE on known byte counts (scripts/micro/fetch_calibrate.hip): a plain timed run, then separate --pmc passes (never combined with tracing).
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_fetch_cal; mkdir -p $OUT
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
BIN=scripts/micro/fetch_calibrate
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN scripts/micro/fetch_calibrate.hip
$BIN $OUT/known.json > $OUT/plain.txt 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- $BIN > $OUT/p$i.log 2>&1
done
python3 scripts/fetch_calibrate_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
