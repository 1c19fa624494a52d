#!/bin/bash
# Usage (GPU box, repo root): bash scripts/pmc_scene.sh <scene> <spp> <tag>
# Separate rocprofv3 --pmc passes (never combined with tracing) for one bench.py workload; summaries -> gpurun_out/<tag>/
SCENE=${1:-blob}; SPP=${2:-16}; TAG=${3:-pmc_$SCENE}
OUT=gpurun_out/$TAG; mkdir -p $OUT
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
export RTX_K0_OVERLAP=0
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum" \
  "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 bench.py --scene $SCENE --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline --headline-only > $OUT/p$i.log 2>&1
  python3 scripts/pmc_summary.py $OUT/p$i > $OUT/p$i.txt 2>&1
done
cat $OUT/p*.txt
