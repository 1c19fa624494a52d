#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile_round.sh <tag> [scene]
# Kernel-trace summary of the benchmark command + two separate PMC passes (never combined with tracing).
TAG=${1:-r01}
SCENE=${2:-cornell}
OUT=gpurun_out/${TAG}_${SCENE}
mkdir -p $OUT
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
# PMC passes run at the scene's full spp: a pass holds 2^29 paths = all 1024 samples of a 2^19-pixel batch, so only the full sample count gives the launch size of
# the timed run (roofline.traffic is bytes PER LAUNCH)
PSPP=1024
if [ "$SCENE" = "blob" ]; then PSPP=256; fi
if [ "$SCENE" = "mis" ]; then PSPP=512; fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --scene $SCENE --steps 1 --warmup 0 --spp $PSPP --no-cpu-baseline --headline-only --detail $OUT/fetch_detail.json > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --scene $SCENE --steps 1 --warmup 0 --spp $PSPP --no-cpu-baseline --headline-only --detail $OUT/write_detail.json > $OUT/write.log 2>&1
# one SQ pass (8 counters): lanes per VALU instruction, VALU instructions per unit of work, how much of a wave's life it waits / issues
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 bench.py --scene $SCENE --steps 1 --warmup 0 --spp $PSPP --no-cpu-baseline --headline-only --detail $OUT/sq_detail.json > $OUT/sq.log 2>&1
python3 scripts/profile_round.py $TAG $SCENE
# the traced bench line last: it then compares with the PMC file of these very kernels (no stale-traffic warning in it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --scene $SCENE --steps 2 --warmup 1 --no-cpu-baseline --headline-only --detail $OUT/stats_detail.json > $OUT/stats.log 2>&1
python3 scripts/profile_round.py $TAG $SCENE
