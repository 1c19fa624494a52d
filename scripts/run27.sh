#!/bin/bash
mkdir -p gpurun_out/s27
for rep in 1 2; do
for r in 16 8 12 24 32; do
    RTX_REFILL_MIN=$r timeout 300 python bench.py --scene blob --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s27/r${r}_$rep.json 2> gpurun_out/s27/r${r}_$rep.err
    python scripts/ab_line.py "refill$r" blob gpurun_out/s27/r${r}_$rep.json
done
done
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
for r in 16 8 32; do
  export RTX_REFILL_MIN=$r
  rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/s27/sq$r -- python3 bench.py --scene blob --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --headline-only > gpurun_out/s27/sq$r.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/s27/sq$r | grep k_trace_pair
done
