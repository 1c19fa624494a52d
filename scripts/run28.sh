#!/bin/bash
mkdir -p gpurun_out/s28
for rep in 1 2 3; do
for b in 19 18 17; do
    RTX_BATCH_LOG2=$b timeout 300 python bench.py --scene cornell --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s28/b${b}_$rep.json 2> gpurun_out/s28/b${b}_$rep.err
    python scripts/ab_line.py "batch$b" cornell gpurun_out/s28/b${b}_$rep.json
done
done
for b in 19 18; do
    RTX_BATCH_LOG2=$b timeout 300 python bench.py --scene blob --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s28/bb${b}.json 2> gpurun_out/s28/bb${b}.err
    python scripts/ab_line.py "batch$b" blob gpurun_out/s28/bb${b}.json
done
