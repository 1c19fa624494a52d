#!/bin/bash
mkdir -p gpurun_out/s46
for rep in 1 2 3 4 5 6; do
for q in 1 0; do
    RTX_K0_OVERLAP=$q timeout 300 python bench.py --scene cornell --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s46/k${q}_$rep.json 2> gpurun_out/s46/k${q}_$rep.err
    python scripts/ab_line.py "overlap$q" cornell gpurun_out/s46/k${q}_$rep.json
done
done
