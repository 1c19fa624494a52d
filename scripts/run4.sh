#!/bin/bash
mkdir -p gpurun_out/s4
python -m pytest tests -m gpu -x -q > gpurun_out/s4/pytest.log 2>&1; tail -15 gpurun_out/s4/pytest.log
REPS=2 STEPS=3 bash scripts/ab_bench.sh cornell mis > gpurun_out/s4/ab.log 2>&1; tail -8 gpurun_out/s4/ab.log
REPS=1 STEPS=2 bash scripts/ab_bench.sh room blob > gpurun_out/s4/ab2.log 2>&1; tail -8 gpurun_out/s4/ab2.log
for sc in mis-spheres instances-10k; do
  python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s4/${sc}_fast.json 2> gpurun_out/s4/${sc}_fast.err
  python scripts/ab_line.py fast $sc gpurun_out/s4/${sc}_fast.json
  RTX_TRACE_GENERAL=big python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/s4/${sc}_big.json 2> gpurun_out/s4/${sc}_big.err
  python scripts/ab_line.py big $sc gpurun_out/s4/${sc}_big.json
done
tail -3 gpurun_out/s4/*.err
