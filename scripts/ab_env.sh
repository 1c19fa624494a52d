#!/bin/bash
# A/B of an environment knob on ONE box: bash scripts/ab_env.sh "VAR=value" scene... (first the default, then with the assignment), 2 timed frames each
KNOB=$1; shift
SCENES=${@:-cornell blob mis room}
mkdir -p gpurun_out/ab
for sc in $SCENES; do
  python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only --detail gpurun_out/ab/default_${sc}.json > gpurun_out/ab/default_${sc}.line 2> gpurun_out/ab/default_${sc}.err
  python scripts/ab_line.py default $sc gpurun_out/ab/default_${sc}.json
  env $KNOB python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --headline-only --detail gpurun_out/ab/knob_${sc}.json > gpurun_out/ab/knob_${sc}.line 2> gpurun_out/ab/knob_${sc}.err
  python scripts/ab_line.py "$KNOB" $sc gpurun_out/ab/knob_${sc}.json
done
