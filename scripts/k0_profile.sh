ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
for mode in 0 1; do
  RTX_K0_PARALLEL=$mode RTX_K0_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k0prof$mode -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --detail gpurun_out/k0prof${mode}_detail.json > gpurun_out/k0prof$mode.log 2>&1
  f=$(ls gpurun_out/k0prof$mode/*/*_kernel_stats.csv | head -1)
  echo "mode $mode"; grep -i "sampler" $f | cut -c1-160
done
