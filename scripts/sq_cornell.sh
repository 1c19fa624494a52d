ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
mkdir -p gpurun_out/sqx
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/sqx/a -- python3 bench.py --scene cornell --steps 1 --warmup 0 --spp 1024 --no-cpu-baseline --headline-only --detail gpurun_out/sqx/a_detail.json > gpurun_out/sqx/a.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/sqx/a | grep "k_shade<1\|k_trace<" 
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/sqx/b -- python3 bench.py --scene cornell --steps 1 --warmup 0 --spp 1024 --no-cpu-baseline --headline-only > gpurun_out/sqx/b.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/sqx/b | grep "k_shade<1\|k_trace<"
