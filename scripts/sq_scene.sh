# SQ counters of one scene's kernels (GPU box, repo root): bash scripts/sq_scene.sh <scene> <kernel grep pattern> - three short passes (issue / waits, memory pipes, LDS)
SC=${1:-mis}; PAT=${2:-k_trace}
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
mkdir -p gpurun_out/sqx
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/sqx/a_$SC -- python3 bench.py --scene $SC --steps 1 --warmup 0 --no-cpu-baseline --headline-only > gpurun_out/sqx/a_$SC.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/sqx/a_$SC | grep "$PAT"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/sqx/c_$SC -- python3 bench.py --scene $SC --steps 1 --warmup 0 --no-cpu-baseline --headline-only > gpurun_out/sqx/c_$SC.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/sqx/c_$SC | grep "$PAT"
