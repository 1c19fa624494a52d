# Memory-path counters of one workload's kernels (GPU box): bash scripts/pmc_mem_blob.sh <scene> <spp>
SC=${1:-blob}; SPP=${2:-64}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mem
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\|TCC\)_[A-Z0-9_a-z]*" | sort -u > gpurun_out/mem/avail.txt
wc -l gpurun_out/mem/avail.txt
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
  "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d gpurun_out/mem/p$i -- python3 bench.py --scene $SC --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline --headline-only > gpurun_out/mem/p$i.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/mem/p$i 2>&1 | grep "k_trace_pair<false\|k_trace_top<\|k_trace_quad<true\|k_shade<" | cut -c1-600
done
