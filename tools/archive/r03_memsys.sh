# usage (GPU box): bash tools/r03_memsys.sh [bench args] -- memory-system counters of the step kernel (L1 = TCP, L2 = TCC, its
# memory-side interface EA, the address unit TA), per-launch medians; a few counters per --pmc pass (more than the hardware can
# schedule aborts rocprofv3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
{
bash tools/prof_counters.sh ms1 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum" "$@"
bash tools/prof_counters.sh ms1b "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" "$@"
bash tools/prof_counters.sh ms2 "TCC_BUSY_avr TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum" "$@"
bash tools/prof_counters.sh ms2b "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" "$@"
bash tools/prof_counters.sh ms3 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "$@"
bash tools/prof_counters.sh ms4 "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "$@"
bash tools/prof_counters.sh ms4b "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "$@"
bash tools/prof_counters.sh ms5 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "$@"
} 2>&1 | grep -v "^Counter_Name" | tee gpurun_out/r03/memsys.log
rm -rf gpurun_out/prof
