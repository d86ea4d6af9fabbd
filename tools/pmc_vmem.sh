#!/bin/bash
# Development aid: vector-memory pipe counters of one kernel.  usage (GPU box): KERNEL=k_huff_write bash tools/pmc_vmem.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export JDGPU_WWG_WAVES=8
B="python3 bench.py --batch ${PMC_BATCH:-1024} --distinct 16 --steps 1 --warmup 1 --no-cpu-baseline --lean"
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcv_1 -- $B > gpurun_out/pmcv_1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCP_TA_DATA_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES --output-format csv -d gpurun_out/pmcv_2 -- $B > gpurun_out/pmcv_2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY --output-format csv -d gpurun_out/pmcv_3 -- $B > gpurun_out/pmcv_3.log 2>&1 || exit 1
python3 tools/pmc_summary.py "gpurun_out/pmcv_1/*/*counter_collection.csv" "gpurun_out/pmcv_2/*/*counter_collection.csv" "gpurun_out/pmcv_3/*/*counter_collection.csv" | awk -v k="${KERNEL:-k_huff_write}" '/^[a-z_]/ {on = index($0, k) > 0} on {print}'
