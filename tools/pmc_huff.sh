# PMC passes over the Huffman kernels at a batch that fills the chip (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --batch 2048 --distinct 32 --steps 1 --warmup 1 --no-cpu-baseline"
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcH1 -- $B > gpurun_out/pmcH1.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/pmcH2 -- $B > gpurun_out/pmcH2.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmcH3 -- $B > gpurun_out/pmcH3.log 2>&1 || exit 1
ls gpurun_out/pmcH*/*/ | head
