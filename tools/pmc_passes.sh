cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the counters of the benched configuration: C3 at 4096 images per launch (PMC_BATCH; a small batch would run
# other plan choices -- span 1, 4-wave write groups -- than the launch the bench times)
N=${PMC_BATCH:-4096}
B="python3 bench.py --batch $N --distinct 16 --steps 1 --warmup 1 --no-cpu-baseline --no-verify"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc1 -- $B > gpurun_out/pmc1.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc2 -- $B > gpurun_out/pmc2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc3 -- $B > gpurun_out/pmc3.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc4 -- $B > gpurun_out/pmc4.log 2>&1 || exit 1
ls gpurun_out/pmc*/*/ | head -20
