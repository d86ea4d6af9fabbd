#!/bin/bash
# Development aid: the SQ counters of one kernel for several builds of the library.
# usage (GPU box): KERNEL=k_huff_write bash tools/pmc_ab.sh default var_a ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export JDGPU_WWG_WAVES=8
B="python3 bench.py --batch ${PMC_BATCH:-1024} --distinct 16 --steps 1 --warmup 1 --no-cpu-baseline --lean"
for v in "$@"; do
  if [ "$v" = default ]; then unset JPEGAMD_LIB; else export JPEGAMD_LIB=$PWD/jpeg-decoder_amd/build/$v.so; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmcab_${v}_1 -- $B > gpurun_out/pmcab_${v}_1.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcab_${v}_2 -- $B > gpurun_out/pmcab_${v}_2.log 2>&1 || exit 1
  echo "== $v"
  python3 tools/pmc_summary.py "gpurun_out/pmcab_${v}_1/*/*counter_collection.csv" "gpurun_out/pmcab_${v}_2/*/*counter_collection.csv" | awk -v k="${KERNEL:-k_huff_write}" '/^[a-z_]/ {on = index($0, k) > 0} on {print}'
done
