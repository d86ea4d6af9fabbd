#!/bin/bash
# Development aid: the profile set committed under profiles/ for one state of the kernels.
#   1. rocprofv3 --kernel-trace --stats of the lean bench command (C3, 4096 images per launch)
#   2. the PMC passes of tools/pmc_passes.sh at the same launch size, and the traffic file bench.py reads
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <commit>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --lean --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1 || exit 1
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
grep '^{"metric"' gpurun_out/prof_stats.log > gpurun_out/bench_line_under_rocprof.json
bash tools/pmc_passes.sh || exit 1
python3 tools/pmc_summary.py "gpurun_out/pmc1/*/*counter_collection.csv" "gpurun_out/pmc2/*/*counter_collection.csv" "gpurun_out/pmc3/*/*counter_collection.csv" "gpurun_out/pmc4/*/*counter_collection.csv" > gpurun_out/pmc_summary.txt
python3 tools/pmc_traffic.py gpurun_out/pmc_summary.txt ${PMC_BATCH:-4096} ${1:-unrecorded} > gpurun_out/pmc_traffic.json
cut -c1-300 gpurun_out/bench_line_under_rocprof.json
