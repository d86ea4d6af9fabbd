#!/bin/bash
# Development aid: the profile set committed under profiles/ for one state of the kernels.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (the judged line)
#   2. the PMC passes of tools/pmc_passes.sh at 256 images per launch
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --lean --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1 || exit 1
bash tools/pmc_passes.sh || exit 1
python3 tools/pmc_summary.py "gpurun_out/pmc1/*/*counter_collection.csv" "gpurun_out/pmc2/*/*counter_collection.csv" "gpurun_out/pmc3/*/*counter_collection.csv" "gpurun_out/pmc4/*/*counter_collection.csv" > gpurun_out/pmc_summary.txt
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
tail -1 gpurun_out/prof_stats.log | cut -c1-400
