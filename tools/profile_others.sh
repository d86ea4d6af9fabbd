#!/bin/bash
# Development aid: rocprofv3 kernel stats of the other BASELINE configs (c2, c4, c5) and the 2-rank
# rehearsal of the sharded config 5 on one GPU.  usage (GPU box, repo root): bash tools/profile_others.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c2 c4 c5; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -- python3 bench.py --lean --no-cpu-baseline --workload $w --steps 5 > gpurun_out/prof_$w.log 2>&1 || exit 1
  find gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${w}_kernel_stats.csv
  grep '^{"metric"' gpurun_out/prof_$w.log > gpurun_out/${w}_bench_line.json
done
# bench.py starts its own two ranks (no launcher): the command line the driver uses, with N = 2
JD_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --workload c5 --steps 5 --warmup 2 > gpurun_out/c5_rehearsal_2ranks.log 2>&1 || exit 1
grep '^{"metric"' gpurun_out/c5_rehearsal_2ranks.log > gpurun_out/c5_rehearsal_2ranks.json
cut -c1-300 gpurun_out/c5_rehearsal_2ranks.json
