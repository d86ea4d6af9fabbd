#!/bin/bash
# Development aid: stage times of the lean bench under different environment settings.
# usage (GPU box): bash tools/ab_env.sh "VAR=a" "VAR=b" ...   ("-" = no setting; AB_ARGS: further bench.py arguments)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e JD_BENCH_ABLATION=1 timeout -k 10 160 python3 bench.py --lean --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v'.ljust(24), d['value'], d['ms_per_step'], d['stage_ms'])" || exit 1
done
