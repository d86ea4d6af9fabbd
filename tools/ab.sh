#!/bin/bash
# Development aid: stage times of the lean C3 bench for several builds of the library.
# usage (GPU box): bash tools/ab.sh default var_a var_b ...   (names under jpeg-decoder_amd/build/, "default" = the in-tree library)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = default ]; then unset JPEGAMD_LIB; else export JPEGAMD_LIB=$PWD/jpeg-decoder_amd/build/$v.so; fi
  JD_BENCH_ABLATION=1 timeout -k 10 120 python3 bench.py --lean --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v'.ljust(16), d['value'], d['ms_per_step'], d['stage_ms'])" || exit 1
done
