#!/bin/bash
# usage: tools/sweep.sh TAG "F:G:ONLY:CACHE" ...   (runs on the GPU box; one bench line per spec into gpurun_out/)
TAG=$1; shift
for c in "$@"; do
  IFS=: read F G ONLY CACHE STEPS STAG <<< "$c"
  if [ -n "$CACHE" ] && [ "$CACHE" != d ]; then export UVOL_WALK_CACHE=$CACHE; else unset UVOL_WALK_CACHE; fi
  O=""; [ -n "$ONLY" ] && [ "$ONLY" != all ] && O="--only $ONLY"
  timeout 300 python bench.py $EXTRA --no-cpu-baseline --steps ${STEPS:-3} --warmup 1 --frames-per-step $F --geo-streams ${G:-1} --geo-stagger-ms ${STAG:-0} $O > gpurun_out/sw_${TAG}_$c.json 2> gpurun_out/sw_${TAG}_$c.err
done
