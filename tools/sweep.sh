#!/bin/bash
# usage (on the GPU box): [EXTRA="bench flags"] tools/sweep.sh TAG "F:G:ONLY:-:STEPS:STAGGER_MS" ...
#   F frames per step, G geometry streams, ONLY = geo | tex | all; one bench JSON line per spec into gpurun_out/sw_TAG_<spec>.json
#   (summarise with tools/sweep_show.py TAG)
TAG=$1; shift
for c in "$@"; do
  IFS=: read F G ONLY _UNUSED STEPS STAG <<< "$c"
  O=""; [ -n "$ONLY" ] && [ "$ONLY" != all ] && O="--only $ONLY"
  timeout 300 python bench.py $EXTRA --no-cpu-baseline --steps ${STEPS:-3} --warmup 1 --frames-per-step $F --geo-streams ${G:-1} --geo-stagger-ms ${STAG:-0} $O > gpurun_out/sw_${TAG}_$c.json 2> gpurun_out/sw_${TAG}_$c.err
done
