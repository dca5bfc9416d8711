#!/bin/bash
# tools/asan_check.sh: the kernels under AddressSanitizer through tests/hipemu (CPU only), in every kernel form the tests switch between
cd "$(dirname "$0")/.." && make -s -C universal-volumetric_amd hipemu-asan || exit 1
ASAN=$(gcc -print-file-name=libasan.so)
for e in "X=0" "UVOL_SIMT_W=5 UVOL_ENTROPY_W=8" "UVOL_RELABEL=1" "UVOL_RELABEL=1 UVOL_SIMT_W=7" "UVOL_WALK_FORCE=vglobal" "UVOL_WALK_FORCE=global" "UVOL_REC16=1" "UVOL_LATE_JOIN=0" "UVOL_SEL_LCAP=16" "UVOL_DD_SLOTS=4" "UVOL_GEO_MIN_GROUP=1 UVOL_TEX_PART=1" "UVOL_GEO_MIN_GROUP=1 UVOL_TEX_PART=1 UVOL_UPLINK_KERNEL=1"; do
  echo "== $e"
  env $e LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python tools/asan_check.py ${1:-30} 2>&1 | grep -v "makecontext" | tail -3
done
