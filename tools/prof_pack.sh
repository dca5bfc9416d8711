# tools/prof_pack.sh <tag>: the measurement pack of a build (run on the GPU box through gpurun); results under gpurun_out/<tag>/
# PACK_PARTS (default "tests bench rocprof pmc tools variants host ranks"): which parts to run (a gpurun call has a time limit; the parts can go in separate calls)
TAG=$1
ulimit -c 0; export HSA_ENABLE_COREDUMP=0          # a GPU fault must not fill the box's disk with a core dump (it did once, and every later step failed)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
PARTS=${PACK_PARTS:-tests bench rocprof pmc tools variants host ranks}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
  (timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/gpu_tests.log
  (timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) >> $O/gpu_tests.log
fi
if has bench; then
  timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
  timeout 900 python bench.py --no-cpu-baseline > $O/bench_run2.json 2>> $O/bench.err
fi
if has rocprof; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2>> $O/bench.err
  cp $(find $O/kt -name bench_kernel_stats.csv | head -1) $O/kernel_stats.csv
  rm -rf $O/kt
fi
# HBM traffic of every kernel: tools/pmc_pack.sh (own processes per half)
if has pmc; then PMC_TIMEOUT=200 bash tools/pmc_pack.sh $TAG; fi
if has tools; then
  python tools/uastc_timing.py 24 > $O/uastc_timing.json 2>> $O/bench.err
  python tools/dec_timing.py 96 > $O/dec_timing.json 2>> $O/bench.err
  python tools/gdec_timing.py 1920 > $O/gdec_timing.json 2>> $O/bench.err
fi
if has variants; then bash tools/variants.sh $TAG >> $O/bench.err 2>&1; fi
# SURVEY 8(d) boundary through every form of the call (tools/forms.sh)
if has host; then
  P="--host-inputs --host-pinned"; E="--host-enqueued --steps 6 --warmup 1"
  bash tools/forms.sh ${TAG}_host "pinned_enqueued:-:$P $E" "pinned_enqueued_run2:-:$P $E" "pinned_blocking:-:$P --steps 3 --warmup 1" "pageable_enqueued:-:--host-inputs $E" "pageable_blocking:-:--host-inputs --steps 3 --warmup 1" \
       "pinned_enqueued_geometry_alone:-:$P $E --only geo" "pinned_enqueued_texture_alone:-:$P $E --only tex" "pinned_enqueued_uplink_off:UVOL_UPLINK=0:$P $E" > $O/host_forms.txt 2>&1
fi
# the N > 1 code path on this one-GPU box: `python bench.py --gpus 2` launches its two ranks itself, both drive device 0, the gather goes over gloo
if has ranks; then
  UVOL_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --frames-per-step 1000 --strong-frames 1200 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_one_device.json 2>> $O/bench.err
fi
