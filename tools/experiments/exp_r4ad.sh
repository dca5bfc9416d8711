# round 4: the default line twice on the final build (variants through uvol_trim instead of fresh contexts); the rest of the pack is r04_d (same kernels)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_e; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_run2.json 2>> $O/bench.err
timeout 300 python -m pytest tests -m gpu -x -q -k "enqueue or lanes or distinct" > $O/pytest_subset.log 2>&1
