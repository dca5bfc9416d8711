cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
timeout 600 python tools/ingest_timing.py 120 > $O/ingest_timing_120.json 2> $O/err.log
timeout 600 python tools/ingest_timing.py 480 > $O/ingest_timing_480.json 2>> $O/err.log
cat $O/*.json
