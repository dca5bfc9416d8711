# round 4: PNG un-filter asynchronous on its ingest stream (batch k+1 un-filters beside the encode of batch k): parity, stage timing, uvolenc from files
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tex.py -x -q -k "png_scanlines" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python tools/ingest_timing.py 120 > $O/ingest_timing_120.json 2> $O/err.log
D=/tmp/uvol_e2e
rm -rf $D; timeout 1500 python tools/e2e_files.py $D 960 > $O/e2e_960.json 2>> $O/err.log
for A in "" "--host-png-unfilter" "--ingest-threads 8"; do
  rm -rf $D/out; (cd $D && UVOL_TIMING=1 timeout 600 $GRAFT_REPO_ROOT/universal-volumetric_amd/bin/uvolenc project-config.json --batch-frames 120 $A > "$O/e2e_960_$(echo $A | tr ' -' '__').txt" 2> "$O/e2e_960_$(echo $A | tr ' -' '__')_timing.txt")
done
rm -rf $D
tail -3 $O/pytest.log; cat $O/ingest_timing_120.json; grep "frames/s" $O/*.txt
