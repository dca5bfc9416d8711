# round 4: mesh decode with the arrays fetched to host memory through the staged download (host arrays kept between calls)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ak; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "decode or resident or roundtrip" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 900 python tools/gdec_timing.py 1920 > $O/gdec_timing.json 2> $O/gdec.err
timeout 900 python tools/gdec_timing.py 960 > $O/gdec_timing_960.json 2>> $O/gdec.err
