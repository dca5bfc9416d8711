# round 4: decode with one staged upload of the call's files
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ag; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "decode or resident" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/gdec_timing.py 1920 > $O/gdec_timing.json 2> $O/gdec.err
timeout 600 python tools/gdec_timing.py 1920 > $O/gdec_timing2.json 2>> $O/gdec.err
