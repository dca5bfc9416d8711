# round 4: texture decode with host outputs through the staged download; GPU texture tests; mesh decode timing once more
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4al; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_tex.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/dec_timing.py 96 > $O/dec_timing.json 2> $O/dec.err
timeout 600 python tools/uastc_timing.py 24 > $O/uastc_timing.json 2>> $O/dec.err
