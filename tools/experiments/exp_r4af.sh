# round 4: decode path after the LDS tables / byte prefetch of the symbol decoder; full GPU suite + smoke on the final build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4af; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python tools/gdec_timing.py 1920 > $O/gdec_timing.json 2> $O/gdec.err
timeout 600 python tools/gdec_timing.py 1920 > $O/gdec_timing2.json 2>> $O/gdec.err
