# round 4: GPU decode tests after the lying-header case was added
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4ap; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "decode" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
