cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_j; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_tex.py -m gpu -x -q -k uastc > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python tools/uastc_timing.py 24 > $O/uastc_timing.json 2> $O/err.log
