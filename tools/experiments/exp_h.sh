cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_h; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 900 $B --only tex > $O/tex2160.json 2> $O/err.log
timeout 900 $B > $O/full_default.json 2>> $O/err.log
bash tools/variants.sh r02_h >> $O/err.log 2>&1
