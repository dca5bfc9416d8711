# round 4: GPU suite + smoke + measurement pack of the final build (batched component scan, decode symbol streams on a second stream)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4ab
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4ab/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4ab/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4ab/smoke.log 2>&1
bash tools/prof_pack.sh r04_d
