# round 4: GPU suite + the measurement pack of the lanes / record-per-face build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4f/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4f/pytest.log
bash tools/prof_pack.sh r04_a
