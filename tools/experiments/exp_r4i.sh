# round 4: GPU suite + measurement pack of the build with the flag back in the record, the small dedup tables, texture parts, host inputs at the full step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4i/pytest.log
bash tools/prof_pack.sh r04_b
