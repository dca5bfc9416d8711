# round 4: GPU suite + measurement pack of the last build (face normals per face; ingest on the device)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4q/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4q/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4q/smoke.log 2>&1
bash tools/prof_pack.sh r04_c
