# round 4: texture host-input parts on two lanes, GPU-resident ABI forms; GPU suite; host-inputs boundary with part sizes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for P in 24 12 48 0; do
  UVOL_TEX_PART=$P timeout 600 python bench.py --host-inputs --frames-per-step 1080 --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 > $O/host_part$P.json 2>> $O/bench.err
done
UVOL_TIMING=1 timeout 600 python bench.py --host-inputs --frames-per-step 1080 --no-variants --no-cpu-baseline --parity-frames 0 --steps 2 --only geo > $O/host_geo_only.json 2> $O/host_geo_only.err
timeout 600 python bench.py --host-inputs --frames-per-step 1080 --no-variants --no-cpu-baseline --parity-frames 0 --steps 2 --only tex > $O/host_tex_only.json 2>> $O/bench.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2>> $O/bench.err
