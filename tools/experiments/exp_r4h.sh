# round 4: face-visited bits in their own array against the flag inside the record; host-inputs boundary at 2560 frames by texture part size
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_geom.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="python bench.py --only geo --no-variants --no-cpu-baseline --parity-frames 0 --steps 4"
for FB in 1 0; do
  UVOL_FACE_BITS=$FB timeout 300 $B > $O/geo_facebits$FB.json 2>> $O/sweep.err
  UVOL_FACE_BITS=$FB UVOL_GEO_LANES=1 timeout 300 $B > $O/geo_facebits${FB}_lanes1.json 2>> $O/sweep.err
done
UVOL_FACE_BITS=1 timeout 600 python bench.py --no-variants --no-cpu-baseline > $O/bench_facebits1.json 2>> $O/bench.err
UVOL_FACE_BITS=0 timeout 600 python bench.py --no-variants --no-cpu-baseline > $O/bench_facebits0.json 2>> $O/bench.err
for P in 64 0 256; do
  UVOL_TEX_PART=$P timeout 600 python bench.py --host-inputs --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 > $O/host2560_part$P.json 2>> $O/bench.err
done
timeout 600 python bench.py --host-inputs --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 --only geo > $O/host2560_geo_only.json 2>> $O/bench.err
timeout 600 python bench.py --host-inputs --no-variants --no-cpu-baseline --parity-frames 0 --steps 3 --only tex > $O/host2560_tex_only.json 2>> $O/bench.err
