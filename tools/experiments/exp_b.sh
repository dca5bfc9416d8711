# round 2 experiment B: aliased workspace -> larger batches; auto walker policy
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_b; mkdir -p $O
python -m pytest tests/test_gpu_geom.py tests/test_gpu_cli.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 600 $B > $O/full_base.json 2> $O/err.log
for f in 720 1440 2160; do timeout 900 $B --only geo --geo-streams 1 --frames-per-step $f > $O/geo${f}_gs1.json 2>> $O/err.log; done
for f in 1440 2160; do timeout 900 $B --only geo --geo-streams 2 --frames-per-step $f > $O/geo${f}_gs2.json 2>> $O/err.log; done
timeout 900 $B --only geo --geo-streams 3 --frames-per-step 2160 > $O/geo2160_gs3.json 2>> $O/err.log
for w in 8 32; do UVOL_SIMT_W=$w timeout 900 $B --only geo --geo-streams 1 --frames-per-step 1440 > $O/geo1440_gs1_w$w.json 2>> $O/err.log; done
timeout 900 $B --geo-streams 2 --frames-per-step 1440 > $O/full1440_gs2.json 2>> $O/err.log
timeout 900 $B --geo-streams 2 --frames-per-step 2160 > $O/full2160_gs2.json 2>> $O/err.log
timeout 900 $B --geo-streams 1 --frames-per-step 1440 > $O/full1440_gs1.json 2>> $O/err.log
rocm-smi --showmeminfo vram > $O/mem.txt 2>&1
