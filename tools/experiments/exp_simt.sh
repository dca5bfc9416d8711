# tools/experiments/exp_simt.sh — round 2 experiment: wave-per-walker (LDS bitmaps) vs lane-per-walker (UVOL_SIMT_W) geometry walkers
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
UVOL_SIMT_W=4 python -m pytest tests/test_gpu_geom.py -m gpu -x -q > $O/pytest_simt.log 2>&1; echo "pytest rc $?" >> $O/pytest_simt.log
B="python bench.py --no-cpu-baseline --steps 2 --warmup 1"
timeout 600 $B > $O/full_base.json 2> $O/err.log
for w in 0 1 2 4 16 64; do UVOL_SIMT_W=$w timeout 600 $B --only geo --geo-streams 1 --frames-per-step 240 > $O/geo240_w$w.json 2>> $O/err.log; done
for w in 0 1 4 16; do UVOL_SIMT_W=$w timeout 600 $B --only geo --geo-streams 1 --frames-per-step 720 > $O/geo720_w$w.json 2>> $O/err.log; done
for w in 1 4; do UVOL_SIMT_W=$w timeout 600 $B > $O/full_w$w.json 2>> $O/err.log; done
UVOL_SIMT_W=4 timeout 600 $B --geo-streams 1 > $O/full_w4_gs1.json 2>> $O/err.log
UVOL_SIMT_W=4 timeout 600 $B --geo-streams 2 > $O/full_w4_gs2.json 2>> $O/err.log
