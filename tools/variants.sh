# tools/variants.sh <tag>: the diagnostic workload variants the round-1 verdict asked for, next to the unchanged headline (never `value`)
TAG=${1:-r02}; cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG}_variants; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 --mesh-order shuffled > $O/shuffled_order.json 2> $O/err.log
timeout 900 python bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 --frames-per-step 300 --blocking-calls > $O/job_300_frames.json 2>> $O/err.log
timeout 900 python bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 --frames-per-step 150 --blocking-calls > $O/job_150_frames.json 2>> $O/err.log
timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --total-frames 1200 --blocking-calls > $O/strong_1200_frames_1gpu.json 2>> $O/err.log
timeout 300 python tools/latency.py > $O/single_frame_latency.json 2>> $O/err.log
D=/tmp/uvol_e2e; rm -rf $D; timeout 1500 python tools/e2e_files.py $D 240 > $O/uvolenc_e2e_240.json 2>> $O/err.log; rm -rf $D
timeout 1500 python tools/e2e_files.py $D 960 > $O/uvolenc_e2e_960.json 2>> $O/err.log; rm -rf $D
timeout 1500 python tools/e2e_files.py $D 960 > $O/uvolenc_e2e_960_run2.json 2>> $O/err.log; rm -rf $D
