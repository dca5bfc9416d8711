# after the pack: the N > 1 code path of bench.py on one device (two ranks, gloo), walker forms on the round's final build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t; rm -rf $O; mkdir -p $O
UVOL_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --frames-per-step 640 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks_one_device.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_2ranks_one_device.json")); print("2 ranks on one device:", round(d["value"]), "fps n_gpus", d["n_gpus"], "strong_configs3", d.get("strong_configs3"))
except Exception as e: print("2-rank run FAILED", e); print(open("$O/bench_2ranks_one_device.err").read()[-1500:])
PY
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --warmup 1 --steps 4 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "trav", round(g.get("geo.k5_traverse",0)), "walk", round(g.get("geo.k4_eb_walk",0)))
except Exception as e: print("$tag FAILED", e)
PY
}
# two lanes here: geometry alone on three lanes has two regimes (profiles/r05_lane_stagger.json) that would hide the walkers' differences
run forms_lane1 "--only geo" UVOL_GEO_LANES=2 UVOL_TRAV_FORM=lane
run forms_wave1 "--only geo" UVOL_GEO_LANES=2 UVOL_TRAV_W=1
run forms_wave2 "--only geo" UVOL_GEO_LANES=2 UVOL_TRAV_W=2
run forms_wave4 "--only geo" UVOL_GEO_LANES=2 UVOL_TRAV_W=4
run forms_wave8 "--only geo" UVOL_GEO_LANES=2 UVOL_TRAV_W=8
run forms_wave16 "--only geo" UVOL_GEO_LANES=2 UVOL_TRAV_W=16
# stream priorities of the two halves once more, on three lanes
run prio_default "" 
run prio_tex0 "--tex-priority 0"
run prio_geo1_tex0 "--tex-priority 0 --geo-priority 1"
run prio_geo1_tex1 "--geo-priority 1"
