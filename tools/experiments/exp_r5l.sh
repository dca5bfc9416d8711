# XCD census again (flushes, error checks), host link rate out of page-locked memory, HBM in use by the halves of the default step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_l; rm -rf $O; mkdir -p $O
chmod +x tools/xcd/xcd_census; timeout 120 tools/xcd/xcd_census > $O/xcd_census.json 2> $O/xcd_census.err; echo "census rc $?"; head -c 2500 $O/xcd_census.json; head -c 600 $O/xcd_census.err
timeout 300 python tools/h2d_rate.py > $O/h2d_rate.json 2> $O/h2d_rate.err; cat $O/h2d_rate.json
run() { tag=$1; shift; args=$1; shift; env "$@" timeout 900 python bench.py $args --no-variants --no-cpu-baseline --steps 2 --warmup 1 --parity-frames 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY
import json
try:
    d=json.load(open("$O/$tag.json")); g=d["kernel_groups_ms_per_step"]
    print("$tag", round(d["value"]), "fps  ms/step", round(d["ms_per_step"]), "hbm GB", d["config"].get("hbm_in_use_gb_after_timed_steps"))
except Exception as e: print("$tag FAILED", e); print(open("$O/$tag.err").read()[-600:])
PY
}
run full ""
run tex "--only tex"
run geo "--only geo"
run geo3840 "--only geo --frames-per-step 3840"
run geo4480 "--only geo --frames-per-step 4480"
run full3520 "--frames-per-step 3520"
