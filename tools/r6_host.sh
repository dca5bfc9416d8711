# tools/r6_host.sh <tag>: the SURVEY 8(d) boundary (host inputs -> bytes in host memory) through every form of the call, uplink on / off
TAG=$1
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
B="--no-cpu-baseline --no-variants --parity-frames 0"
run() { name=$1; shift; echo "== $name" >> $O/host.err; timeout 600 env "$@" python bench.py $B $ARGS > $O/host_$name.json 2>> $O/host.err; tail -c 400 $O/host_$name.json | head -c 0; python - "$O/host_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"],1), "frames/s", round(d["ms_per_step"],1), "ms/step", d["config"]["hbm_in_use_gb_after_timed_steps"], "GB")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
ARGS="--host-inputs --host-pinned --host-enqueued --steps 5 --warmup 1"; run pinned_enqueued UVOL_X=0
ARGS="--host-inputs --host-pinned --host-enqueued --steps 5 --warmup 1"; run pinned_enqueued_uplink_off UVOL_UPLINK=0
ARGS="--host-inputs --host-pinned --steps 3 --warmup 1"; run pinned_blocking UVOL_X=0
ARGS="--host-inputs --host-enqueued --steps 4 --warmup 1"; run pageable_enqueued UVOL_X=0
ARGS="--host-inputs --steps 3 --warmup 1"; run pageable_blocking UVOL_X=0
