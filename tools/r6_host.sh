# tools/r6_host.sh <tag> [forms...]: the SURVEY 8(d) boundary (host inputs -> bytes in host memory) through the forms of the call.
# A form is name:ENV=VAL,ENV=VAL:bench arguments
TAG=$1; shift
ulimit -c 0; export HSA_ENABLE_COREDUMP=0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
B="--no-cpu-baseline --no-variants --parity-frames 0 --host-inputs"
for form in "$@"; do
  name=${form%%:*}; rest=${form#*:}; envs=${rest%%:*}; args=${rest#*:}
  echo "== $name ($envs) $args" >> $O/host.err
  timeout 600 env $(echo $envs | tr ',' ' ') python bench.py $B $args > $O/host_$name.json 2>> $O/host.err
  python - "$O/host_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"],1), "frames/s", round(d["ms_per_step"],1), "ms/step", d["config"]["hbm_in_use_gb_after_timed_steps"], "GB")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
