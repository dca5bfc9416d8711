# tools/forms.sh <tag> form [form ...]: ONE parametrised runner for measurement sweeps on the GPU box (replaces the one-off scripts of rounds 1 - 5;
# VERDICT r5 item 9).  Run through gpurun:  gpurun -- 'bash tools/forms.sh r06_x "name:ENV=VAL,ENV=VAL:bench arguments" ...'
#   form  = name:environment:arguments   (environment: comma-separated UVOL_* / other variables, "-" for none; arguments: bench.py options)
#   FORMS_BASE (environment of this script) = bench.py options every form gets (default: no variants, no CPU baseline, no parity sample)
# Every form writes gpurun_out/<tag>/<name>.json (the bench line) and prints one summary line.
TAG=$1; shift
ulimit -c 0; export HSA_ENABLE_COREDUMP=0          # a GPU fault must not fill the box's disk with a core dump
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
B=${FORMS_BASE:---no-cpu-baseline --no-variants --parity-frames 0}
for form in "$@"; do
  name=${form%%:*}; rest=${form#*:}; envs=${rest%%:*}; args=${rest#*:}
  [ "$envs" = "-" ] && envs="FORMS_NOENV=1"
  echo "== $name ($envs) $args" >> $O/forms.err
  timeout ${FORMS_TIMEOUT:-900} env $(echo $envs | tr ',' ' ') python bench.py $B $args > $O/$name.json 2>> $O/forms.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); g = d["kernel_groups_ms_per_step"]
    print(sys.argv[2], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 1), "ms/step", d["config"]["hbm_in_use_gb_after_timed_steps"], "GB",
          "walk", round(g.get("geo.k4_eb_walk", 0)), "trav", round(g.get("geo.k5_traverse", 0)), "ent", round(g.get("geo.k7_entropy_encode", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
