cd $GRAFT_REPO_ROOT; O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3c/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
print(json.dumps(d.get("variants"), indent=1))
print({k: round(v,1) for k,v in d["kernel_groups_ms_per_step"].items()})
PY
