cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_i; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_tex.py -m gpu -x -q -k uastc > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python tools/uastc_timing.py 24 > $O/uastc_timing.json 2> $O/err.log
D=/tmp/uvol_e2e
for cfg in "60 64" "120 128" "48 32"; do set -- $cfg; rm -rf $D; python - $D 240 $1 $2 > $O/e2e_b$1_t$2.json 2>> $O/err.log <<'PY'
import sys, os, re, json, subprocess, time
sys.argv = [sys.argv[0]] + sys.argv[1:]
root, n, bf, th = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
ROOT = os.environ["GRAFT_REPO_ROOT"]
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_files.py"), root, str(n)], capture_output=True, text=True)
cmd = [os.path.join(ROOT, "universal-volumetric_amd", "bin", "uvolenc"), os.path.join(root, "project-config.json"), "--batch-frames", bf, "--ingest-threads", th]
import shutil; shutil.rmtree(os.path.join(root, "out"), ignore_errors=True)
t = time.perf_counter(); q = subprocess.run(cmd, cwd=root, capture_output=True, text=True); wall = time.perf_counter() - t
m = re.search(r"encode phase ([0-9.]+) s, ([0-9.]+) frames/s", q.stdout)
print(json.dumps({"batch_frames": int(bf), "ingest_threads": int(th), "rc": q.returncode, "encode_phase_s": float(m.group(1)) if m else None, "fps": float(m.group(2)) if m else None, "wall_fps": n / wall}))
PY
done; rm -rf $D
