#!/usr/bin/env python3
"""tools/e2e_files.py <dir> <frames> [--uastc]: SURVEY 8(d) "with file I/O included" / 8(f)-3: writes a synthetic sequence of <frames>
100,002-vertex OBJ files and 2048^2 PNG files (BASELINE shape), runs `uvolenc` on it from files to .drc / .ktx2 / uvol.json on disk and
prints one JSON line with the end-to-end frames/s (the time uvolenc itself reports for its encode phase, and the wall time of the process)."""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "universal-volumetric_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth
from PIL import Image
root, n = sys.argv[1], int(sys.argv[2]); extra = sys.argv[3:]
os.makedirs(os.path.join(root, "OBJ"), exist_ok=True); os.makedirs(os.path.join(root, "PNG"), exist_ok=True)
t = time.perf_counter()
meshes = [synth.sphere_mesh(frame=k, seed=k) for k in range(4)]
texs = synth.texture_sequence(5, size=2048, seed=0)


def write_obj(path, m):                       # vectorised: 100k-vertex OBJ text in ~1 s
    ip, iu, inn = (m[k].reshape(-1, 3).astype(np.int64) + 1 for k in ("idx_pos", "idx_uv", "idx_nrm"))
    with open(path, "w") as f:
        f.write("\n".join("v %.6f %.6f %.6f" % tuple(v) for v in m["pos"].tolist())); f.write("\n")
        f.write("\n".join("vt %.7f %.7f" % tuple(v) for v in m["uv"].tolist())); f.write("\n")
        f.write("\n".join("vn %.6f %.6f %.6f" % tuple(v) for v in m["nrm"].tolist())); f.write("\n")
        tri = np.stack([ip, iu, inn], -1).reshape(-1, 9)
        f.write("\n".join("f %d/%d/%d %d/%d/%d %d/%d/%d" % tuple(r) for r in tri.tolist())); f.write("\n")


for k in range(4):
    write_obj(os.path.join(root, "OBJ", "src_%d.obj" % k), meshes[k])
for k in range(5):
    Image.fromarray(texs[k], "RGBA").save(os.path.join(root, "PNG", "src_%d.png" % k), compress_level=1)
for k in range(n):                            # distinct files on disk (hard links would let the page cache serve one inode)
    subprocess.check_call(["cp", os.path.join(root, "OBJ", "src_%d.obj" % (k % 4)), os.path.join(root, "OBJ", "frame_%05d.obj" % k)])
    subprocess.check_call(["cp", os.path.join(root, "PNG", "src_%d.png" % (k % 5)), os.path.join(root, "PNG", "export_%05d.png" % k)])
t_gen = time.perf_counter() - t
obj_mb = os.path.getsize(os.path.join(root, "OBJ", "frame_00000.obj")) / 1e6; png_mb = os.path.getsize(os.path.join(root, "PNG", "export_00000.png")) / 1e6
cfg = {"name": "e2e", "OBJFilesPath": os.path.join(root, "OBJ", "frame_#####.obj"), "ImagesPath": os.path.join(root, "PNG", "export_#####.png"),
       "KTX2_FIRST_FILE": 0, "KTX2_FILE_COUNT": n, "KTX2_BATCH_SIZE": 5, "GEOMETRY_FRAME_RATE": 30, "TEXTURE_FRAME_RATE": 30, "OutputDirectory": os.path.join(root, "out")}
json.dump(cfg, open(os.path.join(root, "project-config.json"), "w"))
def quota_cpus():                          # what uvolenc's own default is based on (cgroup v2 quota, else the hardware threads)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return os.cpu_count() if q == "max" else max(1, min(os.cpu_count(), -(-int(q) // int(per))))
    except Exception:
        return os.cpu_count() or 8
threads = "default"
cmd = [os.path.join(ROOT, "universal-volumetric_amd", "bin", "uvolenc"), os.path.join(root, "project-config.json")] + (["--batch-frames", str(min(n, 120))] if "--batch-frames" not in extra else []) + extra
t = time.perf_counter(); r = subprocess.run(cmd, cwd=root, capture_output=True, text=True); wall = time.perf_counter() - t
if os.environ.get("UVOL_TIMING") == "1": sys.stderr.write("".join(l + "\n" for l in r.stderr.splitlines() if "uvolenc-timing" in l))
m = re.search(r"encode phase ([0-9.]+) s, ([0-9.]+) frames/s", r.stdout)
out_dir = os.path.join(root, "out")
print(json.dumps({"what": "uvolenc end to end: %d OBJ (%.1f MB text each) + %d PNG (%.1f MB each) files -> .drc / .ktx2 / uvol.json on disk" % (n, obj_mb, n, png_mb),
                  "rc": r.returncode, "frames": n, "ingest_threads_per_stage": threads, "host_hw_threads": os.cpu_count(), "host_cpus_usable": quota_cpus(), "args": extra,
                  "encode_phase_s": float(m.group(1)) if m else None, "frames_per_s_encode_phase": float(m.group(2)) if m else None,
                  "frames_per_s_process_wall": n / wall, "generate_inputs_s": t_gen,
                  "drc_files": len(os.listdir(os.path.join(out_dir, "geometry_draco"))) if r.returncode == 0 else 0,
                  "tail": r.stdout[-300:] if r.returncode else ""}))
