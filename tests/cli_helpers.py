"""Shared by the CLI tests: write a tiny OBJ + PNG sequence and a project-config.json (with // comments)."""
import json
import os
import numpy as np


def write_obj(path, m, short=False):
    """short: numbers with 9 significant digits (they round-trip float32 and the DEVICE parser takes them); else repr() of the double,
    17 digits, which the device parser hands back to the host parser (more than 2^53 in the mantissa)."""
    n3 = "v %.9g %.9g %.9g\n" if short else "v %r %r %r\n"; t2 = "vt %.9g %.9g\n" if short else "vt %r %r\n"; nn = "vn %.9g %.9g %.9g\n" if short else "vn %r %r %r\n"
    with open(path, "w") as f:
        for v in m["pos"]: f.write(n3 % tuple(float(x) for x in v))
        for v in m["uv"]: f.write(t2 % tuple(float(x) for x in v))
        for v in m["nrm"]: f.write(nn % tuple(float(x) for x in v))
        ip, iu, inn = (m[k].reshape(-1, 3) + 1 for k in ("idx_pos", "idx_uv", "idx_nrm"))
        for a, b, c in zip(ip, iu, inn):
            f.write("f " + " ".join("%d/%d/%d" % (a[k], b[k], c[k]) for k in range(3)) + "\n")


def make_sequence(root, n_frames=10, tex=64, batch=5, comments=True, alpha=False):
    import synth
    from PIL import Image
    os.makedirs(os.path.join(root, "OBJ")); os.makedirs(os.path.join(root, "PNG"))
    meshes = [synth.sphere_mesh(16, 9, charts=(2, 2), frame=k, seed=k) for k in range(n_frames)]
    for k, m in enumerate(meshes):
        write_obj(os.path.join(root, "OBJ", "frame_%05d.obj" % k), m, short=(k % 3 != 1))      # batches mix device-parsed and host-parsed files
    texs = synth.texture_sequence(n_frames, size=tex, seed=3)
    if alpha:                                  # RGBA PNGs with a real alpha channel (a moving soft disc): basisu would write alpha slices
        import numpy as np
        yy, xx = np.mgrid[0:tex, 0:tex]
        texs = [t.copy() for t in texs]
        for k, t in enumerate(texs):
            t[..., 3] = np.clip(((xx - tex // 2 - 2 * k) ** 2 + (yy - tex // 2) ** 2) * (900.0 / tex ** 2), 0, 255).astype(np.uint8)
    for k, t in enumerate(texs):
        Image.fromarray(t, "RGBA").save(os.path.join(root, "PNG", "export_%05d.png" % k))
    cfg = {"name": "test", "OBJFilesPath": os.path.join(root, "OBJ", "frame_#####.obj"), "ImagesPath": os.path.join(root, "PNG", "export_#####.png"),
           "KTX2_FIRST_FILE": 0, "KTX2_FILE_COUNT": n_frames, "KTX2_BATCH_SIZE": batch, "GEOMETRY_FRAME_RATE": 30, "TEXTURE_FRAME_RATE": 30,
           "OutputDirectory": os.path.join(root, "out"), "Q_POSITION_ATTR": 11}
    text = json.dumps(cfg, indent=1)
    if comments:
        text = text.replace('"Q_POSITION_ATTR": 11', '"Q_POSITION_ATTR": 11 // quantization bits for the position attribute, default=11.')
    p = os.path.join(root, "project-config.json")
    open(p, "w").write(text)
    return p, cfg, meshes, texs
