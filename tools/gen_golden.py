#!/usr/bin/env python3
"""tools/gen_golden.py — regenerate tests/golden/* from the reference's own fixtures.

Runs ONLY in the build container (needs /root/reference).  It decodes every
example/public/liam/output/geometry_draco/*.drc and texture_ktx2-*/*.ktx2 with the CPU oracle and
records counters + CRC32s, and copies a few of the binary fixtures (data
files, not source) so the GPU box can run the same checks without /root/reference.
"""
import json, os, shutil, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O

REF = "/root/reference/example/public/liam/output"
GOLD = os.path.join(ROOT, "tests", "golden")
COPY_DRC = ["00000.drc", "00075.drc"]
COPY_KTX2 = ["00000.ktx2"]


def drc_entry(path):
    b = open(path, "rb").read()
    m = O.drc_decode(b)
    p, u, n = m.att("position"), m.att("tex_coord"), m.att("normal")
    return dict(size=len(b), nev=m.nev, nf=m.nf, nad=m.nad, nsym=m.nsym, nsplit=m.nsplit, nts=m.nts,
                ctx_n=m.ctx_n, conn_end=m.conn_end, hdr_end=m.hdr_end, leftover=m.leftover,
                n_pos=p["n"], n_uv=u["n"], n_nrm=n["n"], n_orient=u["n_orient"], n_flip=n["n_flip_set"],
                seam_uv=u["n_seam_corners"], seam_nrm=n["n_seam_corners"],
                crc_pos="%08x" % O.crc32(p["vals"].astype(np.int32)), crc_uv="%08x" % O.crc32(u["vals"].astype(np.int32)),
                crc_nrm="%08x" % O.crc32(n["vals"].astype(np.int32)), crc_c2v="%08x" % O.crc32(m.c2v.astype(np.int32)),
                sum_pos=int(p["vals"].sum()), sum_uv=int(u["vals"].sum()), sum_nrm=int(n["vals"].sum()))


def main():
    os.makedirs(GOLD, exist_ok=True)
    gd = os.path.join(REF, "geometry_draco")
    files = sorted(f for f in os.listdir(gd) if f.endswith(".drc"))
    drc = {f: drc_entry(os.path.join(gd, f)) for f in files}
    agg = "".join(drc[f]["crc_pos"] + drc[f]["crc_uv"] for f in files)
    out = dict(source="example/public/liam/output/geometry_draco", n_files=len(files),
               aggregate_crc="%08x" % zlib.crc32(agg.encode()), files=drc)
    json.dump(out, open(os.path.join(GOLD, "drc_goldens.json"), "w"), indent=0, sort_keys=True)
    for f in COPY_DRC:
        shutil.copy(os.path.join(gd, f), os.path.join(GOLD, f))
    if hasattr(O, "ktx2_goldens"):
        td = os.path.join(REF, "texture_ktx2-fps30-1k_baseColor_default")
        tfiles = sorted(f for f in os.listdir(td) if f.endswith(".ktx2"))
        kt = {f: O.ktx2_goldens(open(os.path.join(td, f), "rb").read()) for f in tfiles}
        json.dump(dict(source="example/public/liam/output/texture_ktx2-fps30-1k_baseColor_default", n_files=len(tfiles), files=kt),
                  open(os.path.join(GOLD, "ktx2_goldens.json"), "w"), indent=0, sort_keys=True)
        for f in COPY_KTX2:
            shutil.copy(os.path.join(td, f), os.path.join(GOLD, f))
    print("wrote goldens for", len(files), "drc files; aggregate", out["aggregate_crc"])


if __name__ == "__main__":
    main()
