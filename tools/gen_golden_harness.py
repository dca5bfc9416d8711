#!/usr/bin/env python3
"""tools/gen_golden_harness.py — golden vectors for the HOST counterpart of scripts/Encoder.py.

Imports the reference driver (with stub `commentjson` / `audioread` / `tqdm` modules: the real ones are not
installed and only wrap json / audio probing) IN THE BUILD CONTAINER and records the outputs of its pure
helper functions on a table of inputs.  Only the resulting JSON (data) is committed; nothing of the reference
travels to the GPU box.
"""
import contextlib, io, json, os, sys, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
for name in ("commentjson", "audioread", "tqdm"):
    m = types.ModuleType(name)
    if name == "commentjson":
        import json as _j
        m.load = _j.load; m.dump = _j.dump; m.loads = _j.loads; m.dumps = _j.dumps
    if name == "tqdm":
        m.tqdm = lambda x, **k: x
    sys.modules[name] = m
sys.path.insert(0, os.path.join(REF, "scripts"))
import Encoder as E      # noqa


def run_check(cfg):
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out):
            E.check_all_fields(dict(cfg))
        return ""
    except SystemExit:
        return out.getvalue().strip()


def main():
    gold = {}
    pounds = ["PNG/export_#####.png", "a_###.png", "nohash.png", "x/[#######].ktx2", "f_#.png"]
    gold["convert_pounds_to_c_style"] = {s: E.convert_pounds_to_c_style(s) for s in pounds}
    pats = [("frame_#####.obj", "frame_00001.obj"), ("frame_#####.obj", "frame_0001.obj"), ("frame_#####.obj", "frame_0000a.obj"),
            ("frame_[#######].obj", "frame_0000001.obj"), ("frame_[#######].obj", "frame_[0000001].obj"), ("#####.drc", "00012.drc"),
            ("#####.drc", "00012.drc.bak"), ("texture_#######.ktx2", "texture_0000003.ktx2"), ("a#b", "a1b"), ("a#b", "axb"),
            ("export_###.png", "export_1234.png"), ("#####.drc", "1234.drc")]
    gold["match_pattern"] = [[p, f, bool(E.match_pattern(p, f))] for p, f in pats]
    base = {"name": "n", "GEOMETRY_FRAME_RATE": 30, "TEXTURE_FRAME_RATE": 30, "OutputDirectory": "out", "KTX2_BATCH_SIZE": 5,
            "OBJFilesPath": "OBJ/f_#####.obj", "ImagesPath": "PNG/t_#####.png", "KTX2_FIRST_FILE": 0, "KTX2_FILE_COUNT": 10}
    cases = {"ok": base}
    for k in ("name", "GEOMETRY_FRAME_RATE", "TEXTURE_FRAME_RATE", "OutputDirectory", "KTX2_BATCH_SIZE"):
        c = dict(base); del c[k]; cases["missing_" + k] = c
    c = dict(base); del c["name"]; del c["KTX2_BATCH_SIZE"]; cases["missing_two"] = c
    c = dict(base); del c["OBJFilesPath"]; cases["no_geometry"] = c
    c = dict(base); del c["OBJFilesPath"]; c["DRACOFilesPath"] = "DRC/#####.drc"; cases["draco_only"] = c
    c = dict(base); del c["ImagesPath"]; cases["no_texture"] = c
    c = dict(base); del c["ImagesPath"]; c["KTX2FilesPath"] = "KTX2/#####.ktx2"; cases["ktx2_only"] = c
    c = dict(base); del c["KTX2_FIRST_FILE"]; cases["images_without_first"] = c
    c = dict(base); c["KTX2_FILE_COUNT"] = "10"; cases["images_count_string"] = c
    c = dict(base); c["OBJFilesPath"] = ""; c["ABCFilePath"] = "a.abc"; cases["abc_only"] = c
    c = dict(base); c["name"] = ""; cases["empty_name_is_present"] = c
    gold["check_all_fields"] = {k: {"config": v, "message": run_check(v)} for k, v in cases.items()}
    # frame accounting on the reference's own fixture directories
    out = os.path.join(REF, "example/public/liam/output")
    cfg = {"DRACOFilesPath": os.path.join(out, "geometry_draco", "#####.drc"), "KTX2FilesPath": os.path.join(out, "texture_ktx2-fps30-1k_baseColor_default", "#####.ktx2"),
           "KTX2_BATCH_SIZE": 5, "GEOMETRY_FRAME_RATE": 30, "TEXTURE_FRAME_RATE": 30}
    with contextlib.redirect_stdout(io.StringIO()):
        dur, ng, nseg = E.check_total_frames(cfg)
    gold["check_total_frames_fixture"] = {"durations": dur, "geometry_frames": ng, "segments": nseg, "batch": 5, "rates": [30, 30]}
    # the literal manifest dict of scripts/Encoder.py:311-328 for a given config
    gold["manifest_encoder_py"] = {"version": "v2", "geometry": {"format": "draco", "frameRate": 30, "frameCount": 250, "path": "DRC/#####.drc"},
                                   "texture": {"targets": [{"format": "ktx2", "frameRate": 30, "sequenceCount": 50, "sequenceSize": 5, "path": "KTX2/#####.ktx2"}]}}
    json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "harness", "encoder_py_goldens.json"), "w"), indent=1, sort_keys=True)
    print("wrote harness goldens:", {k: len(v) for k, v in gold.items()})


if __name__ == "__main__":
    main()
