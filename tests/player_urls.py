"""tests/player_urls.py — the same restatement as tests/player_urls.js (reference src/V2/player.ts:141-174, :207-222, src/utils.ts:26-32)
in Python, so that the URL / target-selection check also runs where node is not installed (the GPU box).  Test infrastructure."""
import json

EXT = {"mp3": ".mp3", "draco": ".drc", "ktx2": ".ktx2", "etc2": ".etc2"}
PRIORITY = {"ktx2": 0, "etc2": 1, "etc1": 2}


def _resolve(template, inputs, n):
    w = template.count("#")
    inputs = dict(inputs); inputs["[" + "#" * w + "]"] = str(n).zfill(w)
    p = template
    for k, v in inputs.items():
        p = p.replace(k, v, 1)          # String.replace replaces the first occurrence only
    return p


def resolve(manifest_path, supports_etc2=False):
    m = json.load(open(manifest_path))
    assert m["version"] == "v2"
    g_target = list(m["geometry"]["targets"].keys())[0]
    names = list(m["texture"]["targets"].keys())
    t_target = names[0]
    for t in sorted(names, key=lambda a: -PRIORITY[m["texture"]["targets"][a]["format"]]):      # stable, like Array.prototype.sort in node >= 11
        if t in ("ktx2", "mp4") or (t == "etc2" and supports_etc2):
            t_target = t
            break
    g = m["geometry"]["targets"][g_target]; t = m["texture"]["targets"][t_target]
    return dict(geometryTarget=g_target, textureTarget=t_target, textureFormat=t["format"], batchSize=t["sequenceSize"], resolution=t["resolution"],
                geometry=[_resolve(m["geometry"]["path"], {"[target]": g_target, "[ext]": EXT[g["format"]]}, i) for i in range(g["frameCount"])],
                texture=[_resolve(m["texture"]["path"], {"[target]": t_target, "[type]": "baseColor", "[tag]": "default", "[ext]": EXT[t["format"]]}, s) for s in range(t["sequenceCount"])])
