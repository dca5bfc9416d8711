#!/usr/bin/env python3
import json, glob, sys
for f in sorted(glob.glob("gpurun_out/sw_%s_*.json" % sys.argv[1])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("sw_%s_" % sys.argv[1])[1][:-5].ljust(18), round(d["value"], 1), round(d["ms_per_step"], 1),
              {a.split(".")[1][:14]: round(b) for a, b in sorted(d["kernel_groups_ms_per_step"].items(), key=lambda kv: -kv[1])[:7]})
    except Exception as e:
        print(f, "ERR", e)
