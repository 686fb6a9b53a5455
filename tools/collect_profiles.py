"""Copy what one `tools/profile_all.sh <tag>` run left under gpurun_out/<tag>/ into profiles/ (tracked):
    python3 tools/collect_profiles.py <tag> <prefix>        e.g.  r03f_prof r03_a
-> profiles/<prefix>_bench_<workload>.json, profiles/<prefix>_<workload>_summary.txt, and profiles/traffic.json
(HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB of the separate --pmc passes, keyed
"<workload>|<points>|<kernel>", which bench.py's roofline.traffic looks up)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
tpath = os.path.join(dst, "traffic.json")
traffic = {}
if os.path.exists(tpath):
    traffic = json.load(open(tpath))
for name in sorted(os.listdir(src)):
    d = os.path.join(src, name)
    if os.path.isdir(d) and os.path.exists(os.path.join(d, "summary.txt")):
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, "%s_%s_summary.txt" % (prefix, name)))
        txt = open(os.path.join(d, "summary.txt")).read()
        m = re.search(r"HBM bytes per launch.*= ([0-9.e+]+)\s*$", txt, re.M)
        bj = os.path.join(src, "bench_%s.json" % name)
        if m and os.path.exists(bj):
            line = json.loads(open(bj).read().strip().split("\n")[-1])
            key = "%s|%d|%s" % (name, line["config"]["points_per_gpu"], line["config"]["kernel"])
            traffic[key] = {"bytes": float(m.group(1)), "source": "profiles/%s_%s_summary.txt" % (prefix, name)}
    if name.startswith("bench_") and name.endswith(".json"):
        shutil.copy(d, os.path.join(dst, "%s_%s" % (prefix, name)))
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
print("traffic.json:", {k: "%.3e" % v["bytes"] for k, v in traffic.items()})
