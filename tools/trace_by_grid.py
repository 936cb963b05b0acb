#!/usr/bin/env python
"""Group a rocprofv3 kernel trace (CSV) by (kernel, grid): python tools/trace_by_grid.py <dir> [substr]"""
import csv, glob, os, sys, collections, re
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if sub and sub not in n: continue
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = n.split("(")[0][:60]
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[0]:60s} grid=({k[1]},{k[2]},{k[3]}) wg={k[4]} n={len(v):4d} med={v2[len(v2)//2]:9.1f} us  total={sum(v)/1e3:8.3f} ms")
