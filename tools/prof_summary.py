#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/<dir>/...) into small, committed summaries under profiles/.

  python tools/prof_summary.py stats  gpurun_out/prof_stats  profiles/r01_kernel_stats.md
  python tools/prof_summary.py pmc    gpurun_out/prof_fetch gpurun_out/prof_write profiles/r01_hbm_traffic.md

The PMC summary applies the corrections of MI355X_MICROARCH.md (HBM section): counter values are KiB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads -> doubled ("corrected");
WRITE_SIZE is reported as measured (uncalibrated).
"""

import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:80]


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[0]


def provenance():
    """One header line for every committed summary: the commit the snapshot was taken from (HEAD_REV, exported by
    the caller on the build machine -- the GPU box has no .git) and the sha256 over ALL kernel sources, computed where the profile ran."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "gsoc-wav2vec2_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "gsoc-wav2vec2_amd", "csrc", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return (f"_provenance: git HEAD {os.environ.get('HEAD_REV', 'unknown (HEAD_REV not exported)')}; sha256 over csrc/*.hip + *.h ({len(files)} files) "
            f"{h.hexdigest()[:16]}; command: {os.environ.get('PROF_CMD', 'n/a')}_")


def stats(src, dst):
    rows = list(csv.DictReader(open(find(src, "_kernel_stats.csv"))))
    trace = list(csv.DictReader(open(find(src, "_kernel_trace.csv"))))
    out = [provenance(), "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows:
        if float(r["Percentage"]) < 0.01:
            continue
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                   f"{float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    # GEMM launches by launch geometry (one row per distinct shape)
    by = defaultdict(list)
    for t in trace:
        if "gemm_f32" in t["Kernel_Name"] or "gemm_bf16" in t["Kernel_Name"]:
            key = (int(t["Grid_Size_X"]) // int(t["Workgroup_Size_X"]), int(t["Grid_Size_Z"]))
            by[key].append((int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3)
    out += ["", "GEMM launches by geometry (tiles x batch):", "", "| tiles | batch | calls | avg us | min us |", "|---|---|---|---|---|"]
    for k, v in sorted(by.items()):
        out.append(f"| {k[0]} | {k[1]} | {len(v)} | {sum(v)/len(v):.1f} | {min(v):.1f} |")
    regs = {}
    for t in trace:
        n = short(t["Kernel_Name"])
        if n.startswith("w2v2::"):
            regs[n] = (t["VGPR_Count"], t["Accum_VGPR_Count"], t["SGPR_Count"], t["LDS_Block_Size"], t["Scratch_Size"])
    out += ["", "| kernel | VGPR | AGPR | SGPR | LDS B | scratch |", "|---|---|---|---|---|---|"]
    for n, r in sorted(regs.items()):
        out.append(f"| `{n}` | {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]} |")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


def pmc(fetch_dir, write_dir, dst):
    acc = defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "nw": 0})
    for d, cname in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
            if r["Counter_Name"] != cname or not short(r["Kernel_Name"]).startswith("w2v2::"):
                continue
            a = acc[short(r["Kernel_Name"])]
            a[cname] += float(r["Counter_Value"])
            a["n" if cname == "FETCH_SIZE" else "nw"] += 1
    out = ["| kernel | launches | FETCH raw MB/launch | FETCH corrected (x2) MB/launch | WRITE MB/launch |", "|---|---|---|---|---|"]
    js = {}
    for k, a in sorted(acc.items()):
        n, nw = max(a["n"], 1), max(a["nw"], 1)
        f = a["FETCH_SIZE"] * 1024 / n / 1e6
        w = a["WRITE_SIZE"] * 1024 / nw / 1e6
        out.append(f"| `{k}` | {a['n']} | {f:.2f} | {2*f:.2f} | {w:.2f} |")
        js[k] = {"launches": a["n"], "fetch_raw_bytes": f * 1e6, "fetch_corrected_bytes": 2 * f * 1e6, "write_bytes": w * 1e6}
    open(dst, "w").write("\n".join(out) + "\n")
    json.dump(js, open(os.path.splitext(dst)[0] + ".json", "w"), indent=1)
    print("\n".join(out))


def gaps(src, dst, steps):
    """Device-side idle time between consecutive kernels of the trace (rocprofv3 --kernel-trace): the launch gaps a hipGraph could
    at best remove.  `steps` = benchmark steps in the trace (warm-up included); the first step is dropped (allocation, first-use)."""
    trace = list(csv.DictReader(open(find(src, "_kernel_trace.csv"))))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in trace)
    ev = [e for e in ev if e[2].startswith("w2v2::") or e[2].startswith("__amd_rocclr")]
    # step boundaries: the optimizer launch closes a training step; a forward-only trace is cut evenly
    closers = [i for i, e in enumerate(ev) if "adam_multi_kernel" in e[2]]
    if len(closers) >= 2:
        lo, hi = closers[0] + 1, closers[-1] + 1            # whole steps between the first and the last optimizer launch
        nsteps = len(closers) - 1
    else:
        per = len(ev) // max(1, steps)
        lo, hi, nsteps = per, per * steps, steps - 1
    seg = ev[lo:hi]
    busy = sum(e[1] - e[0] for e in seg)
    g = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
    span = seg[-1][1] - seg[0][0]
    g_sorted = sorted(g)
    def pct(q): return g_sorted[min(len(g_sorted) - 1, int(q * len(g_sorted)))] / 1e3
    big = sorted(((g[i], seg[i][2], seg[i + 1][2]) for i in range(len(g))), reverse=True)[:8]
    out = [f"# device-side launch gaps ({nsteps} steps, {len(seg)} kernel launches, from the rocprofv3 kernel trace)", "",
           f"* launches per step: {len(seg) / nsteps:.0f}",
           f"* span per step: {span / nsteps / 1e6:.3f} ms; kernels busy: {busy / nsteps / 1e6:.3f} ms; idle between kernels: {sum(g) / nsteps / 1e6:.3f} ms",
           f"* gap per launch: median {pct(0.5):.2f} us, p90 {pct(0.9):.2f} us, p99 {pct(0.99):.2f} us, mean {sum(g) / len(g) / 1e3:.2f} us",
           "", "| largest gaps (us) | after | before |", "|---|---|---|"]
    out += [f"| {a / 1e3:.1f} | `{b}` | `{c}` |" for a, b, c in big]
    # runtime fill / copy kernels (hipMemsetAsync, hipMemcpyAsync): which launches they sit between
    ctx = defaultdict(int)
    for i, e in enumerate(seg):
        if e[2].startswith("__amd_rocclr"):
            ctx[(e[2], seg[i - 1][2] if i else "-", seg[i + 1][2] if i + 1 < len(seg) else "-")] += 1
    if ctx:
        out += ["", "| runtime kernel | after | before | per step |", "|---|---|---|---|"]
        out += [f"| `{k[0]}` | `{k[1]}` | `{k[2]}` | {v / nsteps:.1f} |" for k, v in sorted(ctx.items(), key=lambda kv: -kv[1])[:16]]
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:6]))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
