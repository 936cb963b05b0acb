#!/bin/bash
# Refresh the committed PMC figures bench.py quotes (roofline.traffic) and the PMC cross-check of the shader clock, on the GPU box:
#   tools/pmc_traffic.sh <round>      -> gpurun_out/pmc_traffic/{hbm_traffic_latest.json, hbm_traffic_bf16.json, clock_pmc.md}
# Counters are collected in their own passes (no tracing domains mixed in); FETCH_SIZE is in KiB and x 2 on gfx950 (MI355X_MICROARCH.md).
# Each JSON gets a _meta block: the round and the sha256 of the kernel sources the figures were measured on -- bench.py refuses a figure
# whose sources have changed since (measured_traffic).  Copy the three files into profiles/ afterwards.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_traffic; mkdir -p $O; round=${1:-0}
cd /tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-side"
run() { tag=$1; shift; timeout 500 rocprofv3 "$@" --output-format csv -d $O/raw_$tag -- $B ${EXTRA:-} > $O/raw_$tag.log 2>&1 || echo "pass $tag failed: $(tail -2 $O/raw_$tag.log)"; }
EXTRA="" run f32_fetch --pmc FETCH_SIZE
EXTRA="" run f32_write --pmc WRITE_SIZE
EXTRA="" run f32_clk --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
EXTRA="" run f32_trace --kernel-trace
EXTRA="--precision bf16" run b16f_fetch --pmc FETCH_SIZE
EXTRA="--precision bf16" run b16f_write --pmc WRITE_SIZE
EXTRA="--precision bf16 --mode train" run b16t_fetch --pmc FETCH_SIZE
EXTRA="--precision bf16 --mode train" run b16t_write --pmc WRITE_SIZE
EXTRA="--precision bf16" run b16f_clk --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
EXTRA="--precision bf16" run b16f_trace --kernel-trace
cd $R
python - "$O" "$round" <<'PY'
import csv, glob, hashlib, json, os, re, sys, collections
O, rnd = sys.argv[1], int(sys.argv[2])
ROOT = os.path.dirname(os.path.dirname(O))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
def short(n): return re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "")).split("(")[0]
def counters(tag, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{O}/raw_{tag}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and short(r["Kernel_Name"]).startswith("w2v2::"):
                a = acc[short(r["Kernel_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc
def durations(tag):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{O}/raw_{tag}/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith("w2v2::"):
                a = acc[k]; a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a[1] += 1
    return acc
def meta(prec): return {"round": rnd, "kernel_source_sha16": bench.kernel_source_hash(prec), "kernel_sources": list(bench.KERNEL_SOURCES[prec]),
                        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`; FETCH_SIZE (KiB) x 2 = the gfx950 correction of MI355X_MICROARCH.md"}
# fp32: per kernel
fe, wr = counters("f32_fetch", "FETCH_SIZE"), counters("f32_write", "WRITE_SIZE")
js = {"_meta": meta("fp32")}
for k in sorted(set(fe) | set(wr)):
    f = fe[k][0] * 1024 / max(fe[k][1], 1); w = wr[k][0] * 1024 / max(wr[k][1], 1)
    js[k] = {"launches": fe[k][1], "fetch_raw_bytes": f, "fetch_corrected_bytes": 2 * f, "write_bytes": w}
json.dump(js, open(f"{O}/hbm_traffic_latest.json", "w"), indent=1)
# bf16: family aggregate per mode
out = {"_meta": meta("bf16")}
for mode, tag in (("forward", "b16f"), ("train", "b16t")):
    fe, wr = counters(tag + "_fetch", "FETCH_SIZE"), counters(tag + "_write", "WRITE_SIZE")
    n = tf = tw = 0
    kern = {}
    for k in sorted(fe):
        if "gemm_bf16" not in k: continue
        f = 2 * fe[k][0] * 1024 / max(fe[k][1], 1); w = wr[k][0] * 1024 / max(wr[k][1], 1)
        kern[k] = {"launches_per_pass": fe[k][1], "fetch_corrected_bytes": f, "write_bytes": w}
        n += fe[k][1]; tf += f * fe[k][1]; tw += w * fe[k][1]
    out[mode] = {"launches_per_pass": n, "fetch_corrected_bytes_per_launch": tf / max(n, 1), "write_bytes_per_launch": tw / max(n, 1), "kernels": kern}
json.dump(out, open(f"{O}/hbm_traffic_bf16.json", "w"), indent=1)
# clock cross-check: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, per kernel, profiled passes
lines = [f"# r{rnd:02d} -- shader clock by PMC: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (rocprofv3 --pmc pass and --kernel-trace pass of the same command)", "",
         "The live figure in the bench line (`roofline.clock_mhz_under_load`) comes from a probe wave (s_memtime / s_memrealtime) inside an un-profiled forward;",
         "a PMC pass serialises kernels and adds its own overhead around each, so these read lower -- the guide's DVFS note: never compare a profiled arm with an un-profiled one.", ""]
for tag, title in (("f32", "fp32 forward"), ("b16f", "bf16 forward")):
    g, d = counters(tag + "_clk", "GRBM_GUI_ACTIVE"), durations(tag + "_trace")
    lines += [f"## {title}", "", "| kernel | launches | GRBM_GUI_ACTIVE / launch | duration us (kernel trace) | MHz |", "|---|---|---|---|---|"]
    for k in sorted(g, key=lambda k: -g[k][0]):
        if k not in d or d[k][1] == 0: continue
        cyc = g[k][0] / g[k][1] / 8; us = d[k][0] / d[k][1] / 1e3
        if us < 20: continue
        lines.append(f"| `{k}` | {g[k][1]} | {g[k][0] / g[k][1]:.4g} | {us:.1f} | {cyc / us:.0f} |")
    lines.append("")
open(f"{O}/clock_pmc.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $O/raw_*/
ls -la $O
