#!/bin/bash
# Refresh the committed PMC figures bench.py quotes (roofline.traffic) and the PMC cross-check of the shader clock, on the GPU box:
#   tools/pmc_traffic.sh <round>      -> gpurun_out/pmc_traffic/{hbm_traffic_configs.json, clock_pmc.md}        (ONLY="x3 h2" limits the configurations)
# Counters are collected in their own passes (no tracing domains mixed in); FETCH_SIZE is in KiB and x 2 on gfx950 (MI355X_MICROARCH.md).
# Each JSON gets a _meta block: the round and the sha256 of the kernel sources the figures were measured on -- bench.py refuses a figure
# whose sources have changed since (measured_traffic).  Copy the files into profiles/ afterwards.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_traffic; mkdir -p $O; round=${1:-0}; ONLY=${ONLY:-}
cd /tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-side"
run() { local tag=$1; shift; timeout 500 rocprofv3 "$@" --output-format csv -d $O/raw_$tag -- $B ${EXTRA:-} > $O/raw_$tag.log 2>&1 || echo "pass $tag failed: $(tail -2 $O/raw_$tag.log)"; }
# tag | bench arguments   (every BASELINE configuration and the two split modes of the forward legs: profiles/hbm_traffic_configs.json)
CONFIGS="f32|
b16f|--precision bf16
b16t|--precision bf16 --mode train
x3|--precision bf16x3
h2|--precision f16x2
Lf32|--model large-robust --batch 16
Lx3|--model large-robust --batch 16 --precision bf16x3
Lh2|--model large-robust --batch 16 --precision f16x2
Lb16t|--model large-robust --batch 16 --samples 480000 --precision bf16 --mode train"
while IFS='|' read -r tag extra; do
  [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $tag " && continue
  EXTRA="$extra" run ${tag}_fetch --pmc FETCH_SIZE
  EXTRA="$extra" run ${tag}_write --pmc WRITE_SIZE
done <<< "$CONFIGS"
if [ -z "$ONLY" ]; then
  EXTRA="" run f32_clk --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
  EXTRA="" run f32_trace --kernel-trace
  EXTRA="--precision bf16" run b16f_clk --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
  EXTRA="--precision bf16" run b16f_trace --kernel-trace
fi
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
STEPS = 3      # --steps 2 --warmup 1: every pass runs three identical steps
CONF = [("f32", "base", "fp32", "forward", 246000), ("b16f", "base", "bf16", "forward", 246000), ("b16t", "base", "bf16", "train", 246000),
        ("x3", "base", "bf16x3", "forward", 246000), ("h2", "base", "f16x2", "forward", 246000),
        ("Lf32", "large-robust", "fp32", "forward", 246000), ("Lx3", "large-robust", "bf16x3", "forward", 246000),
        ("Lh2", "large-robust", "f16x2", "forward", 246000), ("Lb16t", "large-robust", "bf16", "train", 480000)]
path = os.path.join(ROOT, "profiles", "hbm_traffic_configs.json")
try:
    out = json.load(open(path))
except (OSError, ValueError):
    out = {}
out["_note"] = ("HBM traffic of the dominant GEMM family per BASELINE configuration: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of "
                "`bench.py --steps 2 --warmup 1 --no-side --no-alt <args>`; FETCH_SIZE (KiB) x 2 = the gfx950 correction of MI355X_MICROARCH.md; "
                "kernel_launches_per_step = the family's kernels in the pass / 3 steps.  Each entry carries the hash of the kernel sources it was measured on.")
for tag, model, prec, mode, L in CONF:
    fe, wr = counters(tag + "_fetch", "FETCH_SIZE"), counters(tag + "_write", "WRITE_SIZE")
    fam = bench.FAMILY_KERNEL_TAG[prec]
    n = tf = tw = 0
    kern = {}
    for k in sorted(fe):
        if fam not in k or "split_weight" in k or "split_planes" in k: continue
        f = 2 * fe[k][0] * 1024 / max(fe[k][1], 1); w = wr[k][0] * 1024 / max(wr[k][1], 1) if k in wr else 0.0
        kern[k] = {"launches_per_pass": fe[k][1], "fetch_corrected_bytes": f, "write_bytes": w}
        n += fe[k][1]; tf += f * fe[k][1]; tw += w * fe[k][1]
    if not n: continue
    out[bench.traffic_key(model, prec, mode, L)] = {
        "round": rnd, "kernel_source_sha16": bench.kernel_source_hash(prec), "kernel_sources": list(bench.KERNEL_SOURCES[prec]),
        "kernel_launches_per_step": n // STEPS, "fetch_corrected_bytes_per_launch": tf / n, "write_bytes_per_launch": tw / n, "kernels": kern}
json.dump(out, open(f"{O}/hbm_traffic_configs.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in out.items() if isinstance(v, dict)}, indent=1))
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
