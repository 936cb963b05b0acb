#!/bin/bash
# PMC passes over one bench.py invocation, one counter set per process (no tracing domains mixed in), summarised per kernel:
#   tools/pmc_bench.sh <tag> <kernel-substring> <bench args...>     -> gpurun_out/pmc_<tag>.md
# Counter sets: MFMA busy / issue, wave stalls, LDS, L2 hit / miss, HBM fetch / write.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; tag=$1; filt=$2; shift; shift
mkdir -p $O; cd /tmp
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE")
i=0
for set in "${SETS[@]}"; do
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/pmcb_${tag}_$i -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-side > $O/pmcb_${tag}_$i.log 2>&1 || echo "set $i ($set) failed: $(tail -2 $O/pmcb_${tag}_$i.log)"
  i=$((i+1))
done
cd $R
python - "$O" "$tag" "$filt" <<'PY'
import csv, glob, sys, collections, re
O, tag, filt = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{O}/pmcb_{tag}_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0]
        if filt in k:
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
lines = []
for k in sorted(acc):
    c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
    n = max(v[1] for v in acc[k].values())
    lines.append(f"### `{k}` ({n} launches per pass)\n")
    lines.append("| counter | per launch |\n|---|---|")
    for name in sorted(c):
        lines.append(f"| {name} | {c[name]:.4g} |")
    g = c.get("GRBM_GUI_ACTIVE")
    d = []
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d.append(f"MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD-normalised) : {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 256 * 4 / 8):.3f} of SIMD-cycles (GRBM_GUI_ACTIVE is summed over 8 XCDs)")
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d.append(f"L2 hit rate {c['TCC_HIT_sum'] / max(1.0, c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
    if "FETCH_SIZE" in c:
        d.append(f"HBM/fabric fetch {2 * c['FETCH_SIZE'] * 1024 / 1e6:.1f} MB per launch (FETCH_SIZE KiB x 2: gfx950 correction)")
    if "WRITE_SIZE" in c:
        d.append(f"write {c['WRITE_SIZE'] * 1024 / 1e6:.1f} MB per launch")
    if "SQ_WAVE_CYCLES" in c:
        d.append("wave-cycle split: waiting (s_waitcnt / barrier) %.2f, issue-stalled %.2f, issuing %.2f" % tuple(c.get(x, 0) / c["SQ_WAVE_CYCLES"] for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")))
    lines += [""] + ["* " + x for x in d] + [""]
# family aggregate: launch-weighted mean over the matching kernels (what bench.py's roofline.traffic quotes)
import json
tot = {"launches": 0, "fetch": 0.0, "write": 0.0, "kernels": {}}
for k in sorted(acc):
    c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
    n = max(v[1] for v in acc[k].values())
    f, w = 2 * c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    tot["kernels"][k] = {"launches_per_pass": n, "fetch_corrected_bytes": f, "write_bytes": w}
    tot["launches"] += n; tot["fetch"] += f * n; tot["write"] += w * n
if tot["launches"]:
    js = {"tag": tag, "launches_per_pass": tot["launches"], "fetch_corrected_bytes_per_launch": tot["fetch"] / tot["launches"],
          "write_bytes_per_launch": tot["write"] / tot["launches"], "kernels": tot["kernels"],
          "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE (KiB) x 2 = the gfx950 correction of MI355X_MICROARCH.md"}
    json.dump(js, open(f"{O}/pmc_{tag}.json", "w"), indent=1)
    lines.append(f"Family aggregate: {tot['launches']} launches per pass, {js['fetch_corrected_bytes_per_launch'] / 1e6:.1f} MB fetched (corrected) + "
                 f"{js['write_bytes_per_launch'] / 1e6:.1f} MB written per launch")
open(f"{O}/pmc_{tag}.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $O/pmcb_${tag}_*/
