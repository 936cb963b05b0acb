#!/bin/bash
# PMC passes for the bf16 attention kernels inside a short training run (B = 8): waits, LDS, MFMA / VALU activity, instruction counts,
# and the shader clock (GRBM_GUI_ACTIVE / 8 / duration needs the kernel-trace pass).  -> gpurun_out/pmc_attn16.md
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp
export BATCH=${BATCH:-8}
CMD="python $R/tools/fwd_families.py --precision bf16 --mode train --batch $BATCH --steps 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmca_$i -- $CMD > $O/pmca_$i.log 2>&1 || echo "set $i failed: $(tail -1 $O/pmca_$i.log)"
  i=$((i+1))
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/pmca_trace -- $CMD > $O/pmca_trace.log 2>&1
cd $R
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
def kind(n): return "dkv" if "bwd_dkv" in n else "dq" if "bwd_dq" in n else "fwd"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{O}/pmca_[0-9]*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_bf16" in r["Kernel_Name"]:
            a = acc[kind(r["Kernel_Name"])][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{O}/pmca_trace/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_bf16" in r["Kernel_Name"]:
            d = dur[kind(r["Kernel_Name"])]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
import os
lines = [f"# bf16 attention kernels, training step at B = {os.environ.get('BATCH', '8')} x 246000 (T = 768, 12 heads): PMC per launch (one counter set per pass)", ""]
for k in ("fwd", "dq", "dkv"):
    c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
    us = dur[k][0] / max(dur[k][1], 1) / 1e3
    lines += [f"## {k}: {us:.1f} us per launch (kernel trace, un-profiled counters)", "", "| counter | per launch |", "|---|---|"]
    lines += [f"| {n} | {c[n]:.4g} |" for n in sorted(c)]
    d = []
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        d.append("wave-cycle split: waiting (s_waitcnt / barrier) %.2f, issue-stalled %.2f, issuing %.2f" % tuple(c.get(x, 0) / wc for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")))
        d.append("of the wave cycles: VALU issue %.2f, LDS issue %.2f, waiting on LDS %.2f" % tuple(c.get(x, 0) / wc for x in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS")))
    g = c.get("GRBM_GUI_ACTIVE")
    if g and us:
        d.append(f"shader clock in the profiled pass {g / 8 / us:.0f} MHz")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d.append(f"MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 256 * 4 / 8):.3f} of SIMD-cycles")
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
        d.append(f"instructions per launch: VALU {c['SQ_INSTS_VALU']:.4g} (incl. MFMA {c['SQ_INSTS_MFMA']:.4g}), LDS {c.get('SQ_INSTS_LDS', 0):.4g}, SALU {c.get('SQ_INSTS_SALU', 0):.4g}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d.append(f"LDS bank-conflict cycles / active cycles {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}")
    lines += [""] + ["* " + x for x in d] + [""]
open(f"{O}/pmc_attn16.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $O/pmca_[0-9]* $O/pmca_trace
