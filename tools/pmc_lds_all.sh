#!/bin/bash
# LDS bank-conflict share per kernel over one short run: tools/pmc_lds_all.sh [fwd_families args...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmcl -- python $R/tools/fwd_families.py "$@" --steps 1 > $O/pmcl.log 2>&1
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/pmcl/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:70]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))
for n, c in rows[:25]:
    a = c.get("SQ_LDS_IDX_ACTIVE", 0); b = c.get("SQ_LDS_BANK_CONFLICT", 0); w = c.get("SQ_WAVE_CYCLES", 1)
    print(f"{n:72s} lds_active {a:.3g}  conflict {b:.3g} ({100*b/max(a,1):.1f} %)  lds/wave_cycles {100*a/w:.1f} %")
PY
rm -rf $O/pmcl
