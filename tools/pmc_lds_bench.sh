#!/bin/bash
# LDS bank-conflict share per kernel over one short bench.py run (all kernels, not one family): tools/pmc_lds_bench.sh <bench args...>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmcl -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-profile --no-alt > $O/pmcl.log 2>&1
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/pmcl/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:70]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))
print("| kernel | LDS active cycles | of them bank conflicts | LDS active / wave cycles |\n|---|---|---|---|")
for n, c in rows[:30]:
    a = c.get("SQ_LDS_IDX_ACTIVE", 0); b = c.get("SQ_LDS_BANK_CONFLICT", 0); w = c.get("SQ_WAVE_CYCLES", 1)
    print(f"| \`{n}\` | {a:.3g} | {b:.3g} ({100*b/max(a,1):.1f} %) | {100*a/w:.1f} % |")
PY
rm -rf $O/pmcl
