#!/bin/bash
# PMC passes for the bf16-operand GEMM (one shape per process): LDS conflicts / activity, waits, MFMA busy.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python $R/tools/gemm_one.py ${1:-ffn1} 5 bf16 > $O/pmc_$tag.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_$tag/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items(): print(f"{k:28s} {v / max(n, 1):.4g}  per launch ({n} launches)")
PY
  rm -rf $O/pmc_$tag
done
