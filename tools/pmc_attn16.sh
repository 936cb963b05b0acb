#!/bin/bash
# PMC passes for the bf16 attention kernels inside a short training run: LDS conflicts / activity, waits, MFMA / VALU busy.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmca_$tag -- python $R/tools/fwd_families.py --precision bf16 --mode train --batch 8 --steps 1 > $O/pmca_$tag.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmca_$tag/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attention_bf16" in n:
            k = ("dkv" if "dkv" in n else "dq" if "bwd_dq" in n else "fwd") + " " + r["Counter_Name"]
            a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc): v, n = acc[k]; print(f"{k:40s} {v / max(n, 1):.4g}  per launch ({n})")
PY
  rm -rf $O/pmca_$tag
done
