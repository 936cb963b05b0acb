#!/bin/bash
# Kernel time + PMC passes for the split (bf16x3) GEMM, one shape per process.  usage: tools/pmc_split.sh ffn1 [nopmc|pmc] [split|planes]
# (split = gemm_split.hip fed fp32 rows; planes = gemm_split_sw.hip fed pre-split planes)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp
shape=${1:-ffn1}; mode=${3:-split}
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_split -- python $R/tools/gemm_one.py $shape 6 $mode > $O/kt_split.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$O/kt_split/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_split" in r["Name"]: print("$shape $mode", r["Name"][:60], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "min us", float(r["MinNs"]) / 1e3)
PY
rm -rf $O/kt_split
[ "$2" = "nopmc" ] && exit 0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python $R/tools/gemm_one.py $shape 4 $mode > $O/pmc_$tag.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_$tag/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_split" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items(): print(f"{k:28s} {v / max(n, 1):.4g}  per launch ({n} launches)")
PY
  rm -rf $O/pmc_$tag
done
