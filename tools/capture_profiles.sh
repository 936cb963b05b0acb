#!/bin/bash
# Capture the rocprofv3 evidence of the bench commands on the GPU box (run through gpurun from the repo root):
#   kernel-trace + stats for forward/train in both precisions, and separate FETCH_SIZE / WRITE_SIZE PMC passes
#   (no tracing domains mixed in) for the headline forward.  Output under gpurun_out/prof_*.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt --no-side"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd_f32   -- $B                                > $O/prof_fwd_f32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd_bf16  -- $B --precision bf16               > $O/prof_fwd_bf16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd_bf16x3 -- $B --precision bf16x3            > $O/prof_fwd_bf16x3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_f32 -- $B --mode train                   > $O/prof_train_f32.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_bf16 -- $B --mode train --precision bf16 > $O/prof_train_bf16.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $B > $O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $B > $O/prof_write.log 2>&1
cd $R
for n in fwd_f32 fwd_bf16 fwd_bf16x3 train_f32 train_bf16; do
  python tools/prof_summary.py stats gpurun_out/prof_$n gpurun_out/stats_$n.md || echo "summary $n failed"
done
python tools/prof_summary.py pmc gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/hbm_traffic.md || echo "pmc summary failed"
# keep the merged-back payload small: drop the raw traces, keep the summaries
rm -rf gpurun_out/prof_fwd_f32 gpurun_out/prof_fwd_bf16 gpurun_out/prof_fwd_bf16x3 gpurun_out/prof_train_f32 gpurun_out/prof_train_bf16 gpurun_out/prof_fetch gpurun_out/prof_write
tail -2 gpurun_out/prof_fwd_f32.log
