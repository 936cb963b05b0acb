#!/bin/bash
# rocprofv3 kernel stats of one bench.py invocation: tools/prof_one.sh <tag> <bench args...>  -> gpurun_out/stats_<tag>.md
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; tag=$1; shift; cd /tmp
export PROF_CMD="rocprofv3 --kernel-trace --stats -- python bench.py $* --no-cpu-baseline --no-profile --no-alt --no-side"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python $R/bench.py "$@" --no-cpu-baseline --no-profile --no-alt --no-side > $O/prof_$tag.log 2>&1
cd $R; python tools/prof_summary.py stats gpurun_out/prof_$tag gpurun_out/stats_$tag.md
# GAPS=<steps in the trace>: also the device-side idle time between consecutive kernels (gpurun_out/gaps_<tag>.md)
[ -n "${GAPS:-}" ] && python tools/prof_summary.py gaps gpurun_out/prof_$tag gpurun_out/gaps_$tag.md $GAPS
rm -rf gpurun_out/prof_$tag
