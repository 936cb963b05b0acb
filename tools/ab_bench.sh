#!/bin/bash
# End-to-end A/B of one tuning knob of the tools-only library, arms interleaved on one box:
#   tools/ab_bench.sh <KNOB> "<v1> <v2> ..." <reps> <bench args ...>      -> gpurun_out/abb_<KNOB>.txt  (ms_per_step, family TF, family ms)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
knob=$1; vals=$2; reps=$3; shift 3
export W2V2_NATIVE_LIB=$R/gsoc-wav2vec2_amd/lib/libw2v2_tuning.so
out=$O/abb_$knob.txt; echo "# $knob in {$vals}, bench.py $* (tools-only library)" > $out
for rep in $(seq 1 $reps); do
  for v in $vals; do
    line=$(env $knob=$v python $R/bench.py "$@" --no-cpu-baseline --no-side --no-alt 2>/dev/null >/dev/null; python -c "
import sys,json
d=json.load(open('$R/gpurun_out/bench_full.json')); f=d.get('families',{})      # (stdout carries the compact line only: the complete object is in the file)
print(d['ms_per_step'], d['roofline']['achieved'], ' '.join(f'{k}={v[\"ms_per_step\"]}' for k,v in f.items()))")
    echo "$knob=$v rep $rep: $line" >> $out
  done
done
cat $out
