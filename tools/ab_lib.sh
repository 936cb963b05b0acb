#!/bin/bash
# End-to-end A/B of two builds of the library on one box, arms interleaved:  tools/ab_lib.sh <tag> <libA.so> <libB.so> <reps> <bench args ...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
tag=$1; A=$2; B=$3; reps=$4; shift 4
out=$O/abl_$tag.txt; echo "# A = $A, B = $B, bench.py $*" > $out
for rep in $(seq 1 $reps); do
  for arm in A B; do
    lib=$A; [ $arm = B ] && lib=$B
    line=$(W2V2_NATIVE_LIB=$R/$lib python $R/bench.py "$@" --no-cpu-baseline --no-side --no-alt 2>/dev/null >/dev/null; python -c "
import sys,json
d=json.load(open('$R/gpurun_out/bench_full.json')); f=d.get('families',{})      # (stdout carries the compact line only: the complete object is in the file)
print(d['ms_per_step'], d['roofline']['achieved'], ' '.join(f'{k}={v[\"ms_per_step\"]}' for k,v in f.items()))")
    echo "$arm rep $rep: $line" >> $out
  done
done
cat $out
