#!/bin/bash
# A/B of one tuning knob of the tools-only library (lib/libw2v2_tuning.so), op level and end to end, arms interleaved on one box:
#   tools/ab_env.sh <KNOB> <value A> <value B> [bench args of the end-to-end arm ...]      -> gpurun_out/ab_<KNOB>.txt
# (bench.py and the tools load the tuning build through W2V2_NATIVE_LIB; the shipping library reads no environment.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
knob=$1; a=$2; b=$3; shift 3
export W2V2_NATIVE_LIB=$R/gsoc-wav2vec2_amd/lib/libw2v2_tuning.so
out=$O/ab_$knob.txt; : > $out
for v in $a $b; do
  echo "== $knob=$v: tools/gemm16_ab.py --time-only" >> $out
  env $knob=$v python $R/tools/gemm16_ab.py --time-only >> $out 2>&1
done
for rep in 1 2; do
  for v in $a $b; do
    for mode in "--precision bf16" "--precision bf16 --mode train"; do
      ms=$(env $knob=$v python $R/bench.py $mode --no-cpu-baseline --no-side --no-alt --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])")
      echo "$knob=$v rep $rep bench $mode: ms_per_step, family TF = $ms" >> $out
    done
  done
done
cat $out
