#!/bin/bash
# the whole GPU suite + the bf16 benches (round-3 checkpoints)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-suite}; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
for m in "" "--mode train"; do
timeout 300 python bench.py --precision bf16 $m --no-cpu-baseline > $O/bench_bf16$(echo $m | tr -d ' -').json 2>/dev/null; python -c "
import json,sys
j=json.loads(open('$O/bench_bf16$(echo $m | tr -d ' -').json').read().strip().splitlines()[-1]); print('bf16 $m', j['ms_per_step'], j['roofline']['achieved'], j.get('max_abs_logit_err'), {k:v['ms_per_step'] for k,v in j['families'].items()})"
done
timeout 400 python bench.py --mode train --precision bf16 --model large-robust --batch 16 --samples 480000 --no-cpu-baseline --steps 8 --warmup 3 > $O/bench_lr_train.json 2>/dev/null; python -c "
import json
j=json.loads(open('$O/bench_lr_train.json').read().strip().splitlines()[-1]); print('large-robust bf16 train', j['ms_per_step'], j['roofline']['achieved'])"
