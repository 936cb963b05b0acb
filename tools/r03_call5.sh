#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/c5; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_train_gpu.py -q -x -k "bf16" > $O/tests_bf16.log 2>&1; tail -4 $O/tests_bf16.log
for m in "--mode train"; do
timeout 300 python bench.py --precision bf16 $m --no-cpu-baseline > $O/bench_bf16$(echo $m | tr -d ' -').json 2>/dev/null; python -c "
import json,sys
j=json.loads(open('$O/bench_bf16$(echo $m | tr -d ' -').json').read().strip().splitlines()[-1]); print('bf16 $m', j['ms_per_step'], j['roofline']['achieved'], j.get('max_abs_logit_err'), {k:v['ms_per_step'] for k,v in j['families'].items()})"
done
timeout 400 python bench.py --mode train --precision bf16 --model large-robust --batch 16 --samples 480000 --no-cpu-baseline --steps 8 --warmup 3 > $O/bench_lr_train.json 2>/dev/null; python -c "
import json
j=json.loads(open('$O/bench_lr_train.json').read().strip().splitlines()[-1]); print('large-robust bf16 train', j['ms_per_step'], j['roofline']['achieved'])"
bash tools/prof_one.sh train_bf16_sw --mode train --precision bf16 --steps 5 --warmup 2 > /dev/null 2>&1; head -16 gpurun_out/stats_train_bf16_sw.md; grep -A14 "GEMM launches" gpurun_out/stats_train_bf16_sw.md
