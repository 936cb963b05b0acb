#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/c8; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "weight_grad or transposed" 2>&1 | tail -3
timeout 300 python bench.py --precision bf16 --mode train --no-cpu-baseline > $O/bench_bf16_train.json 2>/dev/null; python -c "
import json,sys
j=json.loads(open('$O/bench_bf16_train.json').read().strip().splitlines()[-1]); print('bf16 train', j['ms_per_step'], j['roofline']['achieved'], {k:v['ms_per_step'] for k,v in j['families'].items()})"
bash tools/prof_one.sh train_bf16_c8 --mode train --precision bf16 --steps 5 --warmup 2 > /dev/null 2>&1; head -8 gpurun_out/stats_train_bf16_c8.md
