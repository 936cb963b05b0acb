#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/c3; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -x -k "bf16 or gemm or golden" > $O/tests_bf16.log 2>&1; tail -4 $O/tests_bf16.log
timeout 900 python -m pytest tests/test_train_gpu.py -q -x -k "bf16" > $O/tests_train_bf16.log 2>&1; tail -4 $O/tests_train_bf16.log
timeout 300 python bench.py --precision bf16 --no-cpu-baseline > $O/bench_bf16_n1.json 2>/dev/null; python -c "
import json,sys
j=json.loads(open('$O/bench_bf16_n1.json').read().strip().splitlines()[-1]); print('bf16 fwd', j['ms_per_step'], j['roofline']['achieved'], j['max_abs_logit_err'], {k:v['ms_per_step'] for k,v in j['families'].items()})"
timeout 300 python bench.py --mode train --precision bf16 --no-cpu-baseline > $O/bench_bf16_train_n1.json 2>/dev/null; python -c "
import json,sys
j=json.loads(open('$O/bench_bf16_train_n1.json').read().strip().splitlines()[-1]); print('bf16 train', j['ms_per_step'], j['roofline']['achieved'], {k:v['ms_per_step'] for k,v in j['families'].items()})"
