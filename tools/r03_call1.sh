#!/bin/bash
# round 3, first GPU call: full GPU suite after the env-knob refactor, the new two-process trainer test, phase trace + PMC baseline
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/c1; mkdir -p $O; cd $R
(rocprofv3 --list-avail 2>/dev/null | grep -i -E "MFMA|TCC_HIT|TCC_MISS|TCC_REQ|TCC_EA0_R|TCC_EA0_W|FETCH_SIZE|WRITE_SIZE" | head -80) > $O/counters.txt 2>&1
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q > $O/dist_gpu.log 2>&1; tail -5 $O/dist_gpu.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_dist_gpu.py > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
timeout 300 python tools/gemm16_trace.py > $O/trace.log 2>&1; tail -60 $O/trace.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 1500 $O/bench_n1.json
timeout 300 python bench.py --precision bf16 --no-cpu-baseline > $O/bench_bf16_n1.json 2>/dev/null; tail -c 600 $O/bench_bf16_n1.json
timeout 300 python bench.py --mode train --precision bf16 --no-cpu-baseline > $O/bench_bf16_train_n1.json 2>/dev/null; tail -c 600 $O/bench_bf16_train_n1.json
bash tools/pmc_bench.sh fwd_bf16 gemm_bf16 --precision bf16 > $O/pmc_fwd.log 2>&1; tail -40 $O/pmc_fwd.log
cp gpurun_out/pmc_fwd_bf16.md $O/ 2>/dev/null
