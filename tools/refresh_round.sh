#!/bin/bash
# End-of-round evidence on the GPU box (through gpurun, from the repo root): the full -m gpu suite, smoke(), the bench lines of every
# reported configuration, and rocprofv3 kernel stats of the forward benches and the two fine-tune steps.  Everything lands under
# gpurun_out/final/; HEAD_REV (the commit of the snapshot) must be exported by the caller: the summaries carry it with the kernel-source hash (tools/prof_summary.py::provenance).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -6 $O/smoke.log
# the driver's own command (headline + the three BASELINE legs beside it), then the same line under the launcher with one rank (RCCL group of 1)
# every bench line is kept twice: the compact stdout line the driver parses (*.json) and the complete object (*_full.json, from gpurun_out/bench_full.json)
run() { local name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; cp gpurun_out/bench_full.json $O/${name}_full.json 2>/dev/null; }
run bench_n1 python bench.py --steps 20 --warmup 5
run bench_n1_launched python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline
S="--no-cpu-baseline --no-side"
run bench_bf16_n1 python bench.py --precision bf16 $S
run bench_bf16x3_n1 python bench.py --precision bf16x3 $S
run bench_f16x2_n1 python bench.py --precision f16x2 $S
run bench_bf16_train_n1 python bench.py --mode train --precision bf16 $S
run bench_train_n1 python bench.py --mode train $S
run bench_large_robust_bf16_train_n1 python bench.py --mode train --precision bf16 --model large-robust --batch 16 --samples 480000 $S --steps 8 --warmup 3
run bench_large_robust_bf16_n1 python bench.py --precision bf16 --model large-robust --batch 16 --samples 480000 $S --steps 8 --warmup 3
run bench_large_robust_fwd_n1 python bench.py --model large-robust --batch 16 --samples 246000 $S --steps 5 --warmup 2
for f in $O/bench_*_full.json; do echo "$(basename $f): $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"frac": [0-9.]*' $f | head -1)"; done
bash tools/prof_one.sh fwd_f32
bash tools/prof_one.sh fwd_bf16 --precision bf16
bash tools/prof_one.sh train_bf16 --mode train --precision bf16 --steps 5 --warmup 2
bash tools/prof_one.sh train_lr_bf16 --mode train --precision bf16 --model large-robust --batch 16 --samples 480000 --steps 3 --warmup 1
cp gpurun_out/stats_fwd_f32.md gpurun_out/stats_fwd_bf16.md gpurun_out/stats_train_bf16.md gpurun_out/stats_train_lr_bf16.md $O/ 2>/dev/null
