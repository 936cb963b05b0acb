#!/bin/bash
# End-of-round evidence on the GPU box (through gpurun, from the repo root): the full -m gpu suite, smoke(), the bench lines of every
# reported configuration, and rocprofv3 kernel stats of the forward benches and the two fine-tune steps.  Everything lands under
# gpurun_out/final/; tools/collect_profiles.py copies the summaries into profiles/ with the round's prefix.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -6 $O/smoke.log
# the driver's own command (headline + the three BASELINE legs beside it), then the same line under the launcher with one rank (RCCL group of 1)
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n1_launched.json 2> $O/bench_n1_launched.err
S="--no-cpu-baseline --no-side"
python bench.py --precision bf16 $S > $O/bench_bf16_n1.json 2>/dev/null
python bench.py --precision bf16x3 $S > $O/bench_bf16x3_n1.json 2>/dev/null
python bench.py --precision f16x2 $S > $O/bench_f16x2_n1.json 2>/dev/null
python bench.py --mode train --precision bf16 $S > $O/bench_bf16_train_n1.json 2>/dev/null
python bench.py --mode train $S > $O/bench_train_n1.json 2>/dev/null
python bench.py --mode train --precision bf16 --model large-robust --batch 16 --samples 480000 $S --steps 8 --warmup 3 > $O/bench_large_robust_bf16_train_n1.json 2>/dev/null
python bench.py --precision bf16 --model large-robust --batch 16 --samples 480000 $S --steps 8 --warmup 3 > $O/bench_large_robust_bf16_n1.json 2>/dev/null
python bench.py --model large-robust --batch 16 --samples 246000 $S --steps 5 --warmup 2 > $O/bench_large_robust_fwd_n1.json 2>/dev/null
for f in $O/bench_*.json; do echo "$(basename $f): $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"frac": [0-9.]*' $f | head -1)"; done
bash tools/prof_one.sh fwd_f32
bash tools/prof_one.sh fwd_bf16 --precision bf16
bash tools/prof_one.sh train_bf16 --mode train --precision bf16 --steps 5 --warmup 2
bash tools/prof_one.sh train_lr_bf16 --mode train --precision bf16 --model large-robust --batch 16 --samples 480000 --steps 3 --warmup 1
cp gpurun_out/stats_fwd_f32.md gpurun_out/stats_fwd_bf16.md gpurun_out/stats_train_bf16.md gpurun_out/stats_train_lr_bf16.md $O/ 2>/dev/null
