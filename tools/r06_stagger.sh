# round 6: A/B of the bf16 GEMM phase stagger (tools-only knob W2V2_SW_STAGGER_PCT: % of half... of a tile period the second block of a CU waits)
set -u
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh W2V2_SW_STAGGER_PCT "0 50 25 75" 2 --precision bf16 --mode train --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/abb_W2V2_SW_STAGGER_PCT.txt gpurun_out/abb_STAG_b16t.txt
bash tools/ab_bench.sh W2V2_SW_STAGGER_PCT "0 50 25 75" 2 --model large-robust --batch 16 --samples 480000 --precision bf16 --mode train --steps 6 --warmup 2 > /dev/null 2>&1; cp gpurun_out/abb_W2V2_SW_STAGGER_PCT.txt gpurun_out/abb_STAG_Lb16t.txt
bash tools/ab_bench.sh W2V2_SW_STAGGER_PCT "0 50 25 75" 2 --precision bf16 --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/abb_W2V2_SW_STAGGER_PCT.txt gpurun_out/abb_STAG_b16f.txt
cat gpurun_out/abb_STAG_*.txt | cut -c1-150
