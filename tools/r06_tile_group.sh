set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -4
bash tools/ab_bench.sh W2V2_TILE_GROUP "0 -1" 2 --model large-robust --batch 16 --precision f16x2 --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/abb_W2V2_TILE_GROUP.txt gpurun_out/abb_TG_Lh2.txt
bash tools/ab_bench.sh W2V2_TILE_GROUP "0 -1" 2 --model large-robust --batch 16 --samples 480000 --precision bf16 --mode train --steps 6 --warmup 2 > /dev/null 2>&1; cp gpurun_out/abb_W2V2_TILE_GROUP.txt gpurun_out/abb_TG_Lb16t.txt
bash tools/ab_bench.sh W2V2_TILE_GROUP "0 -1" 2 --precision bf16 --mode train --steps 10 --warmup 3 > /dev/null 2>&1; cp gpurun_out/abb_W2V2_TILE_GROUP.txt gpurun_out/abb_TG_b16t.txt
bash tools/ab_bench.sh W2V2_TILE_GROUP "0 -1" 2 --model large-robust --batch 16 --steps 6 --warmup 2 > /dev/null 2>&1; cp gpurun_out/abb_W2V2_TILE_GROUP.txt gpurun_out/abb_TG_Lf32.txt
cat gpurun_out/abb_TG_*.txt | cut -c1-200
ONLY="Lf32 Lh2 b16t Lb16t" bash tools/pmc_traffic.sh 6 2>&1 | tail -15
