set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/sweep
T="--precision bf16 --mode train --steps 10 --warmup 3"
L="--precision bf16 --mode train --model large-robust --batch 16 --samples 480000 --steps 5 --warmup 2"
for spec in "W2V2_DW_BLOCKS|384 512 768" "W2V2_DBC_BLOCKS|4096 8192 16384" "W2V2_EW_BLOCKS|8192 16384 32768" "W2V2_LN_BLOCKS|512 1024 2048" "W2V2_DW_UNEVEN|0 1"; do
  k=${spec%%|*}; v=${spec#*|}
  bash tools/ab_bench.sh $k "$v" 2 $T > /dev/null; cut -c1-30,1-40 gpurun_out/abb_$k.txt | awk '{print $1,$2,$3,$4}' > gpurun_out/sweep/base_$k.txt
  bash tools/ab_bench.sh $k "$v" 1 $L > /dev/null; awk '{print $1,$2,$3,$4}' gpurun_out/abb_$k.txt > gpurun_out/sweep/large_$k.txt
done
tail -n +1 gpurun_out/sweep/*.txt
