#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash scratch/collect_profiles.sh r02
# kernel-trace stats (single stream = isolated kernels; default = overlapped streams), HBM traffic of gemm_f16_kernel from
# two separate --pmc passes, the b = 32 breakdown, pre-train and ViT-B/16 stats, SQ counters of two contrasting GEMMs.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --unfolded-steps 0 "$@" > $OUT/$name.log 2>&1
  cp $(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/$name.csv
  echo "$name: $(tail -1 $OUT/$name.log | cut -c1-160)"
}
run_stats ${TAG}_bench_b256_kernel_stats_single_stream --steps 4 --warmup 2 --roofline-steps 2 --single-stream --vit-forward-iters 0
run_stats ${TAG}_bench_b256_kernel_stats --steps 4 --warmup 2 --roofline-steps 1 --vit-forward-iters 0
run_stats ${TAG}_bench_pretrain_b128_kernel_stats_single_stream --mode pretrain --steps 4 --warmup 2 --roofline-steps 1 --single-stream --vit-forward-iters 0
run_stats ${TAG}_bench_vitb16_b16_f24_kernel_stats_single_stream --clip ViT-B/16 --frames 24 --batch 16 --steps 4 --warmup 2 --roofline-steps 1 --single-stream --vit-forward-iters 0
# the fp32 regime (model.float(): the regime whose logits are held to 1e-3), round 5
run_stats ${TAG}_bench_fp32_b256_kernel_stats_single_stream --regime fp32 --steps 2 --warmup 1 --roofline-steps 1 --single-stream
# b = 32 with the per-step breakdown (kernel trace)
rm -rf /tmp/prof_b32
rocprofv3 --kernel-trace --stats -d /tmp/prof_b32 -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --batch 32 --steps 6 --warmup 2 --roofline-steps 0 --single-stream --vit-forward-iters 0 --unfolded-steps 0 --reserve-cus 16 > $OUT/${TAG}_b32.log 2>&1
cp $(find /tmp/prof_b32 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_b32_kernel_stats_single_stream.csv
python3 $R/scratch/trace_gaps.py $(find /tmp/prof_b32 -name "*kernel_trace.csv" | head -1) 45 > $OUT/${TAG}_bench_b32_kernel_breakdown_single_stream.txt 2>&1
rm -rf /tmp/prof_b32o
rocprofv3 --kernel-trace --stats -d /tmp/prof_b32o -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-hbm-roofline --batch 32 --steps 6 --warmup 2 --roofline-steps 0 --vit-forward-iters 0 --unfolded-steps 0 --reserve-cus 16 > $OUT/${TAG}_b32o.log 2>&1
python3 $R/scratch/trace_gaps.py $(find /tmp/prof_b32o -name "*kernel_trace.csv" | head -1) 45 > $OUT/${TAG}_bench_b32_kernel_breakdown.txt 2>&1
# HBM traffic of gemm_f16_kernel: separate passes per counter, kernel trace only
CMD="--steps 2 --warmup 1 --no-cpu-baseline --no-hbm-roofline --roofline-steps 0 --vit-forward-iters 0 --unfolded-steps 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python3 $R/bench.py $CMD > $OUT/pmc_$c.log 2>&1
done
python3 $R/scratch/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
  $OUT/${TAG}_gemm_f16_hbm_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py $CMD" 3 > $OUT/traffic.log 2>&1   # 1 warm-up + 2 timed steps; nothing runs after the timed region with --roofline-steps 0
# SQ counters of the c_fc forward and the in_proj weight-gradient shapes
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/sq$i -o p --output-format csv -- python3 $R/scratch/gemm_pmc.py > $OUT/sq_run$i.log 2>&1 || echo "sq pass $i failed"
done
python3 $R/scratch/pmc_summary.py $(find /tmp/sq1 /tmp/sq2 /tmp/sq3 -name "*counter_collection.csv") > $OUT/${TAG}_gemm_f16_sq_counters.txt 2>&1
ls -la $OUT
# eval leg (fused scorer): kernel stats of bench.py --mode eval
cd /tmp
rm -rf /tmp/prof_eval
rocprofv3 --kernel-trace --stats -d /tmp/prof_eval -o t --output-format csv -- python3 $R/bench.py --mode eval --frames 24 --no-cpu-baseline > $OUT/${TAG}_bench_eval_f24_kernel_stats.log 2>&1
cp $(find /tmp/prof_eval -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_eval_f24_kernel_stats.csv
