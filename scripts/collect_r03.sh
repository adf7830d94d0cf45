#!/bin/bash
# Everything profiles/ holds for round 3, from one GPU box and one tree (run on the GPU box; copy the files named at the
# end from gpurun_out/r03/ into profiles/).  Usage: bash scripts/collect_r03.sh
OUT=gpurun_out/r03; mkdir -p $OUT
python bench.py > $OUT/r03_bench.json 2> $OUT/bench.err || echo "bench failed"
bash scripts/prof_bench.sh $OUT/prof --no-cpu-baseline > $OUT/kernel_stats_top.txt
cp $OUT/prof/run_kernel_stats.csv $OUT/r03_kernel_stats.csv; cp $OUT/prof/bench.json $OUT/r03_bench_under_rocprof.json
python3 scripts/trace_window.py $OUT/prof $OUT/prof/bench.json > $OUT/r03_trace_window.json
# counters over one bench batch, every launch alone on the chip (separate passes: SQ x2, FETCH_SIZE, WRITE_SIZE)
bash scripts/pmc_r02.sh $OUT/pmc > $OUT/pmc_top.txt
cp $OUT/pmc/summary.json $OUT/r03_pmc_summary.json
python3 scripts/make_traffic.py $OUT/pmc/summary.json "k_runs<256, false, false>" 16384 > $OUT/traffic_r03.json
# attribution of the DTW launch's instructions: stream_debug masks (4 no side list, 1 no DP, 2 no staging) on the diagnostic instance
CFGS="stream_debug=128 stream_debug=132 stream_debug=133 stream_debug=135" bash scripts/pmc_debug_masks.sh > $OUT/r03_pmc_attribution.txt 2>&1
python scripts/stream_probe.py 16384 "" stream_debug=256 stream_debug=4 stream_debug=5 stream_debug=7 stream_blocks_per_cu=3 > $OUT/r03_stream_probe.txt 2>&1
python scripts/rounds_probe.py > $OUT/r03_rounds_probe.txt 2>&1
# configs[2]: the traceback kernels
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p $OUT/tb && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tb -o run -- python3 scripts/bench_modes.py --mode traceback --reads 8192 > $OUT/tb/out.json 2> $OUT/tb/err.txt )
cp $OUT/tb/run_kernel_stats.csv $OUT/r03_tb_kernel_stats.csv
mkdir -p $OUT/r03_modes
for m in "global_full 8192" "global_banded 8192" "traceback 8192"; do set -- $m
  python scripts/bench_modes.py --mode $1 --reads $2 > $OUT/r03_modes/$1_$2.json 2> $OUT/r03_modes/$1_$2.err || echo "mode $1 $2 failed"; done
bash scripts/sensitivity.sh $OUT/sens > $OUT/sens_top.txt; cp $OUT/sens/summary.json $OUT/r03_sensitivity.json
ls $OUT
