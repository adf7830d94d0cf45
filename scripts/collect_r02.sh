#!/bin/bash
# Everything profiles/ holds for round 2, from one GPU box and one tree (run on the GPU box; copy the files named at the
# end from gpurun_out/r02/ into profiles/).  Usage: bash scripts/collect_r02.sh
OUT=gpurun_out/r02; mkdir -p $OUT
python bench.py > $OUT/r02_bench.json 2> $OUT/bench.err || echo "bench failed"
bash scripts/prof_bench.sh $OUT/prof --no-cpu-baseline > $OUT/kernel_stats_top.txt
cp $OUT/prof/run_kernel_stats.csv $OUT/r02_kernel_stats.csv; cp $OUT/prof/bench.json $OUT/r02_bench_under_rocprof.json
python3 scripts/trace_window.py $OUT/prof $OUT/prof/bench.json > $OUT/r02_trace_window.json
bash scripts/pmc_r02.sh $OUT/pmc > $OUT/pmc_top.txt
cp $OUT/pmc/summary.json $OUT/r02_pmc_summary.json
python3 scripts/make_traffic.py $OUT/pmc/summary.json "k_stream<256>" 16384 > $OUT/traffic_r02.json
# attribution of the DTW launch's instructions: stream_debug masks (4 no side list, 1 no DP, 2 no staging, 16 no marks)
CFGS="stream_debug=0 stream_debug=4 stream_debug=5 stream_debug=7 stream_debug=23" bash scripts/pmc_debug_masks.sh > $OUT/r02_pmc_attribution.txt 2>&1
python scripts/stream_probe.py 16384 "" stream_debug=4 stream_debug=32 stream_debug=96 stream_tile_radius=2 stream_blocks_per_cu=5,tile_lds_floats=4800 > $OUT/r02_stream_probe.txt 2>&1
python scripts/wreg_probe.py > $OUT/r02_wreg_probe.txt 2>&1
bash scripts/sensitivity.sh $OUT/sens > $OUT/sens_top.txt; cp $OUT/sens/summary.json $OUT/r02_sensitivity.json
mkdir -p $OUT/r02_modes
for m in "global_full 1024" "global_full 8192" "global_banded 8192" "traceback 1024" "traceback 8192"; do set -- $m
  python scripts/bench_modes.py --mode $1 --reads $2 > $OUT/r02_modes/$1_$2.json 2> $OUT/r02_modes/$1_$2.err || echo "mode $1 $2 failed"; done
python scripts/host_costs.py 16384 > $OUT/r02_host_costs.json 2> $OUT/host_costs.err || echo "host_costs failed"
ls $OUT
