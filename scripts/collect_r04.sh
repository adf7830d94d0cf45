#!/bin/bash
# Everything profiles/ holds for round 4, from one GPU box and one tree (run on the GPU box; the files named r04_* and
# traffic_r04.json are then copied from gpurun_out/r04/ into profiles/).  Every step prints a line when it ends, and long
# steps write under gpurun_out/ as they go.  Usage: bash scripts/collect_r04.sh [first_step [last_step]]
OUT=gpurun_out/r04; mkdir -p $OUT
FIRST=${1:-1}; LAST=${2:-99}
step() { echo "[$(date +%T)] step $1: $2"; }
want() { [ $1 -ge $FIRST ] && [ $1 -le $LAST ]; }
if want 1; then
  timeout -k 10 1100 python -u bench.py > $OUT/r04_bench.json 2> $OUT/bench.err; step 1 "bench rc=$?"
fi
if want 2; then
  bash scripts/prof_bench.sh $OUT/prof --no-cpu-baseline > $OUT/kernel_stats_top.txt; step 2 "bench under rocprofv3 rc=$?"
  cp $OUT/prof/run_kernel_stats.csv $OUT/r04_kernel_stats.csv; cp $OUT/prof/bench.json $OUT/r04_bench_under_rocprof.json
  python3 scripts/trace_window.py $OUT/prof $OUT/prof/bench.json > $OUT/r04_trace_window.json; step 2 "trace window rc=$?"
fi
if want 3; then  # counters over one bench batch, every launch alone on the chip (separate passes: SQ x2, FETCH_SIZE, WRITE_SIZE)
  timeout -k 10 600 bash scripts/pmc_r02.sh $OUT/pmc 16384 > $OUT/pmc_top.txt 2>&1; step 3 "pmc passes rc=$?"
  cp $OUT/pmc/summary.json $OUT/r04_pmc_summary.json
  python3 scripts/make_traffic.py $OUT/pmc/summary.json "k_runs<256, false>" 16384 > $OUT/traffic_r04.json; step 3 "traffic rc=$?"
fi
if want 4; then  # attribution of k_runs' instructions: stream_debug masks on the diagnostic instance (1 no DP, 2 no staging)
  : > $OUT/r04_pmc_attribution.txt
  for cfg in stream_debug=128 stream_debug=129 stream_debug=131; do
    CFGS="$cfg" PMC_OUT=$OUT/pmc_masks timeout -k 10 300 bash scripts/pmc_debug_masks.sh >> $OUT/r04_pmc_attribution.txt 2>&1; step 4 "attribution $cfg rc=$?"
  done
fi
if want 5; then
  timeout -k 10 400 python -u scripts/stream_probe.py 16384 "" stream_debug=256 stream_debug=1 stream_debug=3 stream_debug=4 stream_blocks_per_cu=3 wide_beside=1 fold_mode=3 > $OUT/r04_stream_probe.txt 2>&1; step 5 "stream_probe rc=$?"
  timeout -k 10 400 python -u scripts/rounds_probe.py > $OUT/r04_rounds_probe.txt 2>&1; step 5 "rounds_probe rc=$?"
  timeout -k 10 200 python -u scripts/wreg_probe.py > $OUT/r04_wreg_probe.txt 2>&1; step 5 "wreg_probe rc=$?"
fi
if want 6; then  # configs[2]: the traceback kernels
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p $OUT/tb && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tb -o run -- python3 scripts/bench_modes.py --mode traceback --reads 8192 > $OUT/tb/out.json 2> $OUT/tb/err.txt ); step 6 "traceback profile rc=$?"
  cp $OUT/tb/run_kernel_stats.csv $OUT/r04_tb_kernel_stats.csv
  mkdir -p $OUT/r04_modes
  for m in "global_full 8192" "global_banded 8192" "traceback 8192"; do set -- $m
    timeout -k 10 400 python -u scripts/bench_modes.py --mode $1 --reads $2 > $OUT/r04_modes/$1_$2.json 2> $OUT/r04_modes/$1_$2.err; step 6 "mode $1 $2 rc=$?"; done
fi
if want 7; then  # how the batches in flight share the chip: kernel trace of the fresh-batch loop alone
  bash scripts/prof_bench.sh $OUT/ovl --no-cpu-baseline --modes-reads 0 --rounds 0 --trace-fresh 400 > /dev/null
  { tail -1 $OUT/ovl/bench.err; python3 scripts/experiments/trace_overlap.py $OUT/ovl/run_kernel_trace.csv k_scan; } > $OUT/r04_pipeline_overlap.txt 2>&1; step 7 "pipeline overlap rc=$?"
fi
if want 8; then
  bash scripts/sensitivity.sh $OUT/sens > $OUT/sens_top.txt; cp $OUT/sens/summary.json $OUT/r04_sensitivity.json; step 8 "sensitivity rc=$?"
fi
if want 9; then  # the chunk-round mapper: phases of a round, chaining on the host / on the device; its kernels by rocprofv3
  timeout -k 10 400 python -u scripts/mapper_probe.py 16384 1,16,0 1,16,0,dev 2,16,0,dev 1,1,0,dev 1,16,0,all 1,16,0,all,dev > $OUT/r04_mapper_probe.txt 2>&1; step 9 "mapper_probe rc=$?"
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p $OUT/mp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mp -o run -- python3 scripts/mapper_probe.py 16384 1,16,0,dev 1,16,0,all,dev > $OUT/mp/out.txt 2> $OUT/mp/err.txt ); step 9 "mapper profile rc=$?"
  cp $OUT/mp/run_kernel_stats.csv $OUT/r04_mapper_kernel_stats.csv
fi
ls $OUT
