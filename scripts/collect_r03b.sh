#!/bin/bash
# Second half of scripts/collect_r03.sh (the part a silent stretch cut off): probes, configs[2] kernels, modes.
# Every step prints a line when it ends so that the box sees progress.  Usage: bash scripts/collect_r03b.sh
OUT=gpurun_out/r03; mkdir -p $OUT
step() { echo "[$(date +%T)] $*"; }
timeout -k 10 400 python -u scripts/stream_probe.py 16384 "" stream_debug=256 stream_debug=4 stream_debug=5 stream_debug=7 stream_blocks_per_cu=3 > $OUT/r03_stream_probe.txt 2>&1; step "stream_probe rc=$?"
timeout -k 10 400 python -u scripts/rounds_probe.py > $OUT/r03_rounds_probe.txt 2>&1; step "rounds_probe rc=$?"
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p $OUT/tb && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tb -o run -- python3 scripts/bench_modes.py --mode traceback --reads 8192 > $OUT/tb/out.json 2> $OUT/tb/err.txt ); step "tb profile rc=$?"
cp $OUT/tb/run_kernel_stats.csv $OUT/r03_tb_kernel_stats.csv
mkdir -p $OUT/r03_modes
for m in "global_full 8192" "global_banded 8192" "traceback 8192"; do set -- $m
  timeout -k 10 400 python -u scripts/bench_modes.py --mode $1 --reads $2 > $OUT/r03_modes/$1_$2.json 2> $OUT/r03_modes/$1_$2.err; step "mode $1 $2 rc=$?"; done
ls $OUT
