#!/bin/bash
# The steps of scripts/collect_r03.sh a fault in the diagnostic instance cut off (stream_probe, attribution); one line a step.
OUT=gpurun_out/r03; mkdir -p $OUT
step() { echo "[$(date +%T)] $*"; }
timeout -k 10 400 python -u scripts/stream_probe.py 16384 "" stream_debug=256 stream_debug=4 stream_debug=5 stream_debug=7 stream_blocks_per_cu=3 > $OUT/r03_stream_probe.txt 2>&1; rc=$?; step "stream_probe rc=$rc"
[ $rc -eq 0 ] || exit 1
for cfg in stream_debug=128 stream_debug=132 stream_debug=133 stream_debug=135; do
  CFGS="$cfg" PMC_OUT=$OUT/pmc_masks timeout -k 10 300 bash scripts/pmc_debug_masks.sh >> $OUT/r03_pmc_attribution.txt 2>&1; rc=$?; step "attribution $cfg rc=$rc"
  [ $rc -eq 0 ] || exit 1
done
