#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "" "fold_mode=3"; do  # kernel durations of one bench batch alone on the chip, by rocprofv3 trace
  OUT=gpurun_out/kernel_times/x$cfg; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 scripts/stream_probe.py 16384 "$cfg" > $OUT/log 2>&1
  echo "== $cfg"; grep -E "k_fold_select|k_chain_fold|k_read_select|k_scan|k_side|k_runs|k_plan|k_wide" $OUT/run_kernel_stats.csv | awk -F'","' '{print substr($1,2,60), $2, $4}'
done
