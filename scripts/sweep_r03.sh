#!/bin/bash
# bench.py under a few RAWDTW_OPTS / --inflight settings (GPU box).  Usage: bash scripts/sweep_r03.sh "opts1" "opts2" ...
for cfg in "$@"; do
  opts=${cfg%%|*}; extra=${cfg#*|}; [ "$extra" = "$cfg" ] && extra=""
  RAWDTW_OPTS="$opts" python bench.py --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-60s value %.1f  ms/step %.4f  alone %s  pcie %.1f  replay %.1f' % ('$cfg', d['value'], d['ms_per_step'], d["launches"]["alone_ms"], d['pipeline_pcie']['gcups'], d['kernel_replay']['gcups']))"
done
