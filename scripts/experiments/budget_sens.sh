#!/bin/bash
# image budget against the workload: bash scripts/experiments/budget_sens.sh "7000 5800" 
OUT=gpurun_out/budget_sens; mkdir -p $OUT
for b in $1; do for w in "--hit-prob 0.10" "--hit-prob 0.40" "--decoy-gap-median 48" "--decoy-gap-median 200" "--decoys-per-read 4"; do
  n=$(echo "$b $w" | tr -d ' -.'); RAWDTW_OPTS="tile_lds_floats=$b" timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 $w > $OUT/$n.json 2> $OUT/$n.err
  python3 -c "
import json; d=json.loads(open('$OUT/$n.json').read().strip().splitlines()[-1]); print('$b', '$w', round(d['value'],1), d['launches']['alone_ms'])"
done; done
