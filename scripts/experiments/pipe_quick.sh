#!/bin/bash
# the fresh-batch pipeline under option sets: bash scripts/experiments/pipe_quick.sh "name:opts" ...
OUT=gpurun_out/pipe_quick; mkdir -p $OUT
for spec in "$@"; do name=${spec%%:*}; opts=${spec#*:}
  RAWDTW_OPTS="$opts" timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 > $OUT/$name.json 2> $OUT/$name.err; python3 -c "
import json,sys; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],4), 'pcie', round(d['value_pcie'],1), d['launches']['alone_ms'], d['launches']['in_pipeline_ms'])"
done
