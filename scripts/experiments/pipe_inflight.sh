#!/bin/bash
OUT=gpurun_out/pipe_quick; mkdir -p $OUT
for n in "$@"; do
  RAWDTW_OPTS="$OPTS" timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 --inflight $n > $OUT/inf$n.json 2> $OUT/inf$n.err; python3 -c "
import json,sys; d=json.loads(open('$OUT/inf$n.json').read().strip().splitlines()[-1]); print('inflight $n', round(d['value'],1), round(d['ms_per_step'],4), 'pcie', round(d['value_pcie'],1), d['host_ms_per_step']['fetch'], d['host_ms_per_step']['submit'])"
done
