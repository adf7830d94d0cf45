#!/bin/bash
# What each launch costs the fresh-batch pipeline: the bench loop with launches left out (results are then wrong).
OUT=gpurun_out/pipe_parts; mkdir -p $OUT
run() { name=$1; shift; RAWDTW_OPTS="$1" timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 > $OUT/$name.json 2> $OUT/$name.err; python3 -c "
import json,sys; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],4), d['launches']['alone_ms'], d['launches']['in_pipeline_ms'])"; }
run default ""
run nofold "debug_skip_tail=1"
run notail "debug_skip_tail=3"
run nodp "stream_debug=1"
run nostage "stream_debug=3"
run nowide "stream_debug=4"
