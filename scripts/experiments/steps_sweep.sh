#!/bin/bash
# the fresh-batch loop's region time against the number of steps (start-up and drain of the four-deep pipeline against its steady state)
for k in 20 40 80 160 320; do RAWDTW_BENCH_DUMP=1 timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 --steps $k > gpurun_out/st$k.json 2> gpurun_out/st$k.err; python3 -c "
import json; d=json.loads(open('gpurun_out/st$k.json').read().strip().splitlines()[-1]); print('steps $k', round(d['value'],1), round(d['ms_per_step'],4), d['repeats'], d['region_ms'], 'pcie', round(d['value_pcie'],1))"; done
