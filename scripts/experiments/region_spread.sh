for i in 1 2 3 4 5; do timeout -k 10 300 python bench.py --no-cpu-baseline --modes-reads 0 --rounds 0 --min-region-ms 400 > gpurun_out/mr$i.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/mr$i.json').read().strip().splitlines()[-1]); print('mr$i', round(d['value'],1), d['repeats'], d['region_ms'])"; done
