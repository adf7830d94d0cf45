#!/bin/bash
# one PMC pass (instruction mix) over the bench with serialised launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmcq}; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial-launches --inflight 1 > $OUT/b.json 2> $OUT/b.err
python3 scripts/pmc_summary.py $OUT | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if 'rawdtw' in k: print(k[:44], {a:round(b/1e6,2) for a,b in v.items() if a!='dispatches'})"
