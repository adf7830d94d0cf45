#!/bin/bash
# Round-2 PMC passes over the stream probe (one bench batch, every launch alone on the chip): instruction mix, wait
# cycles, and HBM FETCH_SIZE / WRITE_SIZE in their own passes (MI355X_MICROARCH.md HBM section).
# Usage (GPU box): bash scripts/pmc_r02.sh <outdir> [probe args]
OUT=${1:-gpurun_out/pmc_r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/stream_probe.py "${PROBE_ARGS[@]}" > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
PROBE_ARGS=("$@")
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_summary.py $OUT > $OUT/summary.json
python3 -c "
import json,sys; d=json.load(open('$OUT/summary.json'))
for k,v in d.items():
    if 'rawdtw' in k: print(k[:40].ljust(40), {a:round(b/1e6,3) for a,b in v.items()})"
