#!/bin/bash
# two PMC passes (instruction mix, wait/busy cycles) over the stream probe: one batch, launches alone on the chip.
# Usage (GPU box): bash scripts/pmc_quick.sh <outdir> [probe args]
OUT=${1:-gpurun_out/pmcq}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq1 -- python3 scripts/stream_probe.py "$@" > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 scripts/stream_probe.py "$@" > $OUT/sq2.log 2>&1
python3 scripts/pmc_summary.py $OUT | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if 'rawdtw' in k or 'rocprim' in k: print(k[:40].ljust(40), {a:round(b/1e6,3) for a,b in v.items()})"
