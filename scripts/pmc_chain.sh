#!/bin/bash
# PMC passes over the device-chained mapper's first round (scripts/mapper_probe.py): instruction mix and wait cycles of k_chain (the anchor sort and
# the chaining DP, a wave a read), and its HBM bytes in passes of their own.  Usage (GPU box): bash scripts/pmc_chain.sh <outdir>
OUT=${1:-gpurun_out/pmc_chain}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/mapper_probe.py 16384 1,16,0,dev > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY
import csv, glob, json, collections
out = {}
for name in ("sq1", "sq2", "fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_chain(" not in r["Kernel_Name"] or int(r["Grid_Size"]) != 16384 * 64:
                continue  # (the rounds of all 16 384 reads)
            out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
# one row a dispatch and counter (summed over the chip's instances by rocprofv3's csv: one line per dispatch and counter)
summ = {k: sum(v) / len(v) for k, v in out.items()}
summ["dispatches"] = {k: len(v) for k, v in out.items()}
json.dump(summ, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
