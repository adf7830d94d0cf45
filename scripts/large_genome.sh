#!/bin/bash
# One point beyond the Infinity Cache (BASELINE.json configs[3]'s per-GPU share: the human reference's signal arrays are
# 24.8 GB; rawindex.h:32-34): the bench line and the DTW launch's HBM traffic on a reference of GENOME bases (default 10^9:
# an 8 GB arena, 31 times the 256 MiB MALL -- the E. coli arena's 37 MB sit inside it).  GPU box.
# Usage: bash scripts/large_genome.sh <outdir> [genome bases]
OUT=${1:-gpurun_out/large}; GENOME=${2:-1000000000}; mkdir -p $OUT
export RAWDTW_SYNTH_CACHE=/tmp/rawdtw_synth
step() { echo "[$(date +%T)] $1"; }
timeout -k 10 900 python -u bench.py --genome $GENOME --modes-reads 0 --rounds 0 --mapper-reads 0 --cpu-seconds 1.5 > $OUT/bench_large.json 2> $OUT/bench_large.err; step "bench rc=$?"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  RAWDTW_PROBE_GENOME=$GENOME timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/$c -- python3 scripts/stream_probe.py 16384 > $OUT/pmc_$c.log 2>&1; step "pmc $c rc=$?"
done
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc/summary.json
python3 scripts/make_traffic.py $OUT/pmc/summary.json "k_runs<256, false>" 16384 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over scripts/stream_probe.py with RAWDTW_PROBE_GENOME=$GENOME (one bench batch on a reference of $GENOME bases, every launch alone on the chip): scripts/large_genome.sh" > $OUT/traffic_large.json; step "traffic rc=$?"
rm -rf /tmp/rawdtw_synth
ls -la $OUT
