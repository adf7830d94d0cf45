#!/bin/bash
# Soak of the chunk-round mapper (thread pool, two read groups, chaining on the device and on the host): many short runs with changing thread counts;
# every run must end and report the same PAF hash for its flow.  bash scripts/experiments/mapper_soak.sh [minutes]
MIN=${1:-5}
END=$(( $(date +%s) + MIN * 60 ))
n=0
while [ $(date +%s) -lt $END ]; do
  for t in 1 2 3 5 8 13 16; do
    timeout -k 10 120 python -u scripts/mapper_probe.py 2048 1,$t,0 2,$t,1 1,$t,0,dev 2,$t,0,dev 2,$t,0,all,dev > gpurun_out/x/soak_one.txt 2>&1 || { echo "FAILED at run $n threads $t rc=$?"; tail -5 gpurun_out/x/soak_one.txt; exit 1; }
    n=$((n+1))
  done
  echo "[$(date +%T)] $n runs done"
done
echo "soak ok: $n runs"
