cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in ${CFGS:-"stream_debug=128" "stream_debug=129" "stream_debug=131"}; do  # k_runs: nothing off, no DP, no DP and no staging
  OUT=${PMC_OUT:-gpurun_out/pmc_masks}/$cfg; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq1 -- python3 scripts/stream_probe.py 16384 "$cfg" > $OUT/log 2>&1
  echo "== $cfg"; python3 scripts/pmc_summary.py $OUT | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if 'k_runs' in k and 'cells' not in k: print({a:round(b/1e6,2) for a,b in v.items() if a!='dispatches'})"
done
