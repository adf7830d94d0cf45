for o in "" "stream_blocks_per_cu=3" "stream_blocks_per_cu=2" ; do
  for inf in 4 6; do
    RAWDTW_OPTS="$o" python bench.py --steps 20 --warmup 5 --min-region-ms 40 --no-cpu-baseline --inflight $inf 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('opts=[$o] inflight=$inf value=%.1f ms/step=%.3f pcie=%.1f replay=%.1f kstream_pipe=%.3f' % (d['value'], d['ms_per_step'], d['pipeline_pcie']['gcups'], d['kernel_replay']['gcups'], d['launches']['in_pipeline_ms']['k_runs']))"
  done
done
