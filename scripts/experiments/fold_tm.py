import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
import rawalign_amd as ra
from rawalign_amd import synth
ref = synth.make_reference([4_600_000], seed=20231007)
eng = ra.Engine(0); eng.upload_reference(ref.forward, ref.reverse)
offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=16384), seed=20231007 + 7919)
eng.set_option("stream_debug", 16384)
eng.upload_events(cb.events)
b = ra.Batch(eng, ra.MapOpt(), cb)
cnt = (C.c_uint64 * 64)(); n = C.c_uint32()
prev = None
for reps in (1, 10, 10):
    b.run_reps(reps, timed=False)
    eng.lib.rawdtw_batch_stream_counters(eng._ctx, b._h, cnt, 64, C.byref(n))
    cur = list(cnt[52:58])
    print("fold stamps (ticks): max stage %d max loop %d  sum stage %d sum loop %d  max anchors %d max chains %d" % tuple(cur))
    if prev: print("   per run and wave: stage %.1f loop %.1f" % ((cur[2] - prev[2]) / reps / 2048, (cur[3] - prev[3]) / reps / 2048))
    prev = cur
import torch
print(torch.cuda.get_device_properties(0))
