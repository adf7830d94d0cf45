import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rawalign_amd as ra
from rawalign_amd.align import CandidateBatch
from tests import test_stream_path as T
shapes, opts = T._medium, {"tile_lds_floats": 2048}
rng = np.random.default_rng(hash((shapes.__name__, tuple(sorted(opts)))) & 0xFFFF)
ref = [rng.normal(size=60000).astype(np.float32), rng.normal(size=60000).astype(np.float32)]
eng = ra.Engine(0)
for k, v in opts.items(): eng.set_option(k, v)
eng.set_option("stream_debug", 32768)
eng.upload_reference([ref[0]], [ref[1]])
events, chain_off, anchor_off, anchors, slot, read_base = T._chains(rng, 120, 60000, shapes)
strand_of = [1 if s == 0 else 0 for s in slot]
ref_base = np.array([eng.reference_offset(0, st) for st in strand_of], np.uint64)
cb = CandidateBatch(events, chain_off, anchor_off, anchors, ref_base, read_base)
eng.upload_events(events)
b = ra.Batch(eng, ra.MapOpt(dtw_min_score=5.0), cb)
try:
    print(b.verify_plan())
except Exception as e:
    print("ERR", e)
