#!/usr/bin/env python3
"""Bring-up of the device planner: one batch, device-planned, verified, compared with the host-planned one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rawalign_amd as ra
from rawalign_amd import synth
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 300
def say(*a): print(*a, file=sys.stderr, flush=True)
ref = synth.make_reference([150000], seed=91)
res = {}
for dev in (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")):
    eng = ra.Engine(0)
    eng.set_option("device_plan", dev); eng.set_option("device_plan_min_jobs", 0)
    eng.upload_reference(ref.forward, ref.reverse)
    offs = {(0, st): eng.reference_offset(0, st) for st in (0, 1)}
    cb, _ = synth.make_candidate_batch(ref, offs, synth.SynthParams(n_reads=reads, max_chunks=4, decoys_per_read=2.0), seed=4322)
    eng.upload_events(cb.events)
    say("dev", dev, "creating batch")
    b = ra.Batch(eng, ra.MapOpt(dtw_border_constraint=int(os.environ.get("BORDER", "1")), dtw_fill_method=int(os.environ.get("FILL", "1"))), cb)
    say("created; verifying")
    say("device planned:", b.verify_plan())
    say(b.info())
    b.run(); res[dev] = b.fetch(with_job_costs=True)
    say("ran")
if len(res) == 2:
    print("costs equal:", np.array_equal(res[0][2].view(np.uint32), res[1][2].view(np.uint32)), "scores equal:", np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32)))
