#!/usr/bin/env python3
"""Randomised sweep: the library's mapper with the chaining on the host against the same with the chaining on the device -- PAF lines and score log
equal -- over random references (1-9 sequences), read counts, chunk counts, thread / group counts, flags, --min-events and stop rules.
python scripts/experiments/mapper_chain_sweep.py [cases] [seed]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rawalign_amd as ra  # noqa: E402
from rawalign_amd import mapper, synth  # noqa: E402
from rawalign_amd.mapping import StopOpt  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for it in range(cases):
    n_seq = int(rng.integers(1, 10))
    ref = synth.make_reference([int(rng.integers(6000, 60000)) for _ in range(n_seq)], seed=int(rng.integers(1, 10 ** 6)))
    n = int(rng.integers(20, 250))
    seeds = mapper.SyntheticSeeds(ref, n, seed=int(rng.integers(1, 10 ** 6)), max_chunks=int(rng.integers(1, 7)), hit_prob=float(rng.choice([0.1, 0.2, 0.35])),
                                  false_hits=int(rng.choice([5, 25, 80])))
    names, lens = [f"seq{s}" for s in range(ref.n_seq)], [len(x) for x in ref.forward]
    slot = max(rd["n_ev"] for rd in seeds.reads) + 8
    never = rng.random() < 0.4
    me = int(rng.choice([50, 50, 300]))
    stop = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6, min_events=me) if never else StopOpt(min_events=me)
    flag = int(rng.choice([0x2, 0x2, 0x2 | 0x8, 0x8, 0x2 | 0x4]))
    border, fill = (1, 1) if flag & 0x4 == 0 or rng.random() < 0.5 else (int(rng.choice([0, 1])), 0)
    opt = ra.MapOpt(dtw_border_constraint=border, dtw_fill_method=fill, flag=flag)
    out = {}
    for dev in (0, 1):
        threads, groups = int(rng.integers(1, 9)), int(rng.integers(1, 3))
        eng = ra.Engine(0)
        eng.upload_reference(ref.forward, ref.reverse)
        cm = mapper.CMapper(eng, opt, stop, names, lens, slot_events=slot, max_reads=n, carry=bool(rng.integers(0, 2)), threads=threads, groups=groups, device_chain=bool(dev))
        lines, rounds = mapper.map_reads_c(seeds, list(range(n)), cm)
        out[dev] = (hashlib.sha1("\n".join(lines).encode()).hexdigest(), hashlib.sha1(cm.log().encode()).hexdigest(), rounds, cm.timing()["anchor_bytes"])
        cm.close()
        eng.close()
    ok = out[0][:3] == out[1][:3]
    bad += 0 if ok else 1
    print("case", it, "ok " if ok else "BAD", "seqs", n_seq, "reads", n, "flag", hex(flag), "border", border, "fill", fill, "never" if never else "stop", "min_events", me,
          "rounds", out[0][2], "host-chained rounds' anchor bytes on the device side", out[1][3], flush=True)
print("mismatches:", bad, "of", cases)
sys.exit(1 if bad else 0)
