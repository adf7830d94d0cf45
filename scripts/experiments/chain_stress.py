#!/usr/bin/env python3
"""Randomised parity sweep of rawdtw_chain_round against the host restatement (tests/test_device_chain.py's helpers) over many more reads and option sets
than the test suite holds: python scripts/experiments/chain_stress.py [rounds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rawalign_amd as ra  # noqa: E402
from rawalign_amd import mapping as M  # noqa: E402
from tests.test_device_chain import compare, device_round, random_read  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = ra.Engine(0)
t0 = time.time()
total = 0
for it in range(rounds):
    copt = M.ChainOpt(int(rng.choice([50, 500, 2000])), int(rng.choice([100, 1000, 5000])), int(rng.choice([5, 40, 5000])), int(rng.choice([0, 2, 25, 200])),
                      int(rng.choice([1, 2, 3, 5])), int(rng.choice([1, 2, 3, 6])), float(rng.choice([0.0, 10.0, 25.0])), int(rng.choice([3, 6, 9])), int(rng.random() < 0.2))
    reads = []
    for _ in range(int(rng.integers(20, 120))):
        n = int(rng.choice([0, 1, 2, 5, 30, 64, 65, 127, 128, 129, 200, 400, 900, 1500, 2048]) if rng.random() < 0.3 else rng.integers(1, 500))
        reads.append(random_read(rng, n, int(rng.integers(1, 7)), int(rng.integers(50, 30000)), dup=float(rng.choice([0.0, 0.05, 0.3])), lines=int(rng.integers(1, 5))))
    out = device_round(eng, copt, reads, n_keys=16)
    if out[0] != 0:
        print("round", it, "declined:", eng.lib.rawdtw_last_error(eng._ctx).decode(), flush=True)
        continue
    compare(eng.lib, copt, reads, out)
    total += len(reads)
    print("round", it, "ok:", len(reads), "reads, options", [getattr(copt, f[0]) for f in copt._fields_], "chains", int(out[1][-1]), flush=True)
print("all equal:", total, "reads in %.1f s" % (time.time() - t0))
