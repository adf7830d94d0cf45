import sys, numpy as np
sys.path.insert(0, '.')
import torch
import rawalign_amd as ra
from oracle.loader import Oracle
from tests.util import make_arena_jobs
orc = Oracle()
rng = np.random.default_rng(5)
eng = ra.Engine(0)
cases = []
shapes = [(100, 90, 64), (100, 90, 70), (300, 280, 70), (620, 600, 70), (640, 600, 70), (700, 650, 70), (1300, 1250, 70), (500, 480, 140), (520, 480, 140), (1100, 1000, 140),
          (600, 500, 256), (1200, 1100, 260), (3000, 2800, 300), (5000, 4600, 480), (2600, 2600, 255), (2600, 2600, 256), (2600, 2600, 260), (3000, 2900, 450), (2000, 1900, 318), (2000, 2000, 319), (2000, 2000, 320), (2000, 2000, 447), (2000, 2000, 448), (2000, 2000, 511), (1500, 700, 300), (4000, 3900, 400), (700, 690, 40), (800, 700, 40), (1500, 1400, 40), (300, 100, 30), (900, 300, 30), (2000, 300, 20)]
for n, m, r0 in shapes:
    for ex in (0, 1):
        cases.append((rng.normal(size=n).astype(np.float32), rng.normal(size=m).astype(np.float32), r0, ex))
        cases.append((rng.normal(size=m).astype(np.float32), rng.normal(size=n).astype(np.float32), r0, ex))
jobs, ev, rf = make_arena_jobs(cases)
eng.upload_reference([rf], [rf])
jobs["ref_off"] += eng.reference_offset(0, 1)
got = eng.score_batch(jobs, ev)
for k, (a, b, r0, ex) in enumerate(cases):
    want = orc.dtw_banded(a, b, r0, ex)
    N, M = max(len(a), len(b)), min(len(a), len(b))
    R = r0 + ((N - M) * r0 + N - 1) // N
    print("%5d x %5d r0 %3d K %3d ex %d  %s  got %.4f want %.4f" % (len(a), len(b), r0, R + 1, ex, "ok " if got[k].view(np.uint32) == np.float32(want).view(np.uint32) else "BAD", got[k], want))
