import numpy as np, sys
sys.path.insert(0,'.')
import rawalign_amd as ra
from rawalign_amd import mapper, synth
from rawalign_amd.mapping import StopOpt
ref = synth.make_reference([29903], seed=20231005 + 1)
seeds = mapper.SyntheticSeeds(ref, 10, seed=3, max_chunks=5)
eng = ra.Engine(0); eng.upload_reference(ref.forward, ref.reverse)
never = StopOpt(min_bestmap_ratio=1e9, min_meanmap_ratio=1e9, min_chain_anchor=10 ** 6)
rs = mapper.RoundScorer(eng, slot_events=max(rd["n_ev"] for rd in seeds.reads) + 8, n_slots=10)
orig = rs.score
def spy(reads, opt, read_keys=None):
    before=(rs.jobs_scored, rs.jobs_reused)
    out = orig(reads, opt, read_keys)
    print("round: reads", len(reads), "scored+", rs.jobs_scored-before[0], "reused+", rs.jobs_reused-before[1], flush=True)
    return out
rs.score = spy
mapper.map_reads(seeds, list(range(10)), rs, ra.MapOpt(), never)
