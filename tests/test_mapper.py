"""End-to-end control flow (chunk rounds -> chaining -> DTW batch -> primary chains / MAPQ / stop rule ->
PAF): the device path must produce exactly the PAF lines of the same flow scored by the CPU oracle.
d1-like scale (configs[0]: SARS-CoV-2, 29,903 bp) on the CPU leg."""
import numpy as np
import pytest

import rawalign_amd as ra
from rawalign_amd import mapper, synth


def setup(n_reads=24):
    ref = synth.make_reference([29903], seed=20231005 + 1)
    seeds = mapper.SyntheticSeeds(ref, n_reads, seed=3, max_chunks=4)
    return ref, seeds


def test_oracle_path_produces_paf(oracle):
    ref, seeds = setup(12)
    opt = ra.MapOpt()
    lines, rounds = mapper.map_reads(seeds, list(range(12)), mapper.OracleScorer(oracle, ref), opt)
    assert len(lines) == 12 and rounds >= 1
    mapped = [l for l in lines if l.split("\t")[4] in "+-"]
    assert len(mapped) >= 6  # most mappable reads map
    for l in mapped:
        f = l.split("\t")
        r = int(f[0].split("_")[1])
        rd = seeds.reads[r]
        assert f[4] == ("-" if rd["strand"] else "+")
        assert int(f[6]) == 29898  # signal length of the only sequence (29903 - k + 1)


@pytest.mark.gpu
def test_device_path_paf_identical_to_oracle_path(oracle):
    ref, seeds = setup(40)
    eng = ra.Engine(0)
    eng.upload_reference(ref.forward, ref.reverse)
    for opt in (ra.MapOpt(), ra.MapOpt(dtw_border_constraint=0, dtw_fill_method=1),
                ra.MapOpt(dtw_border_constraint=1, dtw_fill_method=0)):
        a, ra_ = mapper.map_reads(seeds, list(range(40)), mapper.OracleScorer(oracle, ref), opt)
        b, rb_ = mapper.map_reads(seeds, list(range(40)), mapper.DeviceScorer(eng), opt)
        assert ra_ == rb_ and a == b
