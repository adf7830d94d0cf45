"""Access to tests/golden/dtw_golden.npz (captured from the reference by scripts/make_golden.py)."""
import os
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


@dataclass
class GoldenCase:
    a: np.ndarray
    b: np.ndarray
    R0: int
    exclude_last: int
    global_bits: int
    banded_bits: int
    tb: tuple | None  # (cost_bits, i, j, d_bits)


def load_golden():
    z = np.load(os.path.join(HERE, "golden", "dtw_golden.npz"))
    vals = z["vals"]
    out = []
    for k, (n, m, R0, ex, off, tb_off) in enumerate(z["cases"]):
        a = np.ascontiguousarray(vals[off:off + n])
        b = np.ascontiguousarray(vals[off + n:off + n + m])
        tb = None
        if tb_off >= 0:
            ln = int(z["tb_len"][k])
            tb = (int(z["tb_cost"][k]), z["tb_i"][tb_off:tb_off + ln], z["tb_j"][tb_off:tb_off + ln],
                  z["tb_d"][tb_off:tb_off + ln])
        out.append(GoldenCase(a, b, int(R0), int(ex), int(z["global_"][k]), int(z["banded"][k]), tb))
    return out


def bits(x) -> int:
    return int(np.float32(x).view(np.uint32))
