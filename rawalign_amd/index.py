"""RawAlign `.ind` index files: reader for what the DTW path needs (sequence table + per-strand
reference signal arrays, src/rawindex.cpp:317-377) and a writer of the same format
(src/rawindex.cpp:275-315) used to make synthetic indices for tests and benchmarks.  The hash
buckets (seeding, out of scope) are written empty and never read."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from ._lib import load_library

RI_IDX_MAGIC = b"RI"  # src/rawindex.h:7-8, two bytes


class Index:
    def __init__(self, path: str):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.rawdtw_index_open(path.encode(), C.byref(h))
        if st != 0:
            raise ValueError(f"{path}: not a RawAlign index (status {st})")
        self._h = h
        n = C.c_uint32()
        pars = (C.c_uint32 * 8)()
        self.lib.rawdtw_index_info(h, C.byref(n), pars)
        self.n_seq = n.value
        self.w, self.e, self.n, self.q, self.lq, self.k, _, self.flag = [int(x) for x in pars]
        self.names, self.lens = [], []
        for i in range(self.n_seq):
            name, ln = C.c_char_p(), C.c_uint32()
            self.lib.rawdtw_index_seq(h, i, C.byref(name), C.byref(ln))
            self.names.append((name.value or b"").decode())
            self.lens.append(ln.value)

    def signal(self, seq: int, strand: int) -> np.ndarray:
        out = np.empty(self.lens[seq], np.float32)
        st = self.lib.rawdtw_index_read_signal(self._h, seq, strand, out.ctypes.data_as(C.c_void_p))
        if st != 0:
            raise IOError("short index file")
        return out

    def upload(self, engine):
        """Stream every sequence's forward/reverse signal into the engine's reference arena."""
        engine._check(self.lib.rawdtw_index_upload(engine._ctx, self._h))

    def close(self):
        if getattr(self, "_h", None) is not None:
            self.lib.rawdtw_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_index(path: str, names, forward, reverse, w=0, e=6, n=0, q=9, lq=3, k=6, flag=0, b=14):
    """ri_idx_dump's layout with empty hash buckets (2^b of them: u32 n=0, u32 size=0)."""
    with open(path, "wb") as f:
        f.write(RI_IDX_MAGIC)
        f.write(struct.pack("<8I", w, e, n, q, lq, k, len(names), flag))
        for name, fw, rv in zip(names, forward, reverse):
            nb = name.encode()
            assert len(nb) < 256 and len(fw) == len(rv)
            f.write(struct.pack("<B", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", len(fw)))
            f.write(np.ascontiguousarray(fw, "<f4").tobytes())
            f.write(np.ascontiguousarray(rv, "<f4").tobytes())
        f.write(b"\x00" * (8 * (1 << b)))
