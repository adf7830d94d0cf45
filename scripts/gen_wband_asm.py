#!/usr/bin/env python3
"""Writes rawalign_amd/csrc/rawdtw_wband_asm.h: the main loop of the wave-per-job band body (wband_gen<C>, rawdtw_dp.h) as
hand-scheduled gfx950 assembly, for C = 1, 2, 4, 8 registers a lane (bands of up to 64 C slots).

Why assembly: a long band is one wave's dependency chain, alone on its SIMD -- every instruction costs an issue slot of ~5
clocks whatever it is, so the column's time is its instruction COUNT.  Left to the compiler the loop carried a dozen register
copies a column (the three roles of the a-window, the two of the b-window and the two DP buffers meet in phi nodes), turned
the uniform row-advance branch into per-lane selects and put a fill move in front of every DPP shift.  Here:

  * six columns a loop iteration: the a-window's roles (previous / this / next column) rotate over three register sets, the
    b-window's (this / next) over two -- after six columns every value is back in the register it started in: no copies;
  * the DP buffers are updated in place: the secondary antidiagonal X overwrites d0 (its own top-left operand), the primary
    overwrites d1; a column without a row advance (the rare kind: out of line, behind the loop) computes its primary into d0
    and swaps the two (v_swap_b32);
  * the windows of column c + 1 are asked for (ds_read, immediate offsets within the six columns) while column c is computed;
    s_waitcnt lgkmcnt(n) leaves exactly those in flight;
  * the two DPP shifts a column write into registers whose fill lane holds 1e10 for good (bound_ctrl off: a lane without a
    source keeps the destination's value);
  * two slots' operand differences are one packed subtract (v_pk_add_f32 with the second operand negated);
  * the instructions between a value's last write and its DPP read cover the two wait states the hardware wants there (the
    assembler does not insert them inside inline assembly).

The DP state (d0, d1: 2 C floats a lane) enters and leaves through the wave's piece of LDS -- inline assembly takes at most
thirty operands -- and the extent masks of the 2 C slot registers are made inside from three lane masks a kind: with the
blocked layout (lane l holds slots l C .. l C + C - 1) a register's lanes inside an extent [lo, hi) are `lanes <= hi / C` for
the registers below hi % C and `lanes < hi / C` for the others (and the primaries' lo = 1 takes lane 0 out of register 0).
From three registers a lane on, only the slot right behind an extent is forced to 1e10 -- one select through the VGPR index (`force`).

Cells, neighbours and masks are those of wband_step<C, false> / wreg_gen_step (dtw.cpp:361-485): bit-identical costs
(tests/test_gpu_parity.py::test_random_wave_band, tests/test_stream_path.py: the wave-per-job bands at the register layouts'
edges, scripts/experiments/wband_dbg.py)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VBASE = 64   # first scratch VGPR of the block (clobbered): above what the C++ around it needs, so that the kernels' register count stays low
SBASE = 36   # first scratch SGPR of the block (clobbered): the 2 C extent masks


def gen(C):
    Cp = C + (C & 1)                      # registers a set: 64-bit operands (ds_read2, v_pk_add) want even-aligned pairs
    wide = C >= 4 and C % 4 == 0 and os.environ.get("WBAND_B128", "1") == "1"   # 16-byte reads at any 4-byte boundary (the LDS runs in unaligned mode)
    npair, odd = C // 2, C & 1
    edge = C >= 3   # only the slot behind an extent is forced to 1e10 (through the VGPR index), not every register: see force()
    nd = (C // 4) if wide else (1 if C == 1 else npair + odd)          # ds_read instructions per window
    P = [[VBASE + s * Cp + c for c in range(C)] for s in range(3)]          # a-window register sets
    B = [[VBASE + 3 * Cp + s * Cp + c for c in range(C)] for s in range(2)]  # b-window register sets
    t = VBASE + 5 * Cp
    DIFF = [t + c for c in range(C)]
    D0 = [t + Cp + c for c in range(C)]
    D1 = [t + 2 * Cp + c for c in range(C)]
    t = t + 3 * Cp - 3 * C  # (the four singles below sit behind the last set)
    KHI, KLO, KLO2, VINF = t + 3 * C, t + 3 * C + 1, t + 3 * C + 2, t + 3 * C + 3
    last_v = VINF
    MS = [SBASE + 2 * c for c in range(C)]              # s[MS[c] : MS[c] + 1]: the secondaries' mask of register c
    MP = [SBASE + 2 * C + 2 * c for c in range(C)]
    last_s = SBASE + 4 * C - 1
    names = ["va", "vb", "rem", "it", "adv0", "adv1", "st", "vst", "M", "N", "step", "s_lo", "s_hi", "s_c0", "p_lo", "p_hi", "p_c0", "p_0"]
    op = {n: f"%{k}" for k, n in enumerate(names)}
    v = lambda r: f"v{r}"  # noqa: E731
    d0 = [v(r) for r in D0]
    d1 = [v(r) for r in D1]
    sm = lambda r: f"s[{r}:{r + 1}]"  # noqa: E731
    main, tail = [], []

    def read_window(e, dst, addr, dw_off):
        """C consecutive dwords at LDS byte address `addr` + 4 * dw_off into registers dst[0..C)"""
        if C == 1:
            e(f"ds_read_b32 {v(dst[0])}, {addr} offset:{4 * dw_off}")
        elif wide:
            for q in range(C // 4):
                e(f"ds_read_b128 v[{dst[4 * q]}:{dst[4 * q + 3]}], {addr} offset:{4 * (dw_off + 4 * q)}")
        else:
            for q in range(npair):
                e(f"ds_read2_b32 v[{dst[2 * q]}:{dst[2 * q + 1]}], {addr} offset0:{dw_off + 2 * q} offset1:{dw_off + 2 * q + 1}")
            if odd:
                e(f"ds_read_b32 {v(dst[C - 1])}, {addr} offset:{4 * (dw_off + C - 1)}")

    def diffs(e, a, b):
        """DIFF[c] = a[c] - b[c]"""
        if C == 1:
            e(f"v_sub_f32 {v(DIFF[0])}, {v(a[0])}, {v(b[0])}")
        else:
            for q in range(npair):
                e(f"v_pk_add_f32 v[{DIFF[2 * q]}:{DIFF[2 * q + 1]}], v[{a[2 * q]}:{a[2 * q + 1]}], v[{b[2 * q]}:{b[2 * q + 1]}] neg_lo:[0,1] neg_hi:[0,1]")
            if odd:
                e(f"v_sub_f32 {v(DIFF[C - 1])}, {v(a[C - 1])}, {v(b[C - 1])}")

    def next_adv(e, dst):
        """row advance of the NEXT column: rem += M; adv = rem >= N; rem -= adv ? N : 0; dst = adv ? 4 : 0 (bytes the b-window moves)"""
        e(f"s_add_u32 {op['rem']}, {op['rem']}, {op['M']}")
        e(f"s_cmp_ge_u32 {op['rem']}, {op['N']}")
        e(f"s_cselect_b32 {op['st']}, {op['N']}, 0")
        e(f"s_cselect_b32 {dst}, {op['step']}, 0")
        e(f"s_sub_u32 {op['rem']}, {op['rem']}, {op['st']}")

    e = main.append
    # ---- prologue: the masks, constants, the DP state, the windows of the first column ----
    e("s_waitcnt lgkmcnt(0)")  # (nothing else of this wave's may be counted by the waits below)
    for c in range(C if not edge else 0):
        e(f"s_cmp_lt_u32 {c}, {op['s_c0']}")
        e(f"s_cselect_b64 {sm(MS[c])}, {op['s_hi']}, {op['s_lo']}")
        if c == 0:
            e(f"s_mov_b64 {sm(MP[0])}, {op['p_0']}")
        else:
            e(f"s_cmp_lt_u32 {c}, {op['p_c0']}")
            e(f"s_cselect_b64 {sm(MP[c])}, {op['p_hi']}, {op['p_lo']}")

    def force(e, regs, kind):
        """an antidiagonal's cells outside its extent read as 1e10.  C <= 2: every register is selected on its lanes' mask.  C >= 3: a slot outside
        the extent is read by a slot inside only if it is the one right behind it (a cell's neighbours are its own slot and the slots next to it)
        or, for the primaries that start at slot 1, slot 0 -- so ONE select, on the register that holds that slot (s_c0 / p_c0: picked through the
        VGPR index, the mask s_hi / p_hi has every lane but the slot's), and for the primaries one more on register 0 (p_0); what lies further
        out is never read from inside and may hold anything finite."""
        if not edge:
            masks = MS if kind == "s" else MP
            order = reversed(range(C)) if kind != "tail" else range(C)
            for c in order:
                e(f"v_cndmask_b32_e64 {regs[c]}, {v(VINF)}, {regs[c]}, {sm(masks[c])}")
            return
        idx, mask = (op["s_c0"], op["s_hi"]) if kind == "s" else (op["p_c0"], op["p_hi"])
        e(f"s_set_gpr_idx_on {idx}, 0xa")   # dst and src1 relative
        e("s_nop 0")
        e(f"v_cndmask_b32_e64 {regs[0]}, {v(VINF)}, {regs[0]}, {mask}")
        e("s_set_gpr_idx_off")
        if kind != "s":
            e(f"v_cndmask_b32_e64 {regs[0]}, {v(VINF)}, {regs[0]}, {op['p_0']}")
    e(f"v_mov_b32 {v(VINF)}, 0x501502f9")
    e(f"v_mov_b32 {v(KHI)}, {v(VINF)}")
    e(f"v_mov_b32 {v(KLO)}, {v(VINF)}")
    e(f"v_mov_b32 {v(KLO2)}, {v(VINF)}")
    read_window(e, D0, op["vst"], 0)
    read_window(e, D1, op["vst"], C)
    # va = address of the window of column c0 + 6 (what the group's last column asks for): window(c) = va + 4 * (c0 + 6 - c)
    read_window(e, P[0], op["va"], 7)      # window(c0 - 1)
    read_window(e, P[1], op["va"], 6)      # window(c0)
    next_adv(e, op["adv0"])                # the first column's own advance
    e(f"v_add_u32 {op['vb']}, {op['adv0']}, {op['vb']}")
    read_window(e, B[0], op["vb"], 0)
    e("1:")
    for kcol in range(6):
        pp, pc, pn = P[kcol % 3], P[(kcol + 1) % 3], P[(kcol + 2) % 3]
        bc, bn = B[kcol % 2], B[(kcol + 1) % 2]
        adv, advn = op[f"adv{kcol % 2}"], op[f"adv{(kcol + 1) % 2}"]
        # the next column: its advance, its windows
        next_adv(e, advn)
        e(f"v_add_u32 {op['vb']}, {advn}, {op['vb']}")
        read_window(e, bn, op["vb"], 0)
        read_window(e, pn, op["va"], 5 - kcol)
        e(f"s_waitcnt lgkmcnt({2 * nd})")
        e(f"s_cmp_eq_u32 {adv}, 0")
        e(f"s_cbranch_scc1 2{kcol}f")
        # ---- a column with a row advance: secondary X into d0, primary into d1 ----
        e(f"v_mov_b32_dpp {v(KHI)}, {d1[0]} wave_shl:1 row_mask:0xf bank_mask:0xf")
        diffs(e, pp, bc)
        for c in range(C):  # ascending: X[c] reads d1[c], d1[c + 1] (untouched) and its own d0[c]
            left = d1[c + 1] if c + 1 < C else v(KHI)
            e(f"v_min3_f32 {d0[c]}, {d1[c]}, {left}, {d0[c]}")
        for c in range(C):
            e(f"v_add_f32 {d0[c]}, |{v(DIFF[c])}|, {d0[c]}")
        force(e, d0, "s")  # (C <= 2: the last slot first -- the shift below reads it)
        diffs(e, pc, bc)
        if C == 1:
            e("s_nop 0")  # (two wait states between the select that wrote d0[C-1] and the DPP that reads it: the subtraction is one)
        e(f"v_mov_b32_dpp {v(KLO)}, {d0[C - 1]} wave_shr:1 row_mask:0xf bank_mask:0xf")
        for c in range(C):  # primary[c] reads X[c - 1], X[c] (d0: untouched here) and its own d1[c]
            top = d0[c - 1] if c > 0 else v(KLO)
            e(f"v_min3_f32 {d1[c]}, {top}, {d0[c]}, {d1[c]}")
        for c in range(C):
            e(f"v_add_f32 {d1[c]}, |{v(DIFF[c])}|, {d1[c]}")
        force(e, d1, "p")
        e(f"3{kcol}:")
        # ---- a column without (behind the loop): X = d1; the primary (X[p-1], X[p], d0[p-1]) into d0, then d0 <-> d1 ----
        T = tail.append
        T(f"2{kcol}:")
        T(f"v_mov_b32_dpp {v(KLO)}, {d1[C - 1]} wave_shr:1 row_mask:0xf bank_mask:0xf")
        T(f"v_mov_b32_dpp {v(KLO2)}, {d0[C - 1]} wave_shr:1 row_mask:0xf bank_mask:0xf")
        diffs(T, pc, bc)
        for c in reversed(range(C)):  # descending: primary[c] reads d0[c - 1], not yet overwritten
            top = d1[c - 1] if c > 0 else v(KLO)
            tl = d0[c - 1] if c > 0 else v(KLO2)
            T(f"v_min3_f32 {d0[c]}, {top}, {d1[c]}, {tl}")
        for c in range(C):
            T(f"v_add_f32 {d0[c]}, |{v(DIFF[c])}|, {d0[c]}")
        force(T, d0, "tail")
        for c in range(C):
            T(f"v_swap_b32 {d0[c]}, {d1[c]}")
        T(f"s_branch 3{kcol}b")
    e(f"v_subrev_u32 {op['va']}, 24, {op['va']}")
    e(f"s_sub_u32 {op['it']}, {op['it']}, 1")
    e(f"s_cmp_lg_u32 {op['it']}, 0")
    e("s_cbranch_scc1 1b")
    e("s_branch 9f")
    main.extend(tail)
    e("9:")
    e("s_waitcnt lgkmcnt(0)")
    # the DP state back where it came from
    if C == 1:
        e(f"ds_write_b32 {op['vst']}, {d0[0]}")
        e(f"ds_write_b32 {op['vst']}, {d1[0]} offset:4")
    else:
        for q in range(npair):
            e(f"ds_write2_b32 {op['vst']}, {d0[2 * q]}, {d0[2 * q + 1]} offset0:{2 * q} offset1:{2 * q + 1}")
        if odd:
            e(f"ds_write_b32 {op['vst']}, {d0[C - 1]} offset:{4 * (C - 1)}")
        for q in range(npair):
            e(f"ds_write2_b32 {op['vst']}, {d1[2 * q]}, {d1[2 * q + 1]} offset0:{C + 2 * q} offset1:{C + 2 * q + 1}")
        if odd:
            e(f"ds_write_b32 {op['vst']}, {d1[C - 1]} offset:{4 * (2 * C - 1)}")
    # (the advance computed for the column behind the last one is taken back: rem and vb describe the last column done)
    e(f"v_subrev_u32 {op['vb']}, {op['adv0']}, {op['vb']}")
    e(f"s_cmp_lg_u32 {op['adv0']}, 0")
    e(f"s_cselect_b32 {op['st']}, {op['N']}, 0")
    e(f"s_add_u32 {op['rem']}, {op['rem']}, {op['st']}")
    e(f"s_sub_u32 {op['rem']}, {op['rem']}, {op['M']}")
    e("s_waitcnt lgkmcnt(0)")
    body = "\n".join(f'        "{ln}\\n"' for ln in main)
    outs = '"+v"(va), "+v"(vb), "+s"(rem), "+s"(iters), "=&s"(t_adv0), "=&s"(t_adv1), "=&s"(t_st)'
    ins = '"v"(vstate), "s"(M), "s"(N), "s"(step), "s"(m.sec_lo), "s"(m.sec_hi), "s"(m.sec_c0), "s"(m.prim_lo), "s"(m.prim_hi), "s"(m.prim_c0), "s"(m.prim_0)'
    clob = ", ".join(f'"v{r}"' for r in range(VBASE, last_v + 1)) + ", " + ", ".join(f'"s{r}"' for r in range(SBASE, last_s + 1)) + ', "scc", "memory"'
    return f"""// {C} register(s) a lane: 6 * iters columns from column c0 on.  va: LDS byte address (this lane's) of the a-window of column c0 + 6; vb: of the
// b-window at the centre row BEFORE column c0; vstate: of this lane's DP state (d0[0..{C}), d1[0..{C}): read at the start, written back at the end);
// rem: the Bresenham remainder before c0.  On return rem and vb describe the last column done.
template <> __device__ __forceinline__ void wband_loop_asm<{C}>(uint32_t &va, uint32_t &vb, const uint32_t vstate, uint32_t &rem, uint32_t iters, const uint32_t M,
                                                              const uint32_t N, const WbandMasks &m)
{{
    uint32_t t_adv0, t_adv1, t_st;
    const uint32_t step = 4u; // (a row advance moves the b-window by one element, whatever C is)
    asm volatile(
{body}
        : {outs}
        : {ins}
        : {clob});
}}
"""


def main():
    out = ["// rawdtw_wband_asm.h -- GENERATED by scripts/gen_wband_asm.py (edit that, not this): the main loop of wband_gen<C> (rawdtw_dp.h).",
           "#pragma once", "namespace rawdtw {", "",
           "// the extents of a band's antidiagonals as lane masks of the blocked layout (lane l holds slots l C .. l C + C - 1): register c's lanes",
           "// inside the secondaries' [0, S) are sec_hi for c < sec_c0 and sec_lo otherwise; the primaries' [SH, SH + P) likewise, register 0 apart",
           "// (C >= 3: sec_hi / prim_hi = every lane but the one of the slot right behind the extent, sec_c0 / prim_c0 = that slot's register, prim_0 =",
           "// every lane but lane 0 when the primaries start at slot 1: gen_wband_asm.py, force)",
           "struct WbandMasks { unsigned long long sec_lo, sec_hi, prim_lo, prim_hi, prim_0; uint32_t sec_c0, prim_c0; };",
           "template <int C> __device__ __forceinline__ void wband_loop_asm(uint32_t &va, uint32_t &vb, const uint32_t vstate, uint32_t &rem, uint32_t iters,",
           "                                                              const uint32_t M, const uint32_t N, const WbandMasks &m);", ""]
    for C in (1, 2, 3, 4, 5, 7, 8, 9):
        out.append(gen(C))
    out.append("} // namespace rawdtw")
    path = os.path.join(ROOT, "rawalign_amd", "csrc", "rawdtw_wband_asm.h")
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
