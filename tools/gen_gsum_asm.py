#!/usr/bin/env python3
"""Writes mistra_amd/csrc/gsum_exec_asm.inc: the gfx950 instruction stream of the gather-sum machine (ros3_kernel.hip: gsum_run), one
variant per placement of the look-ahead ring, as C string literals.

    python tools/gen_gsum_asm.py          (the output is committed; tests/test_capi.py checks that it is up to date)

The machine, per table row (schedule.hpp: four LDS byte addresses — the first one carries the flush mark in bit 0 — and four float
coefficients per lane):
    acc = acc + (double)c0 * M[a0];  ... c1, c2, c3 likewise, left to right, one rounding per operation;
    a marked row completes the lane's current output:  M[out] = acc;  out += stride;  acc = -0.0
Why assembly: a wave that walks a long sum alone (four species of tot have 108 terms, 27 rows, where the average wave has 3) is bound
by instruction ISSUE — one instruction per 4-7 cycles whatever it is — and the compiler's version of a row was ~39 instructions: the
table words copied out of the ring (8 moves), then masked, converted and used.  Here the consumers read the ring registers themselves
and the loads take scalar bases (fixed; the lane's offset moves on by 8 KiB per group of four rows): 25 instructions per row.  Software pipeline as in the LDS VM executor: the gathers of row r+1 are
issued before the additions of row r (two operand sets, A and B); only the additions are serial along a sum.  Same terms, same order,
same three operations per term: bit-identical sums.

Ring: 8 slots of 4 registers in caller-saved blocks (ros3_kernel.hip: vm_ring_load), row r of a group of four in slots 2r (addresses)
and 2r+1 (coefficients); a row is refilled with the row four further on as soon as its words have been consumed.
Wait counts: table loads return in order, 8 in flight when a row is taken: vmcnt(6) = its two have landed.  LDS returns in order; when
set P is summed the only younger operations in flight are the 4 gathers of the other set (and possibly the store of a flush in
between, which lgkmcnt(4) then waits for as well).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "mistra_amd", "csrc", "gsum_exec_asm.inc")

SLOTS = {"LOW": [64, 68, 80, 84, 96, 100, 112, 116], "HIGH": [192, 196, 208, 212, 224, 228, 240, 244]}


def variant(name):
    slot = SLOTS[name]
    L = []
    emit = L.append

    def load_row(r):            # rows 0, 1 of a group through %[b0] (immediates < 4096), rows 2, 3 through %[b1] = b0 + 4096
        base, off = ("%[b0]", r * 2048) if r < 2 else ("%[b1]", (r - 2) * 2048)
        a, c = slot[2 * r], slot[2 * r + 1]
        emit("global_load_dwordx4 v[%d:%d], %%[voff], %s offset:%d" % (a, a + 3, base, off))
        emit("global_load_dwordx4 v[%d:%d], %%[voff], %s offset:%d" % (c, c + 3, base, off + 16))

    def fetch(r, s):            # take row r's words: mark, gathers, coefficients into set s; then refill its slots
        a, c = slot[2 * r], slot[2 * r + 1]
        emit("s_waitcnt vmcnt(6)")
        emit("v_and_b32 %%[ad], -8, v%d" % a)
        emit("ds_read_b64 %%[x%s0], %%[ad]" % s)
        for k in (1, 2, 3):
            emit("ds_read_b64 %%[x%s%d], v%d" % (s, k, a + k))
        emit("v_readfirstlane_b32 %%[fl%s], v%d" % (s, a))
        for k in range(4):
            emit("v_cvt_f64_f32 %%[c%s%d], v%d" % (s, k, c + k))
        load_row(r)

    def summ(s, tag):
        emit("s_waitcnt lgkmcnt(4)")
        for k in range(4):
            emit("v_mul_f64 %%[c%s%d], %%[c%s%d], %%[x%s%d]" % (s, k, s, k, s, k))
            emit("v_add_f64 %%[acc], %%[acc], %%[c%s%d]" % (s, k))
        emit("s_bitcmp1_b32 %%[fl%s], 0" % s)
        emit("s_cbranch_scc0 Lgs_nf%s_%%=" % tag)
        emit("ds_write_b64 %[out], %[acc]")
        emit("v_add_u32 %[out], %[out], %[stride]")
        emit("v_mov_b64 %[acc], %[mzero]")
        emit("Lgs_nf%s_%%=:" % tag)

    emit("s_waitcnt vmcnt(0)")          # nothing of the caller's may sit between the counted loads
    emit("s_nop 4")                     # (the bases may have been written by v_readfirstlane just before: 5 wait states before a memory instruction reads them)
    for r in range(4):
        load_row(r)
    emit("v_add_u32 %[voff], 0x2000, %[voff]")      # the next group of four rows
    emit("s_cmp_lt_i32 %[n], 1")
    emit("s_cbranch_scc1 Lgs_exit_%=")
    fetch(0, "A")
    emit("Lgs_loop_%=:")
    fetch(1, "B"); summ("A", "0")
    fetch(2, "A"); summ("B", "1")
    fetch(3, "B"); summ("A", "2")
    emit("v_add_u32 %[voff], 0x2000, %[voff]")
    emit("s_sub_i32 %[n], %[n], 4")
    emit("s_cmp_lt_i32 %[n], 1")
    emit("s_cbranch_scc1 Lgs_last_%=")
    fetch(0, "A"); summ("B", "3")
    emit("s_branch Lgs_loop_%=")
    emit("Lgs_last_%=:")
    emit("s_waitcnt lgkmcnt(0)")
    summ("B", "4")
    emit("Lgs_exit_%=:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")      # the look-ahead loads have landed before the ring registers are reused; stores done
    clob = ", ".join('"v%d"' % (b + k) for b in slot for k in range(4))
    return L, clob


def render():
    out = ["// GENERATED by tools/gen_gsum_asm.py — do not edit.  Instruction stream of the gather-sum machine (ros3_kernel.hip: gsum_run).", ""]
    for name in ("LOW", "HIGH"):
        lines, clob = variant(name)
        out.append("#define MISTRA_GSUM_ASM_%s \\" % name)
        for i, ln in enumerate(lines):
            sep = "\\n" if ln.endswith(":") else "\\n\\t"
            last = i == len(lines) - 1
            out.append('  "%s%s"%s' % (ln, "" if last else sep, "" if last else " \\"))
        out.append("")
        out.append("#define MISTRA_GSUM_CLOBBER_%s %s" % (name, clob))
        out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", os.path.normpath(OUT), "(%d lines)" % text.count("\n"))
