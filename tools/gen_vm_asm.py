#!/usr/bin/env python3
"""Writes mistra_amd/csrc/vm_exec_asm.inc: the gfx950 instruction stream of the LDS VM executor (ros3_kernel.hip: vm_run),
one variant per ring depth, as C string literals.  The stream is regular but long (N ring slots x {main line, marked rows,
end of round}); generating it keeps the slot arithmetic (register blocks, table offsets, wait counts) in ONE place.

    python tools/gen_vm_asm.py          (the output is committed; tests/test_capi.py checks that it is up to date)

The executor, per record (schedule.hpp: d0 = target, d1 = aux | marks, d2..d4 = a1,r1,u1, d5..d7 = a2,r2,u2):
    acc = M[d0];  acc -= (M[a1]*M[r1])*M[u1];  acc -= (M[a2]*M[r2])*M[u2];  M[d0] = acc      (+ scale / publish on marked rows)
Software pipeline (round 3): while record S computes, the six operand gathers of record S+1 are already in flight — they
were issued at the head of record S into the OTHER of two operand register sets (A for even slots, B for odd ones).  That is
legal because a round never reads a cell it writes, except a record's own target (schedule.cpp: operands are final before
the round starts): only the target read has to wait for the lane's previous store, and LDS is in-order within a wave.  It
stops at a round's end: the next round's operands may be written by other waves until the barrier, so the end-of-round
path gathers them behind it.  A lone wave then no longer sits out one LDS round trip per record (7 gathers, 6 dependent
operations, a store: ~250 cycles); what is left is the instruction issue of the record itself.

Invariant at Lvm_rec<S>: SGPR fl<P> holds the row marks of record S (P = A|B by slot parity) and its six operand gathers
have been issued into set P.
Wait counts (LDS returns in order; in flight, oldest first: [operands of S] [store of S-1] [target of S] [aux of S]
[operands of S+1]); table loads: slot T = S+1 was refilled N-1 records ago, N-2 younger refills (2 loads each) may stay out.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "mistra_amd", "csrc", "vm_exec_asm.inc")

# ring slots: 8-register blocks that are CALLER-saved in the AMDGPU calling convention (v0-39, then every other block of
# 8: 48-55, 64-71, ...), so the non-inlined executor has nothing to save; the kernels held to 128 registers stop at v103
BLOCKS = {4: [48, 64, 80, 96], 6: [48, 64, 80, 96, 112, 128], 8: [48, 64, 80, 96, 112, 128, 144, 160]}
# table address registers: one per pair of slots (13-bit signed immediate offsets reach 2 rows of 2048 bytes)
BASES = ["%[va]", "%[vb]", "%[vc]", "%[vd]"]

RCP = """v_div_scale_f64 %[a1{P}], vcc, %[acc], %[acc], 1.0
v_rcp_f64 %[r1{P}], %[a1{P}]
v_div_scale_f64 %[u1{P}], vcc, 1.0, %[acc], 1.0
v_fma_f64 %[a2{P}], -%[a1{P}], %[r1{P}], 1.0
v_fma_f64 %[r1{P}], %[r1{P}], %[a2{P}], %[r1{P}]
v_fma_f64 %[a2{P}], -%[a1{P}], %[r1{P}], 1.0
v_fma_f64 %[r1{P}], %[r1{P}], %[a2{P}], %[r1{P}]
v_mul_f64 %[a2{P}], %[u1{P}], %[r1{P}]
v_fma_f64 %[a1{P}], -%[a1{P}], %[a2{P}], %[u1{P}]
s_nop 1
v_div_fmas_f64 %[a1{P}], %[a1{P}], %[r1{P}], %[a2{P}]
v_div_fixup_f64 %[a1{P}], %[a1{P}], %[acc], 1.0"""      # the sequence hipcc emits for 1.0/x (IEEE)


def variant(n, upr=2):
    blk = BLOCKS[n]
    L = []
    emit = L.append

    def d(s, k):
        return "v%d" % (blk[s] + k)

    def refill(s):
        base, off = BASES[s // 2], (s % 2) * 2048
        emit("global_load_dwordx4 v[%d:%d], %s, %%[base] offset:%d" % (blk[s], blk[s] + 3, base, off))
        emit("global_load_dwordx4 v[%d:%d], %s, %%[base] offset:%d" % (blk[s] + 4, blk[s] + 7, base, off + 1024))

    def prefetch(t, q):      # marks of record T into fl<q>, its six operand gathers into set q
        emit("v_readfirstlane_b32 %%[fl%s], %s" % (q, d(t, 1)))
        for k, nm in zip(range(2, 8), ("a1", "r1", "u1", "a2", "r2", "u2")):
            emit("ds_read_b64 %%[%s%s], %s" % (nm, q, d(t, k)))

    def muls(p):
        if upr == 2:         # d2..d7 = (a1, r1, u1), (a2, r2, u2): two updates  (M[a]*M[r])*M[u]
            emit("v_mul_f64 %%[a1%s], %%[a1%s], %%[r1%s]" % (p, p, p))
            emit("v_mul_f64 %%[a1%s], %%[a1%s], %%[u1%s]" % (p, p, p))
            emit("v_mul_f64 %%[a2%s], %%[a2%s], %%[r2%s]" % (p, p, p))
            emit("v_mul_f64 %%[a2%s], %%[a2%s], %%[u2%s]" % (p, p, p))
        else:                # d2..d7 = (a1, u1), (a2, u2), (a3, u3) in the registers named a1 r1 | u1 a2 | r2 u2: three updates M[a]*M[u]
            emit("v_mul_f64 %%[a1%s], %%[a1%s], %%[r1%s]" % (p, p, p))
            emit("v_mul_f64 %%[u1%s], %%[u1%s], %%[a2%s]" % (p, p, p))
            emit("v_mul_f64 %%[r2%s], %%[r2%s], %%[u2%s]" % (p, p, p))

    def adds(p):
        if upr == 2:
            emit("v_add_f64 %%[acc], %%[acc], -%%[a1%s]" % p)
            emit("v_add_f64 %%[acc], %%[acc], -%%[a2%s]" % p)
        else:
            emit("v_add_f64 %%[acc], %%[acc], -%%[a1%s]" % p)
            emit("v_add_f64 %%[acc], %%[acc], -%%[u1%s]" % p)
            emit("v_add_f64 %%[acc], %%[acc], -%%[r2%s]" % p)

    def aux_head(s):         # aux operand: scale by M[aux] (aux = d1 & VM_AUX_MASK), or publish the reciprocal there (VM_D1_RCP lanes)
        emit("v_and_b32 %%[ax], 0xfffff8, %s" % d(s, 1))
        emit("v_and_b32 %%[t], 1, %s" % d(s, 1))
        emit("ds_read_b64 %[sc], %[ax]")
        emit("v_cmp_eq_u32 vcc, 1, %[t]")

    def aux_tail(s, p, wait_sc, tag):
        emit("s_mov_b64 %[sv], exec")
        emit("s_andn2_b64 %[sm], exec, vcc")
        emit("s_mov_b64 exec, %[sm]")                      # lanes that scale
        emit("s_waitcnt lgkmcnt(%d)" % wait_sc)
        emit("v_mul_f64 %[sc], %[acc], %[sc]")
        emit("ds_write_b64 %s, %%[sc]" % d(s, 0))
        emit("s_and_b64 exec, %[sv], vcc")                 # lanes that publish
        emit("s_cbranch_execz Lvm_auxe%s%d_%%=" % (tag, s))
        emit("ds_write_b64 %s, %%[acc]" % d(s, 0))
        for ln in RCP.format(P=p).split("\n"):
            emit(ln)
        emit("ds_write_b64 %%[ax], %%[a1%s]" % p)
        emit("Lvm_auxe%s%d_%%=:" % (tag, s))
        emit("s_mov_b64 exec, %[sv]")

    vm_next = 2 * (n - 2)        # before record S's own refill: slots S+1 .. S-1 are out, S+1 the oldest
    vm_after = 2 * (n - 1)       # behind record S's refill
    # ---- prologue
    emit("s_waitcnt vmcnt(0)")   # nothing of the caller's may sit between the counted loads
    emit("s_nop 4")
    for s in range(n):
        refill(s)
    for b in range(n // 2):
        emit("v_add_u32 %s, %d, %s" % (BASES[b], n * 2048, BASES[b]))
    emit("s_waitcnt vmcnt(%d)" % vm_after)
    prefetch(0, "A")
    # ---- main line
    for s in range(n):
        t, p, q = (s + 1) % n, "AB"[s & 1], "AB"[(s + 1) & 1]
        nxt = "Lvm_rec%d_%%=" % t if t else "Lvm_wrap_%="
        emit("Lvm_rec%d_%%=:" % s)
        emit("s_and_b32 %%[tmp], %%[fl%s], 0x7000000" % p)       # any mark (end of round, null row, aux operand)?
        emit("s_cbranch_scc1 Lvm_slow%d_%%=" % s)
        emit("ds_read_b64 %%[acc], %s" % d(s, 0))
        emit("v_mov_b32 %%[tg], %s" % d(s, 0))
        emit("s_waitcnt vmcnt(%d)" % vm_next)
        prefetch(t, q)
        refill(s)                         # the slot's registers are free: its gathers have been issued, the target address copied
        emit("s_waitcnt lgkmcnt(7)")      # this record's operands (and the lane's previous store) are back
        muls(p)
        emit("s_waitcnt lgkmcnt(6)")      # ... and its target
        adds(p)
        emit("ds_write_b64 %[tg], %[acc]")
        if t == 0:
            emit("Lvm_wrap_%=:")
            for b in range(n // 2):
                emit("v_add_u32 %s, %d, %s" % (BASES[b], n * 2048, BASES[b]))
            emit("s_branch Lvm_rec0_%=")
    # ---- out of line: marked rows
    for s in range(n):
        t, p, q = (s + 1) % n, "AB"[s & 1], "AB"[(s + 1) & 1]
        nxt = "Lvm_rec%d_%%=" % t if t else "Lvm_wrap_%="
        emit("Lvm_slow%d_%%=:" % s)
        emit("s_bitcmp1_b32 %%[fl%s], 25" % p)                    # VM_ROW_NULL (always a round's only row: also its end)
        emit("s_cbranch_scc1 Lvm_eorb%d_%%=" % s)
        emit("ds_read_b64 %%[acc], %s" % d(s, 0))
        emit("s_bitcmp1_b32 %%[fl%s], 24" % p)                    # VM_ROW_EOR
        emit("s_cbranch_scc1 Lvm_eor%d_%%=" % s)
        # aux row inside a round
        aux_head(s)
        emit("s_waitcnt vmcnt(%d)" % vm_next)
        prefetch(t, q)
        emit("s_waitcnt lgkmcnt(8)")
        muls(p)
        emit("s_waitcnt lgkmcnt(7)")
        adds(p)
        aux_tail(s, p, 6, "m")
        refill(s)
        emit("s_branch " + nxt)
        # last row of a round: nothing of the next round is read before the barrier
        emit("Lvm_eor%d_%%=:" % s)
        emit("s_bitcmp1_b32 %%[fl%s], 26" % p)                    # VM_ROW_AUX
        emit("s_cbranch_scc0 Lvm_eorp%d_%%=" % s)
        aux_head(s)
        emit("s_waitcnt lgkmcnt(2)")
        muls(p)
        emit("s_waitcnt lgkmcnt(1)")
        adds(p)
        aux_tail(s, p, 0, "r")
        emit("s_branch Lvm_eorb%d_%%=" % s)
        emit("Lvm_eorp%d_%%=:" % s)
        emit("s_waitcnt lgkmcnt(1)")
        muls(p)
        emit("s_waitcnt lgkmcnt(0)")
        adds(p)
        emit("ds_write_b64 %s, %%[acc]" % d(s, 0))
        emit("Lvm_eorb%d_%%=:" % s)
        refill(s)
        emit("s_bitcmp1_b32 %%[fl%s], 27" % p)                    # VM_ROW_LOCAL: the next round is this wave's own (wave 0, a run of small rounds) -
        emit("s_cbranch_scc1 Lvm_loc%d_%%=" % s)                  # no barrier: what it reads next, it has stored itself, and LDS is in-order within a wave
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_barrier")
        emit("s_sub_u32 %[rounds], %[rounds], 1")
        emit("s_cmp_eq_u32 %[rounds], 0")
        emit("s_cbranch_scc1 Lvm_exit_%=")
        emit("Lvm_loc%d_%%=:" % s)
        emit("s_waitcnt vmcnt(%d)" % vm_after)
        prefetch(t, q)
        emit("s_branch " + nxt)
    emit("Lvm_exit_%=:")
    emit("s_waitcnt vmcnt(0)")       # the look-ahead loads must have landed before the ring registers are reused
    clob = ", ".join('"v%d"' % (b + k) for b in blk for k in range(8))
    return L, clob


def render():
    out = ["// GENERATED by tools/gen_vm_asm.py — do not edit.  Instruction stream of the LDS VM executor (ros3_kernel.hip: vm_run).", ""]
    for n, upr in ((4, 2), (6, 2), (8, 2), (4, 3)):
        lines, clob = variant(n, upr)
        tag = "N%d" % n if upr == 2 else "N%d_SWEEP" % n      # _SWEEP: records of three two-operand updates (schedule.hpp: VM_SWEEP_UPD_PER_REC)
        out.append("#define MISTRA_VM_ASM_%s \\" % tag)
        for i, ln in enumerate(lines):
            sep = "\\n" if ln.endswith(":") else "\\n\\t"
            last = i == len(lines) - 1
            out.append('  "%s%s"%s' % (ln, "" if last else sep, "" if last else " \\"))
        out.append("")
        if upr == 2:
            out.append("#define MISTRA_VM_CLOBBER_N%d %s" % (n, clob))
        out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", os.path.normpath(OUT), "(%d lines)" % text.count("\n"))
