#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — applies one of the two batched-driver patches of INTEGRATION.md §4 to scratch copies of four reference files
(kpp.f90, gas.f, aer.f, tot.f) and writes the unified diff a maintainer would apply.

    two_pass_patch.py [--mode two-pass|drive|liq] <reference src dir> <scratch dir> [<diff out>]

two-pass (shim/kpp_two_pass.patch): the layer loop runs twice around one batched INTEGRATE_x per mechanism; pack, rates, budgets and
hand-over stay in Fortran.  drive (shim/kpp_drive.patch): the loop runs once, x_drive stages its layer behind its /kpp_rate_x/ prologue
and ONE mistra_chem_drive call per mechanism does pack -> rates -> integrator -> budgets -> hand-over on the device (shim/mistra_kpp_drive.f90).

liq (shim/kpp_liq.patch): liq_parm's calls of cw_rc, v_mean_x, henry_x, st_coeff_x, equil_co_x, fast_k_mt_x, dry_cw_rc and dry_rates_x (kpp.f90:609-653) go to the
drop-ins of shim/mistra_kpp_model.f90, which have the reference's own argument lists (kpp.f90 only).

Nothing under the reference tree is touched; the scratch copies live under oracle/_ref/ (git-ignored).  The edits are
anchored on the exact reference lines they follow or replace (kpp.f90:4168-4470, gas.f:172-173, aer.f:216-217,
tot.f:603-604) and fail loudly if an anchor is missing."""
import difflib
import os
import sys


def edit(text, old, new, count=1):
    if text.count(old) != count:
        raise SystemExit("anchor not found exactly %d time(s):\n%s" % (count, old))
    return text.replace(old, new)


def patch_kpp(t):
    # the driver's USE list: the batch module
    t = edit(t, "subroutine kpp_driver (box,dd_ch,n_bl)\n", "subroutine kpp_driver (box,dd_ch,n_bl)\n\n  USE mistra_kpp_batch, ONLY : kpp_pass, kpp_batch_begin, kpp_batch_run   ! two-pass layer loop (shim/)\n")
    # the layer loop runs twice around one batched integration per mechanism
    t = edit(t, "  do k=n_min,n_max\n\n! define temp, H2O, air, ..",
             "  call kpp_batch_begin\n  do kpp_pass=1,2      ! pass 1: pack + rates, layers recorded; pass 2: budgets + hand-over of the batched results\n"
             "  if (kpp_pass.eq.2) call kpp_batch_run (0.d0,dd_ch)\n  do k=n_min,n_max\n\n! define temp, H2O, air, ..")
    # the one statement of the loop body that is not idempotent: advection is applied in pass 1 only
    t = edit(t, "     if (neula.eq.0) then\n        if (k.le.kinv) then\n           do j=1,nadvmax",
             "     if (neula.eq.0 .and. kpp_pass.ne.2) then\n        if (k.le.kinv) then\n           do j=1,nadvmax")
    t = edit(t, "  enddo ! k\n\n! eliminate negative values\n  where (s1 < 0.d0) s1 = 0._dp",
             "  enddo ! k\n  enddo ! kpp_pass\n  kpp_pass=0\n\n! eliminate negative values\n  where (s1 < 0.d0) s1 = 0._dp")
    return t


def patch_drive(t, sfx, indent):
    use_anchor = {"g": "      subroutine gas_drive\n", "a": "      subroutine aer_drive\n", "t": "      subroutine tot_drive\n"}[sfx]
    if t.count(use_anchor) != 1:
        raise SystemExit("x_drive header not found for " + sfx)
    # first USE statement of x_drive: add ours in front of it
    head = t.index(use_anchor)
    first_use = t.index("      USE ", head)
    t = t[:first_use] + "      USE mistra_kpp_batch, ONLY : kpp_pass\n" + t[first_use:]
    call = "%scall INTEGRATE_%s (tkpp" % (indent, sfx)
    at = t.index(call)
    eol = t.index("\n", at)
    t = t[:eol + 1] + "%sif (kpp_pass.eq.1) return   ! pass 1 of the batched driver: this layer's C and RCONST are recorded\n" % indent + t[eol + 1:]
    return t


def patch_kpp_single(t):
    t = edit(t, "subroutine kpp_driver (box,dd_ch,n_bl)\n", "subroutine kpp_driver (box,dd_ch,n_bl)\n\n  USE mistra_kpp_drive, ONLY : kpp_drive_begin   ! batched layer loop: one device call per mechanism (shim/)\n")
    t = edit(t, "  do k=n_min,n_max\n\n! define temp, H2O, air, ..",
             "  call kpp_drive_begin      ! from here on x_drive stages its layer instead of integrating it (kpp_pass = 3)\n  do k=n_min,n_max\n\n! define temp, H2O, air, ..")
    t = edit(t, "  enddo ! k\n\n! eliminate negative values\n  where (s1 < 0.d0) s1 = 0._dp",
             "  enddo ! k\n  call KPP_DRIVE_RUN (0.d0,dd_ch)      ! pack, rates, integrator, budgets, hand-over of all staged layers on the device; kpp_pass = 0 again\n"
             "\n! eliminate negative values\n  where (s1 < 0.d0) s1 = 0._dp")
    return t


def patch_drive_single(t, sfx, anchor):
    use_anchor = {"g": "      subroutine gas_drive\n", "a": "      subroutine aer_drive\n", "t": "      subroutine tot_drive\n"}[sfx]
    if t.count(use_anchor) != 1:
        raise SystemExit("x_drive header not found for " + sfx)
    head = t.index(use_anchor)
    first_use = t.index("      USE ", head)
    t = t[:first_use] + "      USE mistra_kpp_batch, ONLY : kpp_pass\n" + t[first_use:]
    # behind the /kpp_rate_x/ prologue, in front of the first statement of the pack half
    if t.count(anchor) != 1:
        raise SystemExit("pack anchor not found exactly once for " + sfx)
    at = t.index(anchor)
    block = ("      if (kpp_pass.eq.3) then   ! batched driver: the layer is handed over here, the device does the rest\n"
             "         call KPP_DRIVE_STAGE_%s (k,air,h2o)\n"
             "         return\n"
             "      end if\n" % sfx)
    return t[:at] + block + t[at:]


def patch_kpp_liq(text):
    a, b = text.index("subroutine liq_parm (xra,box,n_bl)"), text.index("end subroutine liq_parm")      # (initc calls some of the routines too: not touched)
    t = text[a:b]
    for old, new in (("  call cw_rc (nmaxf)\n", "  call CW_RC_HIP (nmaxf)      ! the particle bins' water and switches on the device (shim/mistra_kpp_model.f90)\n"),
                     ("  call v_mean_a (t,nmaxf)", "  call V_MEAN_HIP_a (t,nmaxf)"), ("  call henry_a (t,nmaxf)\n", "  call HENRY_HIP_a (t,nmaxf)\n"),
                     ("  call st_coeff_a\n", "  call ST_COEFF_HIP_a\n"), ("  call equil_co_a (t,nmaxf)\n", "  call EQUIL_CO_HIP_a (t,nmaxf)\n"),
                     ("call fast_k_mt_a(freep,box,n_bl)\n", "call FAST_K_MT_HIP_a(freep,box,n_bl)\n"),
                     ("     call v_mean_t (t,nmaxf)", "     call V_MEAN_HIP_t (t,nmaxf)"), ("     call henry_t (t,nmaxf)\n", "     call HENRY_HIP_t (t,nmaxf)\n"),
                     ("     call st_coeff_t\n", "     call ST_COEFF_HIP_t\n"), ("     call equil_co_t (t,nmaxf)\n", "     call EQUIL_CO_HIP_t (t,nmaxf)\n"),
                     ("call fast_k_mt_t(freep,box,n_bl)\n", "call FAST_K_MT_HIP_t(freep,box,n_bl)\n"),
                     ("  call dry_cw_rc (nmax)\n", "  call DRY_CW_RC_HIP (nmax)\n"), ("  call dry_rates_g (t,freep,nmax)\n", "  call DRY_RATES_HIP_g (t,freep,nmax)\n"),
                     ("  call dry_rates_a (freep,nmaxf)\n", "  call DRY_RATES_HIP_a (freep,nmaxf)\n"), ("  call dry_rates_t (freep,nmaxf)\n", "  call DRY_RATES_HIP_t (freep,nmaxf)\n")):
        t = edit(t, old, new)
    return text[:a] + t + text[b:]


def main():
    mode = "two-pass"
    if sys.argv[1] == "--mode":
        mode = sys.argv[2]
        del sys.argv[1:3]
    src, scratch = sys.argv[1], sys.argv[2]
    os.makedirs(scratch, exist_ok=True)
    diff = []
    if mode == "drive":
        edits = (("kpp.f90", patch_kpp_single),
                 ("gas.f", lambda t: patch_drive_single(t, "g", "! Transfer Mistra concentration arrays towards KPP arrays\n      do j=1,j1\n         C(gas_m2k_g(1,j))")),
                 ("aer.f", lambda t: patch_drive_single(t, "a", "c include C(ind_)=s1/3(k,)\n      do j=1,j1\n         C(gas_m2k_a(1,j))")),
                 ("tot.f", lambda t: patch_drive_single(t, "t", "! Transfer Mistra concentration arrays towards KPP arrays\n      do j=1,j1\n         C(gas_m2k_t(1,j))")))
    elif mode == "two-pass":
        edits = (("kpp.f90", patch_kpp), ("gas.f", lambda t: patch_drive(t, "g", "      ")),
                 ("aer.f", lambda t: patch_drive(t, "a", "      ")), ("tot.f", lambda t: patch_drive(t, "t", "         ")))
    elif mode == "liq":
        edits = (("kpp.f90", patch_kpp_liq),)
    else:
        raise SystemExit("unknown mode " + mode)
    for name, fn in edits:
        old = open(os.path.join(src, name), errors="replace").read()
        new = fn(old)
        open(os.path.join(scratch, name), "w").write(new)
        diff += list(difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/src/" + name, "b/src/" + name, n=2))
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("".join(diff))
    print("patched %d file(s) into" % len(edits), scratch, "(%s, %d diff lines)" % (mode, len(diff)))


if __name__ == "__main__":
    main()
