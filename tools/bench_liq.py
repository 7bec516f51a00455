#!/usr/bin/env python3
"""Times the liq_parm kernels of SURVEY §8 f3 on the device for ONE model column, the unit liq_parm works on (kpp.f90:516-657: layers 2..nf;
BTZ96 has nf = 100, i.e. 99 layers), beside what the reference's Fortran routines took inside the running model on the build container's host
(profiles/r04_liq_parm_reference_cpu.txt, oracle/time_liq_wrap.c).  The column is the captured layers of tests/golden/*.npz repeated to 99 layers:
real spectra, real temperatures.  Per routine: `device_us` = HIP events around the device-pointer entry on torch's current stream (median of
--reps), `host_us` = wall time of the HOST-buffer entry the Fortran drop-ins call (inputs gathered into a pinned arena, up, the kernel, down; median),
`pinned_us` = the same with the caller's large arrays registered by mistra_chem_pin_host, as the drop-ins do for the model's COMMON blocks.
Usage (GPU box): python tools/bench_liq.py [--layers 99] [--reps 30] > gpurun_out/liq_bench.txt"""
import argparse
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, REPO)


def tile(a, n):
    a = np.asarray(a)
    reps = (n + a.shape[0] - 1) // a.shape[0]
    return np.ascontiguousarray(np.concatenate([a] * reps)[:n])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=99)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    import torch
    from mistra_amd import chem
    assert torch.cuda.is_available(), "needs the GPU"
    chem.init(0)
    L, dev, nl = chem.lib(), torch.device("cuda", 0), args.layers
    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    P = lambda a, t=dp: None if a is None else a.ctypes.data_as(t)
    G = lambda name: {k: v for k, v in np.load(os.path.join(REPO, "tests", "golden", name)).items()}      # (materialised: an NpzFile re-reads the archive at every access)

    def device_us(fn):
        fn(); torch.cuda.synchronize()
        out = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3)
        return statistics.median(out)

    def host_us(item):
        fn, big = item
        out2 = []
        for pin in (False, True):
            if pin:
                for b in big: chem.pin_host(b)
            try:
                out2.append(_host_us(fn))
            finally:
                if pin:
                    for b in big: chem.unpin_host(b)
        return out2

    def _host_us(fn):
        fn()
        out = []
        for _ in range(args.reps):
            t0 = time.perf_counter(); fn(); out.append((time.perf_counter() - t0) * 1e6)
        return statistics.median(out)

    rows = []
    for mech in ("aer", "tot"):
        mid = chem._mech_id(mech)[0]
        k, q = G("kmt_%s.npz" % mech), G("liq_%s.npz" % mech)
        nspec = k["alpha"].shape[1]
        h = {n: tile(k[n], nl) for n in ("ff", "cw", "cm", "freep", "alpha", "vmean", "xkmt_before", "t", "p", "vt_before")}
        kw, ka, ifeed, nkc_l = np.ascontiguousarray(k["kw"], np.int32), int(k["ka"]), int(k["ifeed"]), int(k["nkc_l"])
        d = {n: T(v) for n, v in h.items()}
        rq = T(k["rq"])
        rq_h = np.ascontiguousarray(k["rq"])
        rows.append(("fast_k_mt_" + mech[0],
                     device_us(lambda: chem.fast_k_mt(mech, d["ff"], rq, kw, ka, ifeed, nkc_l, d["cw"], d["cm"], d["freep"], d["alpha"], d["vmean"], d["xkmt_before"],
                                                      d["t"], d["p"], d["vt_before"])),
                     host_us((lambda: chem._check(L.mistra_chem_fast_k_mt(mid, nl, P(h["ff"]), P(rq_h), P(kw, ip), 70, ka, ifeed, nkc_l, P(h["cw"]),
                                                                          P(h["cm"]), P(h["freep"]), P(h["alpha"]), P(h["vmean"]), P(h["xkmt_before"]), P(h["t"]), P(h["p"]),
                                                                          P(h["vt_before"]))), [h["ff"], h["xkmt_before"], h["alpha"], h["vmean"]])),
                     "%d of %d layers with an active bin" % (int((h["cm"] > 0).any(axis=1).sum()), nl)))
        tt = tile(q["henry_tt"], nl)
        out_h, dtt, dout = np.zeros((nl, nspec)), T(tt), T(np.zeros((nl, nspec)))
        rows.append(("henry_" + mech[0], device_us(lambda: chem.henry(mech, dtt, dout)), host_us((lambda: chem._check(L.mistra_chem_henry(mid, nl, P(tt), P(out_h))), [out_h])), ""))
        rows.append(("v_mean_" + mech[0], device_us(lambda: chem.v_mean(mech, dtt, dout)), host_us((lambda: chem._check(L.mistra_chem_v_mean(mid, nl, P(tt), P(out_h))), [out_h])), ""))
        s = G("stcoeff_%s.npz" % mech)
        env = tile(s["env"], nl)
        denv = T(env)
        rows.append(("st_coeff_" + mech[0], device_us(lambda: chem.st_coeff(mech, denv, dout)),
                     host_us((lambda: chem._check(L.mistra_chem_st_coeff(mid, nl, 0, 0, P(env), P(out_h))), [out_h])), ""))
        e = {n: tile(q[n], nl) for n in ("equil_tt", "conv2", "xgamma", "xkef_before", "xkeb_before")}
        de = {n: T(v) for n, v in e.items()}
        nkc, j6 = e["conv2"].shape[1], e["xgamma"].shape[2]
        rows.append(("equil_co_" + mech[0], device_us(lambda: chem.equil_co(mech, de["equil_tt"], de["conv2"], de["xgamma"], de["xkef_before"], de["xkeb_before"])),
                     host_us((lambda: chem._check(L.mistra_chem_equil_co(mid, nl, nkc, j6, P(e["equil_tt"]), P(e["conv2"]), P(e["xgamma"]), P(e["xkef_before"]),
                                                                         P(e["xkeb_before"]))), [e["xkef_before"], e["xkeb_before"], e["xgamma"]])), ""))
    c = G("cwrc.npz")
    ff, feu, cloud = tile(c["wet_ff"], nl), tile(c["wet_feu"], nl), tile(c["wet_cloud"], nl)
    rows.append(("cw_rc", None, host_us((lambda: chem.cw_rc(ff, c["rq"], c["e"], c["kw"], int(c["ka"]), int(c["ifeed"]), feu, cloud, c["crys4"]), [ff])), ""))
    ffd = tile(c["dry_ff"], nl)
    rows.append(("dry_cw_rc", None, host_us((lambda: chem.cw_rc(ffd, c["rq"], c["e"], c["kw"], int(c["ka"]), int(c["ifeed"]), dry=True), [ffd])), ""))
    r = G("dryrates.npz")
    for mech in ("gas", "aer", "tot"):
        a = [tile(r[mech + "_" + n], nl) for n in ("tt", "freep", "rcd")]
        four = tile(r["gas_henry4_before" if mech == "gas" else mech + "_vmean4"], nl)
        rows.append(("dry_rates_" + mech[0], None, host_us(((lambda a=a, four=four: chem.dry_rates(*a, None, four)) if mech == "gas" else (lambda a=a, four=four: chem.dry_rates(*a, four)), [])), ""))
    print("# liq_parm kernels on the device, one column of %d layers; %s; median of %d" % (nl, torch.cuda.get_device_name(0), args.reps))
    print("%-14s %12s %12s %12s  %s" % ("routine", "device_us", "host_us", "pinned_us", "note"))
    for name, du, hu, note in rows:
        print("%-14s %12s %12.1f %12.1f  %s" % (name, "-" if du is None else "%.1f" % du, hu[0], hu[1], note))


if __name__ == "__main__":
    main()
