"""Host-side mirror of the reference's integrator interface over the C ABI (include/mistra_chem.h).

The reference exposes, per mechanism x in {g, a, t}, `SUBROUTINE INTEGRATE_x(TIN, TOUT)` working on
`COMMON /GDATA_x/ C(NSPEC), RCONST(NREACT), ...` (gas.f:710, gas_Global.h:29-58 | aer.f:1408 | tot.f:2812).
Here the same call takes the arrays explicitly and handles any number of cells:

    integrate("tot", var, fix, rconst, tin=0.0, tout=10.0) -> IntegrateResult(var, ierr, stats)

`var`, `fix`, `rconst` are cell-major arrays [ncell, NVAR|NFIX|NREACT]: numpy arrays go through the host-buffer
entry point, torch CUDA tensors stay on the device (`mistra_chem_integrate_device`, asynchronous on the current
torch stream).  There is no CPU implementation in this package: without the HIP library and a GPU the calls raise.
"""
import ctypes as C
import os
from collections import namedtuple

import numpy as np

from .mechtab import MECH_IDS, MECH_NAMES

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISTRA_CHEM_LIB", os.path.join(_PKG, "lib", "libmistra_chem.so"))   # override: diagnostic builds

DIMS = {"gas": (102, 3, 331, 1110), "aer": (257, 5, 979, 6579), "tot": (417, 7, 1627, 13503)}
STAT_NAMES = ("Nfun", "Njac", "Nstp", "Nacc", "Nrej", "Ndec", "Nsol", "Nsng")   # COMMON /Statistics/, gas.f:913
IERR_TEXT = {1: "success", -6: "too many steps", -7: "step size too small", -8: "matrix repeatedly singular"}

IntegrateResult = namedtuple("IntegrateResult", "var ierr stats")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None
_inited_device = None


class MistraChemError(RuntimeError):
    pass


def lib():
    """The C-ABI library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MistraChemError("HIP library %s is missing: run `python -m mistra_amd.build` (or __graft_entry__.build())"
                                  % LIB_PATH)
        try:
            # PyTorch ships its own libamdhip64 / libhsa-runtime64.  Loaded first, the library below binds to those by soname and the
            # process has ONE HIP runtime; loaded after this library (which then has pulled in /opt/rocm's), the process has two,
            # and the one that initialises second sees no device (found with __graft_entry__.build() followed by smoke()).
            import torch  # noqa: F401
        except ImportError:      # a caller without PyTorch (numpy buffers only): the system runtime alone
            pass
        L = C.CDLL(LIB_PATH)
        L.mistra_chem_init.argtypes = [C.c_int]
        L.mistra_chem_init_devices.argtypes = [C.c_int, _ip]
        L.mistra_chem_integrate_ex.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _ip, _ip, _dp]
        L.mistra_chem_integrate_env_ex.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _ip, _ip, _dp]
        L.mistra_chem_integrate_common_status.argtypes = [C.c_int, C.c_void_p, _dp, _dp, _ip, _dp, _dp, _ip]
        L.mistra_chem_finalize.restype = None
        L.mistra_chem_dims.argtypes = [C.c_int, _ip, _ip, _ip, _ip]
        L.mistra_chem_integrate.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _ip, _ip]
        L.mistra_chem_integrate_device.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                                   C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mistra_chem_integrate_device_hstart.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                                          C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                          C.c_void_p]
        L.mistra_chem_rates_env_size.argtypes = [C.c_int]
        L.mistra_chem_update_rconst.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.mistra_chem_update_rconst_device.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mistra_chem_integrate_common.argtypes = [C.c_int, C.c_void_p, _dp, _dp]
        L.mistra_chem_singular_rows.argtypes = [C.c_int, C.c_int, _ip]
        vp = C.c_void_p
        L.mistra_chem_set_species_maps.argtypes = [C.c_int, C.c_int, _ip, _ip, C.c_int, _ip, _ip]
        L.mistra_chem_drive_dims.argtypes = [C.c_int, _ip, _ip, _ip, _ip]
        L.mistra_chem_pack_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
        L.mistra_chem_unpack_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
        L.mistra_chem_budgets_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, vp, vp]
        L.mistra_chem_rates_env_from_c_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp]
        L.mistra_chem_fast_k_mt_device.argtypes = [C.c_int, C.c_int, vp, vp, _ip, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.mistra_chem_henry_device.argtypes = [C.c_int, C.c_int, vp, vp, vp]
        L.mistra_chem_equil_co_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
        if hasattr(L, "mistra_chem_v_mean_device"):      # (see below: older builds in same-box A/B runs)
            L.mistra_chem_v_mean_device.argtypes = [C.c_int, C.c_int, vp, vp, vp]
        if hasattr(L, "mistra_chem_st_coeff_device"):
            L.mistra_chem_st_coeff_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.mistra_chem_drive_device.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp, vp, vp]
        if hasattr(L, "mistra_chem_drive"):      # (an older build of the library loaded for a same-box A/B, tools/ab_many.sh, does not have these)
            L.mistra_chem_drive.argtypes = [C.c_int, C.c_int, _ip, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _ip, _ip, _dp, _dp, C.c_int,
                                            _ip, _dp, _dp]
            L.mistra_chem_debug_set_max_steps.argtypes = [C.c_int]
        if hasattr(L, "mistra_chem_drive_begin"):
            L.mistra_chem_drive_begin.argtypes = L.mistra_chem_drive.argtypes
            L.mistra_chem_drive_end.argtypes = [C.c_int]
        L.mistra_chem_last_error.restype = C.c_char_p
        L.mistra_chem_describe.restype = C.c_char_p
        L.mistra_chem_describe.argtypes = [C.c_int]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MistraChemError(lib().mistra_chem_last_error().decode())


def init(device=0):
    """Select the GPU, load mechanism tables and upload the kernel schedules (idempotent per device).  A device that
    init_devices() already set up is left as it is."""
    global _inited_device
    if _inited_device is not None and (device == _inited_device or (isinstance(_inited_device, tuple) and device in _inited_device)):
        return
    _check(lib().mistra_chem_init(int(device)))
    _inited_device = device


def init_devices(devices):
    """Several GPUs in one process (mistra_chem_init_devices): `devices` = a count or a list of HIP device indices.  Host-buffer
    calls are then split into one block of cells per device; device-buffer calls run where their tensors live."""
    global _inited_device
    ids = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
    arr = (C.c_int32 * len(ids))(*ids)
    _check(lib().mistra_chem_init_devices(len(ids), arr))
    _inited_device = tuple(ids)


def finalize():
    global _inited_device
    if _lib is not None:
        _lib.mistra_chem_finalize()
    _inited_device = None


def describe(mech):
    return lib().mistra_chem_describe(MECH_IDS[mech]).decode()


def _mech_id(mech):
    if isinstance(mech, str):
        return MECH_IDS[mech], mech
    return int(mech), MECH_NAMES[int(mech)]


def integrate(mech, var, fix, rconst, tin=0.0, tout=10.0, device=None):
    """INTEGRATE_x over a batch of cells.  numpy in -> numpy out (synchronous); torch CUDA tensors in -> torch
    tensors out on the same device, enqueued on the current torch stream."""
    mid, name = _mech_id(mech)
    nvar, nfix, nreact, _ = DIMS[name]
    try:
        import torch
        is_torch = isinstance(var, torch.Tensor)
    except ImportError:      # pragma: no cover
        is_torch = False
    if is_torch:
        return _integrate_torch(mid, name, var, fix, rconst, tin, tout)
    if device is not None or _inited_device is None:      # host data: whatever device(s) the library already runs on
        init(0 if device is None else device)
    v = np.ascontiguousarray(var, np.float64).reshape(-1, nvar)
    ncell = v.shape[0]
    f = np.ascontiguousarray(fix, np.float64).reshape(ncell, nfix)
    r = np.ascontiguousarray(rconst, np.float64).reshape(ncell, nreact)
    out = np.empty_like(v)
    ierr = np.zeros(ncell, np.int32)
    stats = np.zeros((ncell, 8), np.int32)
    _check(lib().mistra_chem_integrate(mid, ncell, v.ctypes.data_as(_dp), f.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
                                      float(tin), float(tout), out.ctypes.data_as(_dp), ierr.ctypes.data_as(_ip),
                                      stats.ctypes.data_as(_ip)))
    return IntegrateResult(out, ierr, stats)


def integrate_ex(mech, var, fix, rconst, tin=0.0, tout=10.0, env=None):
    """mistra_chem_integrate_ex on host buffers: what `integrate` returns plus t_h [ncell, 3] — per cell the exit time (-> TIN,
    gas.f:769), the last accepted step (-> STEPMIN, gas.f:770) and H when the integrator returned.  This is the call the batched
    Fortran surface makes (shim/mistra_kpp_shim.f90: INTEGRATE_BATCH_x); with several devices initialised (init_devices) the
    library cuts the batch into one block per device."""
    mid, name = _mech_id(mech)
    nvar, nfix, nreact, _ = DIMS[name]
    if _inited_device is None:
        init(0)
    v = np.ascontiguousarray(var, np.float64).reshape(-1, nvar)
    ncell = v.shape[0]
    f = np.ascontiguousarray(fix, np.float64).reshape(ncell, nfix)
    if env is not None:      # mistra_chem_integrate_env_ex: the rate evaluator's inputs instead of the rate constants
        r = np.ascontiguousarray(env, np.float64).reshape(ncell, -1)
        assert r.shape[1] == lib().mistra_chem_rates_env_size(mid)
    else:
        r = np.ascontiguousarray(rconst, np.float64).reshape(ncell, nreact)
    out = np.empty_like(v)
    ierr = np.zeros(ncell, np.int32)
    stats = np.zeros((ncell, 8), np.int32)
    th = np.zeros((ncell, 3))
    _check((lib().mistra_chem_integrate_env_ex if env is not None else lib().mistra_chem_integrate_ex)(mid, ncell, v.ctypes.data_as(_dp), f.ctypes.data_as(_dp), r.ctypes.data_as(_dp),
                                         float(tin), float(tout), out.ctypes.data_as(_dp), ierr.ctypes.data_as(_ip),
                                         stats.ctypes.data_as(_ip), th.ctypes.data_as(_dp)))
    return IntegrateResult(out, ierr, stats), th


def singular_rows(mech, cell):
    """Rows (1-based) of the zero pivots cell `cell` of the last host-buffer integrate call met (include/mistra_chem.h:
    mistra_chem_singular_rows); meaningful for the first min(Nsng, 8) entries."""
    mid, _ = _mech_id(mech)
    rows = np.zeros(8, np.int32)
    _check(lib().mistra_chem_singular_rows(mid, int(cell), rows.ctypes.data_as(_ip)))
    return rows


def device_count():
    return lib().mistra_chem_device_count()


def _integrate_torch(mid, name, var, fix, rconst, tin, tout, out=None, ierr=None, stats=None, texit_hexit=None, hstart=None):
    import torch
    nvar, nfix, nreact, _ = DIMS[name]
    if not var.is_cuda:
        raise MistraChemError("torch tensors must live on the GPU (there is no CPU path); pass numpy arrays for host data")
    dev = var.device.index or 0
    init(dev)
    for x, n in ((var, nvar), (fix, nfix), (rconst, nreact)):
        if x.dtype != torch.float64 or not x.is_contiguous() or x.shape[-1] != n or x.device != var.device:
            raise MistraChemError("expected contiguous float64 [ncell,%d] tensors on one device" % n)
    ncell = var.numel() // nvar
    if fix.numel() != ncell * nfix or rconst.numel() != ncell * nreact:
        raise MistraChemError("cell counts of var / fix / rconst differ")
    out = torch.empty_like(var) if out is None else out
    ierr = torch.empty(ncell, dtype=torch.int32, device=var.device) if ierr is None else ierr
    stats = torch.empty((ncell, 8), dtype=torch.int32, device=var.device) if stats is None else stats
    stream = torch.cuda.current_stream(var.device).cuda_stream
    _check(lib().mistra_chem_integrate_device_hstart(mid, ncell, var.data_ptr(), fix.data_ptr(), rconst.data_ptr(), float(tin),
                                                    float(tout), out.data_ptr(), ierr.data_ptr(), stats.data_ptr(),
                                                    None if texit_hexit is None else texit_hexit.data_ptr(),
                                                    None if hstart is None else hstart.data_ptr(), C.c_void_p(stream)))
    return IntegrateResult(out, ierr, stats)


def update_rconst(mech, env):
    """Update_RCONST_x for a batch of cells (gas.f:275): env [ncell, rates_env_size] -> rconst [ncell, NREACT].  numpy in ->
    numpy out; torch CUDA tensor in -> torch tensor out, on torch's current stream.  Entry names of env: mistra_amd/mech/<mech>.rates_env.json."""
    mid, name = _mech_id(mech)
    nreact = DIMS[name][2]
    try:
        import torch
        is_torch = isinstance(env, torch.Tensor)
    except ImportError:      # pragma: no cover
        is_torch = False
    if is_torch:
        init(env.device.index or 0)
        ne = lib().mistra_chem_rates_env_size(mid)
        if env.dtype != torch.float64 or not env.is_contiguous() or env.shape[-1] != ne:
            raise MistraChemError("expected a contiguous float64 [ncell,%d] tensor" % ne)
        out = torch.empty((env.numel() // ne, nreact), dtype=torch.float64, device=env.device)
        stream = torch.cuda.current_stream(env.device).cuda_stream
        _check(lib().mistra_chem_update_rconst_device(mid, out.shape[0], env.data_ptr(), out.data_ptr(), C.c_void_p(stream)))
        return out
    if _inited_device is None:
        init(0)
    ne = lib().mistra_chem_rates_env_size(mid)
    e = np.ascontiguousarray(env, np.float64).reshape(-1, ne if ne else 1)
    out = np.empty((e.shape[0], nreact))
    _check(lib().mistra_chem_update_rconst(mid, e.shape[0], e.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
    return out


def debug_set_max_steps(n=0):
    """Test hook: Max_no_steps of the integrator (0 = the reference's 100000); makes IERR = -6 reachable (include/mistra_chem.h)."""
    _check(lib().mistra_chem_debug_set_max_steps(int(n)))


def integrate_into(mech, var, fix, rconst, out, ierr, stats, tin=0.0, tout=10.0, texit_hexit=None, hstart=None):
    """Device path with caller-owned output tensors (no allocation inside the timed region of bench.py).  texit_hexit
    [ncell, 2]: exit time and last step size per cell (what INTEGRATE_x leaves in TIN and STEPMIN).  hstart [ncell]: OPT-IN
    first step size per cell instead of the reference's 1e-3 (include/mistra_chem.h: mistra_chem_integrate_device_hstart)."""
    mid, name = _mech_id(mech)
    return _integrate_torch(mid, name, var, fix, rconst, tin, tout, out, ierr, stats, texit_hexit, hstart)


# ---- the hand-over halves of x_drive on the device (include/mistra_chem.h).  torch CUDA tensors, float64, contiguous,
#      cell-major; everything runs on torch's current stream.
def set_species_maps(mech, gas_m2k, gas_k2m, rad_m2k, rad_k2m):
    """The model's species index maps for one mechanism (module gas_common: gas_m2k_x(1:2,j), gas_k2m_x(j), rad_*; 1-based)."""
    mid, _ = _mech_id(mech)
    if _inited_device is None:
        init(0)
    a = [np.ascontiguousarray(x, np.int32) for x in (gas_m2k, gas_k2m, rad_m2k, rad_k2m)]
    _check(lib().mistra_chem_set_species_maps(mid, len(a[1]), a[0].ctypes.data_as(_ip), a[1].ctypes.data_as(_ip), len(a[3]),
                                              a[2].ctypes.data_as(_ip), a[3].ctypes.data_as(_ip)))


def drive_dims(mech):
    """(j2, j6, nkc, nbgs): dimensions of sl1(j2,nkc,n), sion1(j6,nkc,n) and of bgs(2,nbgs,n)"""
    mid, _ = _mech_id(mech)
    v = [C.c_int32() for _ in range(4)]
    _check(lib().mistra_chem_drive_dims(mid, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def _stream(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def pack(mech, s1, s3, sl1, sion1, scal, var, fix):
    """x_drive up to Update_RCONST_x: var, fix (in/out: entries the driver does not set keep their content) from the layer arrays; sl1 / sion1
    are clamped in place where the driver does (aer, tot)."""
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_pack_device(mid, var.shape[0], _p(s1), _p(s3), _p(sl1), _p(sion1), _p(scal), _p(var), _p(fix), _stream(var)))


def unpack(mech, var, s1, s3, sl1, sion1):
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_unpack_device(mid, var.shape[0], _p(var), _p(s1), _p(s3), _p(sl1), _p(sion1), _stream(var)))


def budgets(mech, var, fix, rconst, dt, bg=None, bgs=None):
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_budgets_device(mid, var.shape[0], _p(var), _p(fix), _p(rconst), float(dt), _p(bg), _p(bgs), _stream(var)))


def rates_env_from_c(mech, var, fix, env):
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_rates_env_from_c_device(mid, var.shape[0], _p(var), _p(fix), _p(env), _stream(var)))


def drive(mech, s1, s3, sl1, sion1, scal, env, var, fix, tin, dt, ierr, stats, texit_hexit=None, bg=None, bgs=None):
    """One x_drive per layer for a batch of layers, device-resident: pack -> rates -> INTEGRATE_x(tin, tin + dt) -> budgets -> hand-over."""
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_drive_device(mid, var.shape[0], _p(s1), _p(s3), _p(sl1), _p(sion1), _p(scal), _p(env), _p(var), _p(fix), float(tin),
                                          float(dt), _p(ierr), _p(stats), _p(texit_hexit), _p(bg), _p(bgs), _stream(var)))


def drive_host(mech, layer, s1, s3, sl1, sion1, scal, env, tin, dt, bg=None, bg_level=None, bgs=None, want_c=False, begin_only=False):
    """mistra_chem_drive: one x_drive per layer for the layers `layer` (k, 1-based) of the model arrays in HOST memory — numpy float64,
    C-contiguous, s1 [n, j1], s3 [n, j5], sl1 [n, nkc*j2], sion1 [n, nkc*j6], bg [nlev, nrxn, 2], bgs [n, 122, 2], updated in place; scal
    [nlayer, 6], env [nlayer, nenv] per layer of the batch.  -> (ierr, stats, t_h[, c_packed]).  begin_only: mistra_chem_drive_begin — the call returns
    with the work enqueued, the outputs are valid after drive_host_end(mech)."""
    mid, name = _mech_id(mech)
    lay = np.ascontiguousarray(layer, np.int32)
    nl = lay.size
    for a in (s1, s3, sl1, sion1) + ((bg,) if bg is not None else ()) + ((bgs,) if bgs is not None else ()):
        if a.dtype != np.float64 or not a.flags.c_contiguous:
            raise MistraChemError("model arrays must be C-contiguous float64 (they are updated in place)")
    sc, ev = np.ascontiguousarray(scal, np.float64), np.ascontiguousarray(env, np.float64)
    ierr, stats, th = np.zeros(nl, np.int32), np.zeros((nl, 8), np.int32), np.zeros((nl, 3))
    nvar, nfix, _, _ = DIMS[name]
    cp = np.zeros((nl, nvar + nfix)) if want_c else None
    lev = np.ascontiguousarray(bg_level, np.int32) if bg_level is not None else None
    P = lambda a: None if a is None else a.ctypes.data_as(_dp)
    fn = lib().mistra_chem_drive_begin if begin_only else lib().mistra_chem_drive
    _check(fn(mid, nl, lay.ctypes.data_as(_ip), s1.shape[0], P(s1), P(s3), P(sl1), P(sion1), P(sc), P(ev), float(tin), float(dt),
              ierr.ctypes.data_as(_ip), stats.ctypes.data_as(_ip), P(th), P(bg), 0 if bg is None else bg.shape[1],
              None if lev is None else lev.ctypes.data_as(_ip), P(bgs), P(cp)))
    out = (ierr, stats, th, cp) if want_c else (ierr, stats, th)
    return out + ((lay, sc, ev, lev),) if begin_only else out      # (begin_only: the inputs ride along so that they outlive the call)


def drive_host_end(mech):
    """mistra_chem_drive_end: waits for the step drive_host(..., begin_only=True) issued for this mechanism and scatters its results."""
    _check(lib().mistra_chem_drive_end(_mech_id(mech)[0]))


def fast_k_mt(mech, ff, rq, kw, ka, ifeed, nkc_l, cw, cm, freep, alpha, vmean, xkmt, t=None, p=None, vt=None):
    """fast_k_mt_a (aer) / fast_k_mt_t (tot) for a batch of layers: xkmt [nlayer, nkc, NSPEC] and, when given, the sedimentation velocity
    vt [nlayer, nkc] (needs t, p [nlayer]) updated in place (include/mistra_chem.h)."""
    mid, _ = _mech_id(mech)
    kwa = np.ascontiguousarray(kw, np.int32)
    _check(lib().mistra_chem_fast_k_mt_device(mid, xkmt.shape[0], _p(ff), _p(rq), kwa.ctypes.data_as(_ip), int(kwa.size), int(ka), int(ifeed), int(nkc_l),
                                              _p(cw), _p(cm), _p(freep), _p(alpha), _p(vmean), _p(xkmt), _p(t), _p(p), _p(vt), _stream(xkmt)))


def henry(mech, tt, out):
    """henry_a (aer) / henry_t (tot) for a batch of layers: out [nlayer, NSPEC] <- tt [nlayer] (include/mistra_chem.h)."""
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_henry_device(mid, out.shape[0], _p(tt), _p(out), _stream(out)))


def v_mean(mech, tt, out):
    """v_mean_a (aer) / v_mean_t (tot) for a batch of layers: out [nlayer, NSPEC] <- tt [nlayer] (include/mistra_chem.h)."""
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_v_mean_device(mid, out.shape[0], _p(tt), _p(out), _stream(out)))


def dry_rates(tt, freep, rcd, vmean4=None, henry4=None):
    """dry_rates_a / dry_rates_t (vmean4 [nlayer, 4] given) or dry_rates_g (henry4 [nlayer, 4] given, returned updated) for a batch of layers
    -> xkmtd [nlayer, 2, 4], xeq [nlayer] (, henry4); numpy arrays (host-buffer entry, include/mistra_chem.h)."""
    L = lib()
    if not hasattr(L, "_dry_rates_typed"):
        L.mistra_chem_dry_rates.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L._dry_rates_typed = True
    gas = vmean4 is None
    f8 = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
    tt, freep, rcd, vmean4 = f8(tt), f8(freep), f8(rcd), f8(vmean4)
    h = None if henry4 is None else np.array(henry4, np.float64, order="C", copy=True)
    nl = tt.shape[0]
    xk, xeq = np.full((nl, 2, 4), np.nan), np.full(nl, np.nan)
    P = lambda a: None if a is None else a.ctypes.data_as(_dp)
    _check(L.mistra_chem_dry_rates(int(gas), nl, P(tt), P(freep), P(rcd), P(vmean4), P(xk), P(xeq), P(h)))
    return (xk, xeq, h) if gas else (xk, xeq)


def pin_host(a):
    """Registers the memory of a C-contiguous numpy array for direct transfers by the host-buffer entries (include/mistra_chem.h: mistra_chem_pin_host).
    The array must stay alive, and must not be resized, until unpin_host(a)."""
    assert a.flags["C_CONTIGUOUS"]
    L = lib()
    L.mistra_chem_pin_host.argtypes = [C.c_void_p, C.c_size_t]
    _check(L.mistra_chem_pin_host(a.ctypes.data, a.nbytes))


def unpin_host(a):
    L = lib()
    L.mistra_chem_unpin_host.argtypes = [C.c_void_p]
    _check(L.mistra_chem_unpin_host(a.ctypes.data))


def cw_rc(ff, rq, e, kw, ka, ifeed, feu=None, cloud=None, crys4=None, dry=False):
    """cw_rc (dry=False: -> rc, cw, cm, conv2 [nlayer, 4], below [nlayer]) or dry_cw_rc (dry=True: -> rcd, cwd [nlayer, 2]) for a batch of layers;
    numpy arrays in and out (host-buffer entry, include/mistra_chem.h)."""
    L = lib()
    if not hasattr(L, "_cw_rc_typed"):
        L.mistra_chem_cw_rc.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _ip, C.c_int, C.c_int, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _ip]
        L._cw_rc_typed = True
    ff = np.ascontiguousarray(ff, np.float64)
    nl, nka, nkt = ff.shape
    nb = 2 if dry else 4
    f8 = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
    rq, e, feu, crys4 = f8(rq), f8(e), f8(feu), f8(crys4)
    kw = np.ascontiguousarray(kw, np.int32)
    cl = None if cloud is None else np.ascontiguousarray(cloud, np.int32)
    rc, cw, cm, cv = (np.full((nl, nb), np.nan) for _ in range(4))
    below = np.full(nl, -1, np.int32)
    P = lambda a, t=_dp: None if a is None else a.ctypes.data_as(t)
    _check(L.mistra_chem_cw_rc(nl, nkt, nka, int(bool(dry)), P(ff), P(rq), P(e), P(kw, _ip), int(ka), int(ifeed), P(feu), P(cl, _ip), P(crys4), P(rc), P(cw),
                               P(cm), P(cv), P(below, _ip)))
    return (rc, cw) if dry else (rc, cw, cm, cv, below)


def st_coeff(mech, env, out, lp_joyce14bc=False, lp_buxmann15alph=False):
    """st_coeff_a (aer) / st_coeff_t (tot) for a batch of layers: out [nlayer, NSPEC] <- env [nlayer, 5] = t, cw(1), cm(1), sion1(13,1), sion1(14,1)
    (include/mistra_chem.h)."""
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_st_coeff_device(mid, out.shape[0], int(bool(lp_joyce14bc)), int(bool(lp_buxmann15alph)), _p(env), _p(out), _stream(out)))


def equil_co(mech, tt, conv2, xgamma, xkef, xkeb):
    """equil_co_a (aer) / equil_co_t (tot) for a batch of layers: xkef, xkeb [nlayer, nkc, NSPEC] updated in place from tt [nlayer],
    conv2 [nlayer, nkc], xgamma [nlayer, nkc, j6] (include/mistra_chem.h)."""
    mid, _ = _mech_id(mech)
    _check(lib().mistra_chem_equil_co_device(mid, xkef.shape[0], xkef.shape[1], xgamma.shape[2], _p(tt), _p(conv2), _p(xgamma), _p(xkef), _p(xkeb), _stream(xkef)))
