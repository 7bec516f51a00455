// C ABI of libmistra_chem.so (include/mistra_chem.h).  Host plumbing only: load tables, compile schedules, move
// buffers, launch the HIP kernel.  There is deliberately NO host compute path here — if the device is unusable the
// calls fail.
#include "../../include/mistra_chem.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kernel_args.hpp"
#include "mech_tables.hpp"
#include "ros3_kernel.hpp"
#include "schedule.hpp"

using namespace mistra;

namespace {

thread_local std::string g_err;
int fail(const std::string& msg) {
  g_err = msg;
  return 1;
}
#define HIP_TRY(expr)                                                                                    \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));               \
  } while (0)

const char* kMechName[3] = {"gas", "aer", "tot"};
const int kDims[3][4] = {{102, 3, 331, 1110}, {257, 5, 979, 6579}, {417, 7, 1627, 13503}};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t upload(const std::vector<T>& v) {
    release();
    n = v.size();
    if (!n) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(p, v.data(), n * sizeof(T), hipMemcpyHostToDevice);
  }
  hipError_t reserve(size_t count) {
    if (count <= n) return hipSuccess;
    release();
    n = count;
    return hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

struct VmBufs {
  DevBuf<uint32_t> wave_base, recs;
  DevBuf<uint16_t> blk_n;
  int nrounds = 0;
  hipError_t upload(const VmProgram& P) {
    nrounds = P.nrounds;
    hipError_t e;
    if ((e = wave_base.upload(P.wave_base)) != hipSuccess) return e;
    if ((e = blk_n.upload(P.blk_n)) != hipSuccess) return e;
    return recs.upload(P.recs);
  }
  VmDev dev() const { return VmDev{wave_base.p, blk_n.p, recs.p, nrounds}; }
  void release() { wave_base.release(); recs.release(); blk_n.release(); }
};

struct GsBufs {
  DevBuf<uint32_t> wave_base, recs;
  DevBuf<uint16_t> rows;
  hipError_t upload(const GsumProgram& P) {
    hipError_t e;
    if ((e = wave_base.upload(P.wave_base)) != hipSuccess) return e;
    if ((e = rows.upload(P.rows)) != hipSuccess) return e;
    return recs.upload(P.recs);
  }
  GsDev dev() const { return GsDev{wave_base.p, rows.p, recs.p}; }
  void release() { wave_base.release(); recs.release(); rows.release(); }
};

struct MechState {
  bool ready = false;
  int nt = 0, n_temps = 0;
  MechTables tab;
  std::string text;
  DevBuf<double> consts;
  DevBuf<uint64_t> fun_fac, jac_fac;
  DevBuf<uint16_t> jvs_pos, zero_pos, diag_pos;
  GsBufs vdot, jvs;
  VmBufs lu, solve_head_fwd, solve_head_bwd;
  DevBuf<uint32_t> tail_fwd, tail_bwd, lu_scale;
  DevBuf<uint32_t> dense_rows;
  int lu_scale_slots = 0;
  // staging for the host-buffer entry point (grow-only)
  DevBuf<double> s_var, s_fix, s_rct, s_out, s_th;
  DevBuf<int32_t> s_ierr, s_stats;
  // one-cell calls (the Fortran shim): one contiguous device block in, one out, pinned host mirrors, a private stream
  double* one_dev = nullptr;      // [C(NSPEC) | RCONST(NREACT)]  then  [VAR out | Texit Hexit | 8 stats + ierr as int32]
  double* one_host = nullptr;
  hipStream_t one_stream = nullptr;
  void release() {
    consts.release(); fun_fac.release(); jac_fac.release(); jvs_pos.release(); zero_pos.release(); diag_pos.release();
    vdot.release(); jvs.release(); lu.release(); solve_head_fwd.release(); solve_head_bwd.release();
    tail_fwd.release(); tail_bwd.release(); lu_scale.release();
    dense_rows.release();
    s_var.release(); s_fix.release(); s_rct.release(); s_out.release(); s_th.release(); s_ierr.release(); s_stats.release();
    if (one_dev) (void)hipFree(one_dev);
    if (one_host) (void)hipHostFree(one_host);
    if (one_stream) (void)hipStreamDestroy(one_stream);
    one_dev = one_host = nullptr;
    one_stream = nullptr;
    ready = false;
  }
};

std::mutex g_mu;
bool g_inited = false;
int g_device = -1;
MechState g_mech[3];

std::string mech_dir() {
  if (const char* e = std::getenv("MISTRA_MECH_DIR")) return e;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&mistra_chem_init), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t k = p.rfind('/');
    std::string dir = k == std::string::npos ? "." : p.substr(0, k);
    return dir + "/../mech";
  }
  return "mech";
}

int default_nt(int mech) {
  const char* names[3] = {"MISTRA_NT_GAS", "MISTRA_NT_AER", "MISTRA_NT_TOT"};
  if (const char* e = std::getenv(names[mech])) return std::atoi(e);
  return mech == MISTRA_MECH_GAS ? 128 : 512;
}

template <class MT>
bool traits_match(const MechTables& t, int n_jnz, int tail_regs, bool scale_pass, const DenseTail& dense) {
  return dense.nd == MT::DENSE_ND && dense.kb == MT::DENSE_KB && tail_regs == MT::TAIL_REGS && scale_pass == MT::SCALE_PASS && t.nvar == MT::NVAR && t.nfix == MT::NFIX && t.nreact == MT::NREACT && t.nnz == MT::NNZ && t.nb == MT::NB &&
         t.nconst == MT::NCONST && n_jnz == MT::NJNZ;
}

int setup_mech(int mech) {
  MechState& S = g_mech[mech];
  std::string err;
  if (!S.tab.load(mech_dir() + "/" + kMechName[mech] + ".mech", &err)) return fail(err);
  S.nt = default_nt(mech);
  const bool nt_ok = (mech == MISTRA_MECH_GAS && S.nt == 128) || (mech == MISTRA_MECH_AER && S.nt == 512) ||
                     (mech == MISTRA_MECH_TOT && S.nt == 512);
  if (!nt_ok) return fail(std::string("no kernel instantiated for workgroup size ") + std::to_string(S.nt) + " of " + kMechName[mech]);
  // LDS byte address of the A/B product array for this <mechanism, workgroup size> (the gather-sum tables hold addresses)
  const uint32_t ab_base = 8u * (uint32_t)(mech == MISTRA_MECH_GAS   ? LdsLayout<GasTraits, 128>::AB
                                           : mech == MISTRA_MECH_AER ? LdsLayout<AerTraits, 512>::AB
                                                                     : LdsLayout<TotTraits, 512>::AB);
  KernelSchedule K;
  try {
    const int max_temps = mech == MISTRA_MECH_GAS ? GasTraits::MAX_TEMPS : mech == MISTRA_MECH_AER ? AerTraits::MAX_TEMPS : TotTraits::MAX_TEMPS;
    const DenseConfig dc = dense_config(S.tab);
    K = build_kernel_schedule(S.tab, S.nt, ab_base, max_temps, dc.nd, dc.kb);
  } catch (const std::exception& ex) {
    return fail(std::string("schedule compiler: ") + ex.what());
  }
  bool ok = mech == MISTRA_MECH_GAS   ? traits_match<GasTraits>(S.tab, K.n_jnz, K.tail.regs, K.lu_scale.nslots > 0, K.dense)
            : mech == MISTRA_MECH_AER ? traits_match<AerTraits>(S.tab, K.n_jnz, K.tail.regs, K.lu_scale.nslots > 0, K.dense)
                                      : traits_match<TotTraits>(S.tab, K.n_jnz, K.tail.regs, K.lu_scale.nslots > 0, K.dense);
  if (!ok) return fail(std::string(kMechName[mech]) + ": mechanism table does not match the compiled kernel sizes");
  S.text = std::string(kMechName[mech]) + ": " + describe(K);
  S.n_temps = K.n_temps;
  HIP_TRY(S.consts.upload(S.tab.consts));
  HIP_TRY(S.fun_fac.upload(K.fun_fac));
  HIP_TRY(S.jac_fac.upload(K.jac_fac));
  HIP_TRY(S.jvs_pos.upload(K.jvs_pos));
  HIP_TRY(S.zero_pos.upload(K.zero_pos));
  HIP_TRY(S.diag_pos.upload(K.diag_pos));
  HIP_TRY(S.vdot.upload(K.vdot));
  HIP_TRY(S.jvs.upload(K.jvs));
  HIP_TRY(S.lu.upload(K.lu));
  HIP_TRY(S.solve_head_fwd.upload(K.solve_head_fwd));
  HIP_TRY(S.solve_head_bwd.upload(K.solve_head_bwd));
  HIP_TRY(S.tail_fwd.upload(K.tail.fwd));
  HIP_TRY(S.tail_bwd.upload(K.tail.bwd));
  HIP_TRY(S.lu_scale.upload(K.lu_scale.recs));
  HIP_TRY(S.dense_rows.upload(K.dense.row_info));
  S.lu_scale_slots = K.lu_scale.nslots;
  S.ready = true;
  return 0;
}

int launch(int mech, const KernelArgs& a, hipStream_t stream) {
  hipError_t e = hipErrorInvalidValue;
  if (mech == MISTRA_MECH_GAS) e = launch_ros3<GasTraits, 128>(a, stream);
  else if (mech == MISTRA_MECH_AER) e = launch_ros3<AerTraits, 512>(a, stream);
  else e = launch_ros3<TotTraits, 512>(a, stream);
  if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
  return 0;
}

KernelArgs make_args(const MechState& S, int ncell, const double* var_in, const double* fix, const double* rconst, double tin,
                     double tout, double* var_out, int32_t* ierr, int32_t* stats, double* th) {
  KernelArgs a;
  a.var_in = var_in; a.fix = fix; a.rconst = rconst; a.var_out = var_out; a.ierr = ierr; a.stats = stats;
  a.texit_hexit = th; a.prof = nullptr; a.n_temps = S.n_temps; a.tin = tin; a.tout = tout; a.ncell = ncell;
  a.consts = S.consts.p; a.fun_fac = S.fun_fac.p; a.jac_fac = S.jac_fac.p; a.jvs_pos = S.jvs_pos.p;
  a.zero_pos = S.zero_pos.p; a.diag_pos = S.diag_pos.p;
  a.vdot = S.vdot.dev(); a.jvs = S.jvs.dev(); a.lu = S.lu.dev();
  if (const char* cut = std::getenv("MISTRA_DIAG_LU_ROUNDS"))      // timing diagnostic only (tools/profile_lu_rounds.py): results are garbage
    a.lu.nrounds = std::max(1, std::min(a.lu.nrounds, std::atoi(cut)));
  a.solve_head_fwd = S.solve_head_fwd.dev(); a.solve_head_bwd = S.solve_head_bwd.dev();
  a.tail = TailDev{S.tail_fwd.p, S.tail_bwd.p};
  a.lu_scale = ScaleDev{S.lu_scale.p, S.lu_scale_slots, S.lu_scale_slots + VM_LOOKAHEAD_ROWS};
  a.dense = DenseDev{S.dense_rows.p};
  return a;
}

int check_call(int mech, int ncell) {
  if (mech < 0 || mech > 2) return fail("unknown mechanism id");
  if (ncell < 0) return fail("negative cell count");
  if (!g_inited || !g_mech[mech].ready) return fail("mistra_chem_init has not been called (or failed)");
  return 0;
}

}  // namespace

extern "C" {

const char* mistra_chem_last_error(void) { return g_err.c_str(); }

int mistra_chem_dims(int mech, int* nvar, int* nfix, int* nreact, int* lu_nonzero) {
  if (mech < 0 || mech > 2) return fail("unknown mechanism id");
  if (nvar) *nvar = kDims[mech][0];
  if (nfix) *nfix = kDims[mech][1];
  if (nreact) *nreact = kDims[mech][2];
  if (lu_nonzero) *lu_nonzero = kDims[mech][3];
  return 0;
}

int mistra_chem_init(int device) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_err.clear();
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return fail("no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= count) return fail("device index out of range");
  HIP_TRY(hipSetDevice(device));
  if (g_inited && g_device == device) return 0;
  for (auto& m : g_mech) m.release();
  g_device = device;
  for (int mech = 0; mech < 3; mech++)
    if (int rc = setup_mech(mech)) return rc;
  g_inited = true;
  return 0;
}

void mistra_chem_finalize(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (auto& m : g_mech) m.release();
  g_inited = false;
  g_device = -1;
}

const char* mistra_chem_describe(int mech) {
  if (mech < 0 || mech > 2 || !g_mech[mech].ready) return "";
  return g_mech[mech].text.c_str();
}

int mistra_chem_integrate_device(int mech, int ncell, const double* d_var_in, const double* d_fix, const double* d_rconst,
                                 double tin, double tout, double* d_var_out, int32_t* d_ierr, int32_t* d_stats,
                                 double* d_texit_hexit, void* hip_stream) {
  if (int rc = check_call(mech, ncell)) return rc;
  if (ncell == 0) return 0;
  if (!d_var_in || !d_fix || !d_rconst || !d_var_out || !d_ierr || !d_stats) return fail("null device pointer");
  KernelArgs a = make_args(g_mech[mech], ncell, d_var_in, d_fix, d_rconst, tin, tout, d_var_out, d_ierr, d_stats, d_texit_hexit);
  return launch(mech, a, static_cast<hipStream_t>(hip_stream));
}

int mistra_chem_integrate(int mech, int ncell, const double* var_in, const double* fix, const double* rconst, double tin,
                          double tout, double* var_out, int32_t* ierr, int32_t* stats) {
  if (int rc = check_call(mech, ncell)) return rc;
  if (ncell == 0) return 0;
  if (!var_in || !fix || !rconst || !var_out) return fail("null host pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  MechState& S = g_mech[mech];
  const size_t nv = (size_t)kDims[mech][0], nf = (size_t)kDims[mech][1], nr = (size_t)kDims[mech][2], nc = (size_t)ncell;
  HIP_TRY(S.s_var.reserve(nc * nv));
  HIP_TRY(S.s_fix.reserve(nc * nf));
  HIP_TRY(S.s_rct.reserve(nc * nr));
  HIP_TRY(S.s_ierr.reserve(nc));
  HIP_TRY(S.s_stats.reserve(nc * 8));
  HIP_TRY(hipMemcpy(S.s_var.p, var_in, nc * nv * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.s_fix.p, fix, nc * nf * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.s_rct.p, rconst, nc * nr * sizeof(double), hipMemcpyHostToDevice));
  KernelArgs a = make_args(S, ncell, S.s_var.p, S.s_fix.p, S.s_rct.p, tin, tout, S.s_var.p, S.s_ierr.p, S.s_stats.p, nullptr);
  // diagnostics: MISTRA_CHEM_PROFILE=1 prints where wave 0 of the workgroups spent its cycles (mean over the cells of the call)
  DevBuf<unsigned long long> prof;
  const bool profile = std::getenv("MISTRA_CHEM_PROFILE") != nullptr;
  if (profile) {
    HIP_TRY(prof.reserve(nc * kProfSlots));
    a.prof = prof.p;
  }
  if (int rc = launch(mech, a, nullptr)) return rc;
  HIP_TRY(hipDeviceSynchronize());
  if (profile) {
    std::vector<unsigned long long> h(nc * kProfSlots);
    HIP_TRY(hipMemcpy(h.data(), prof.p, nc * kProfSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[kProfSlots] = {0};
    for (size_t c = 0; c < nc; c++)
      for (int k = 0; k < kProfSlots; k++) sum[k] += (double)h[c * kProfSlots + k];
    const char* names[kProfSlots] = {"fun", "jac", "prepare", "lu", "solve(rest)", "norm", "other", "total", "solve_head_fwd", "solve_tail", "solve_head_bwd", "lu_scale", "lu_dense", "-", "-", "-"};
    std::fprintf(stderr, "[mistra_chem profile] %s, %zu cells, mean shader-clock ticks per cell:", kMechName[mech], nc);
    for (int k = 0; k < 13; k++) std::fprintf(stderr, " %s=%.0f (%.1f%%)", names[k], sum[k] / nc, 100.0 * sum[k] / sum[7]);
    std::fprintf(stderr, "\n");
    prof.release();
  }
  HIP_TRY(hipMemcpy(var_out, S.s_var.p, nc * nv * sizeof(double), hipMemcpyDeviceToHost));
  if (ierr) HIP_TRY(hipMemcpy(ierr, S.s_ierr.p, nc * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (stats) HIP_TRY(hipMemcpy(stats, S.s_stats.p, nc * 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return 0;
}

int mistra_chem_integrate_common(int mech, void* gdata, double* tin, double* tout) {
  if (!g_inited) {   // the Fortran caller has no init hook: first use selects the device (env MISTRA_CHEM_DEVICE, default 0)
    const char* dev = std::getenv("MISTRA_CHEM_DEVICE");
    if (int rc = mistra_chem_init(dev ? std::atoi(dev) : 0)) return rc;
  }
  if (int rc = check_call(mech, 1)) return rc;
  if (!gdata || !tin || !tout) return fail("null pointer");
  const int nv = kDims[mech][0], nf = kDims[mech][1], nr = kDims[mech][2];
  double* c = static_cast<double*>(gdata);          // C(NSPEC) = VAR | FIX
  double* rconst = c + nv + nf;                     // RCONST(NREACT)
  double* atol = rconst + nr + 2;                   // after TIME, DT
  double* rtol = atol + nv;
  double* stepmin = rtol + nv;
  for (int i = 0; i < nv; i++) { rtol[i] = 1.0e-3; atol[i] = 1.0e-25; }   // INTEGRATE_x, gas.f:745-746
  int32_t ierr = 0;
  MechState& S = g_mech[mech];
  {
    // One call = one cell: what costs here is synchronisation, not bytes.  /GDATA_x/ holds C and RCONST back to back, so
    // the inputs go up in ONE copy from a pinned mirror and everything the kernel writes comes back in ONE, on a private
    // stream with a single wait (seven blocking calls before: 340 us per gas call, of which the kernel is a fraction).
    std::lock_guard<std::mutex> lock(g_mu);
    const size_t n_in = (size_t)(nv + nf + nr), n_out = (size_t)nv + 2 + 5;       // out tail: 9 int32 in 5 doubles
    if (!S.one_dev) {
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&S.one_dev), (n_in + n_out) * sizeof(double)));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S.one_host), (n_in + n_out) * sizeof(double), hipHostMallocDefault));
      HIP_TRY(hipStreamCreateWithFlags(&S.one_stream, hipStreamNonBlocking));
    }
    std::memcpy(S.one_host, c, n_in * sizeof(double));
    HIP_TRY(hipMemcpyAsync(S.one_dev, S.one_host, n_in * sizeof(double), hipMemcpyHostToDevice, S.one_stream));
    double* d_out = S.one_dev + n_in;
    int32_t* d_stats = reinterpret_cast<int32_t*>(d_out + nv + 2);
    KernelArgs a = make_args(S, 1, S.one_dev, S.one_dev + nv, S.one_dev + nv + nf, *tin, *tout, d_out, d_stats + 8, d_stats, d_out + nv);
    if (int rc = launch(mech, a, S.one_stream)) return rc;
    double* h_out = S.one_host + n_in;
    HIP_TRY(hipMemcpyAsync(h_out, d_out, n_out * sizeof(double), hipMemcpyDeviceToHost, S.one_stream));
    HIP_TRY(hipStreamSynchronize(S.one_stream));
    std::memcpy(c, h_out, (size_t)nv * sizeof(double));
    std::memcpy(&ierr, reinterpret_cast<const int32_t*>(h_out + nv + 2) + 8, sizeof ierr);
    if (ierr < 0)   // the reference prints and continues (gas.f:764-767)
      std::printf(" Rosenbrock: Unsucessful step at T=%g (IERR=%d)\n", *tin, ierr);
    *tin = h_out[nv];          // TIN = RPAR(11), exit time
    *stepmin = h_out[nv + 1];  // STEPMIN = RPAR(12), last step
  }
  return 0;
}

}  // extern "C"
