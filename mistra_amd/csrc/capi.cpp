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
#include <thread>
#include <vector>

#include "kernel_args.hpp"
#include "mech_tables.hpp"
#include "pack.hpp"
#include "rates.hpp"
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

}  // namespace

bool mistra::RatesTable::load(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { if (err) *err = "cannot open " + path; return false; }
  int32_t h[8];
  bool ok = std::fread(h, sizeof h, 1, f) == 1 && h[0] == 0x5441524B && h[1] == 2;
  if (ok) {
    nreact = h[2]; nenv = h[3];
    consts.resize((size_t)h[4]); offs.resize((size_t)nreact + 1); words.resize((size_t)h[5]); fslot.resize((size_t)h[6]);
    ok = h[6] == 50 && std::fread(consts.data(), 8, consts.size(), f) == consts.size() && std::fread(offs.data(), 4, offs.size(), f) == offs.size() &&
         std::fread(words.data(), 4, words.size(), f) == words.size() && std::fread(fslot.data(), 4, fslot.size(), f) == fslot.size();
  }
  std::fclose(f);
  if (!ok && err) *err = path + ": not a rate table";
  return ok;
}

bool mistra::StcoeffTable::load(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { if (err) *err = "cannot open " + path; return false; }
  bool ok = true;
  for (int i = 0; ok && i < 4; i++) {
    RatesTable& T = v[i];
    int32_t h[8];
    ok = std::fread(h, sizeof h, 1, f) == 1 && h[0] == 0x5441524B && h[1] == 2 && h[2] > 0 && h[2] <= 4096 && h[3] >= 4 && h[3] <= 64 && h[4] >= 0 &&
         h[4] <= 1 << 20 && h[5] >= 0 && h[5] <= 1 << 20 && h[6] == 0 && h[7] == i;
    if (!ok) break;
    T.nreact = h[2]; T.nenv = h[3];
    T.consts.resize((size_t)h[4]); T.offs.resize((size_t)T.nreact + 1); T.words.resize((size_t)h[5]); T.fslot.clear();
    ok = (T.consts.empty() || std::fread(T.consts.data(), 8, T.consts.size(), f) == T.consts.size()) &&
         std::fread(T.offs.data(), 4, T.offs.size(), f) == T.offs.size() && (T.words.empty() || std::fread(T.words.data(), 4, T.words.size(), f) == T.words.size());
    // every index the evaluator follows is checked here: program bounds, literal and input slots, the functions st_coeff_x calls
    ok = ok && T.offs[0] == 0 && T.offs[(size_t)T.nreact] == (int32_t)T.words.size() && (i == 0 || (T.nreact == v[0].nreact && T.nenv == v[0].nenv));
    for (int r = 0; ok && r < T.nreact; r++) ok = T.offs[(size_t)r] < T.offs[(size_t)r + 1];
    for (size_t w = 0; ok && w < T.words.size(); w++) {
      const int op = T.words[w] & 0xFF, arg = T.words[w] >> 8;
      ok = op == 0 ? (arg >= 0 && arg < (int)T.consts.size()) : op == 1 ? (arg >= 0 && arg < T.nenv) : op <= 6 ? arg == 0 : (op == 7 && arg >= 26 && arg <= 28);
    }
  }
  std::fclose(f);
  if (!ok && err) *err = path + ": not a table of accommodation coefficients";
  return ok;
}

bool mistra::PackTable::load(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { if (err) *err = "cannot open " + path; return false; }
  int32_t h[16];
  bool ok = std::fread(h, sizeof h, 1, f) == 1 && h[0] == 0x4B41504B && h[1] == 1;
  // header counts are bounded before anything is sized by them (a stale or corrupt file must not throw out of an extern "C" entry)
  for (int i = 2; ok && i < 16; i++) ok = h[i] >= 0 && h[i] <= (1 << 20);
  ok = ok && h[2] > 0 && h[3] >= 0 && h[4] > 0 && h[5] > 0 && h[6] >= 1 && h[6] <= 8 && h[7] <= 1 && h[11] <= kBudSlots;
  auto rd = [&](std::vector<int32_t>& v, size_t n) { v.resize(n); return n == 0 || std::fread(v.data(), 4, n, f) == n; };
  if (ok) {
    nvar = h[2]; nfix = h[3]; j2 = h[4]; j6 = h[5]; nkc = h[6]; preclamp = h[7];
    ok = rd(pack, (size_t)h[8] * 4) && rd(fix, (size_t)h[9] * 3) && rd(unpack, (size_t)h[10] * 4) && rd(slot_id, (size_t)h[11]) &&
         rd(slot_first, (size_t)h[11] + 1) && rd(terms, (size_t)h[12] * 3) && rd(term_words, (size_t)h[13]) && rd(acc, (size_t)h[14] * 2) &&
         rd(envc, (size_t)h[15] * 2);
  }
  std::fclose(f);
  if (ok) {      // every index the kernels of pack.hip follow (reaction numbers and env slots: setup_mech, which knows NREACT and the env size)
    const int nspec = nvar + nfix, nl = j2 * nkc, ni = j6 * nkc, nterms = (int)terms.size() / 3, nwords = (int)term_words.size();
    for (int i = 0; ok && i < n_pack(); i++) {
      const int32_t* e = &pack[(size_t)4 * i];
      ok = e[0] >= 0 && e[0] < nspec && (e[1] == 0 || e[1] == 1) && e[2] >= 0 && e[2] < (e[1] == 0 ? nl : ni) && (e[3] == 0 || e[3] == 1);
    }
    for (int i = 0; ok && i < n_fix(); i++) {
      const int32_t* e = &fix[(size_t)3 * i];
      ok = e[0] >= 0 && e[0] < nspec && e[1] >= 0 && e[1] <= 3 && e[2] >= 0 && e[2] < 4;
    }
    for (int i = 0; ok && i < n_unpack(); i++) {
      const int32_t* e = &unpack[(size_t)4 * i];
      ok = (e[0] == 0 || e[0] == 1) && e[1] >= 0 && e[1] < (e[0] == 0 ? nl : ni) && e[2] >= 0 && e[2] < nvar && (e[3] == 0 || e[3] == 1);
    }
    for (int i = 0; ok && i < n_slots(); i++)
      ok = slot_id[(size_t)i] >= 1 && slot_id[(size_t)i] <= kBudSlots && slot_first[(size_t)i] >= 0 && slot_first[(size_t)i] <= slot_first[(size_t)i + 1] &&
           slot_first[(size_t)i + 1] <= nterms;
    for (int q = 0; ok && q < nterms; q++) {
      const int w0 = terms[(size_t)3 * q + 2], w1 = q + 1 < nterms ? terms[(size_t)3 * (q + 1) + 2] : nwords;
      ok = terms[(size_t)3 * q + 1] >= 0 && w0 >= 0 && w0 <= w1 && w1 <= nwords;
    }
    for (int32_t w : term_words) ok = ok && w >= 0 && w < nvar;      // (the budget kernel reads V[] only)
    for (size_t a = 0; ok && a < acc.size() / 2; a++) ok = acc[2 * a] >= 1 && acc[2 * a] <= acc[2 * a + 1] && acc[2 * a + 1] <= kBudSlots;
    for (int i = 0; ok && i < n_envc(); i++) ok = envc[(size_t)2 * i] >= 0 && envc[(size_t)2 * i + 1] >= 0 && envc[(size_t)2 * i + 1] < nspec;
  }
  if (!ok && err) *err = path + ": not a hand-over table";
  return ok;
}

bool mistra::KmtTable::load(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) { if (err) *err = "cannot open " + path; return false; }
  bool ok = std::fscanf(f, "%d %d %d %d", &nx, &nka, &nkt, &nkc) == 4 && nx > 0 && nx <= 64;
  if (ok) {
    lex.resize((size_t)nx);
    for (int i = 0; ok && i < nx; i++) ok = std::fscanf(f, "%d", &lex[(size_t)i]) == 1;
  }
  std::fclose(f);
  if (!ok && err) *err = path + ": not a species list of fast_k_mt";
  return ok;
}

bool mistra::LiqTable::load(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { if (err) *err = "cannot open " + path; return false; }
  int32_t h[8];
  double d[3];
  bool ok = std::fread(h, sizeof h, 1, f) == 1 && h[0] == 0x5451494C && h[1] == 1 && h[2] > 0 && h[2] <= 4096 && h[3] >= 0 && h[3] <= h[2] &&
            h[4] >= 0 && h[4] <= h[2] && h[5] >= 1 && h[5] <= 8 && h[6] >= 0 && h[6] <= 1 << 20 && std::fread(d, sizeof d, 1, f) == 1;
  std::vector<int32_t> hj, hk, ej;
  std::vector<double> a0, b0;
  auto rdi = [&](std::vector<int32_t>& v, size_t n) { v.resize(n); return n == 0 || std::fread(v.data(), 4, n, f) == n; };
  auto rdd = [&](std::vector<double>& v, size_t n) { v.resize(n); return n == 0 || std::fread(v.data(), 8, n, f) == n; };
  if (ok) {
    nspec = h[2]; nh = h[3]; ne = h[4]; nkc_eq = h[5]; nfac = h[6];
    henry_tref = d[0]; henry_fct = d[1]; equil_tref = d[2];
    ok = rdi(hj, (size_t)nh) && rdi(hk, (size_t)nh) && rdd(a0, (size_t)nh) && rdd(b0, (size_t)nh) && rdi(ej, (size_t)ne) && rdi(foff, (size_t)ne + 1) &&
         rdi(boff, (size_t)ne + 1) && rdi(fkind, (size_t)nfac) && rdi(farg, (size_t)nfac) && rdd(fa, (size_t)nfac) && rdd(fb, (size_t)nfac);
  }
  std::fclose(f);
  if (ok) {      // dense per-species forms; every index the kernels follow is checked here
    h_kind.assign((size_t)nspec, -1); h_a0.assign((size_t)nspec, 0.0); h_b0.assign((size_t)nspec, 0.0); e_of.assign((size_t)nspec, -1);
    for (int i = 0; ok && i < nh; i++) {
      ok = hj[(size_t)i] >= 1 && hj[(size_t)i] <= nspec && (hk[(size_t)i] == 0 || hk[(size_t)i] == 1);
      if (ok) { const size_t j = (size_t)hj[(size_t)i] - 1; h_kind[j] = hk[(size_t)i]; h_a0[j] = a0[(size_t)i]; h_b0[j] = b0[(size_t)i]; }
    }
    for (int i = 0; ok && i < ne; i++) {
      ok = ej[(size_t)i] >= 1 && ej[(size_t)i] <= nspec && foff[(size_t)i] >= 0 && foff[(size_t)i] < boff[(size_t)i] && boff[(size_t)i] < foff[(size_t)i + 1] &&
           foff[(size_t)i + 1] <= nfac;
      if (ok) e_of[(size_t)ej[(size_t)i] - 1] = i;
    }
    for (int i = 0; ok && i < nfac; i++) ok = fkind[(size_t)i] >= 0 && fkind[(size_t)i] <= 3 && (fkind[(size_t)i] != 3 || (farg[(size_t)i] >= 1 && farg[(size_t)i] <= 4096));
  }
  if (!ok && err) *err = path + ": not a table of Henry / equilibrium constants";
  return ok;
}

bool mistra::VmeanTable::load(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { if (err) *err = "cannot open " + path; return false; }
  int32_t h[8];
  bool ok = std::fread(h, sizeof h, 1, f) == 1 && h[0] == 0x544E4D56 && h[1] == 1 && h[2] > 0 && h[2] <= 4096 && h[3] >= 0 && h[3] <= h[2] &&
            std::fread(&coef, sizeof coef, 1, f) == 1 && coef > 0.0;
  std::vector<int32_t> j;
  std::vector<double> m;
  if (ok) {
    nspec = h[2];
    j.resize((size_t)h[3]); m.resize((size_t)h[3]);
    ok = (h[3] == 0 || std::fread(j.data(), 4, j.size(), f) == j.size()) && (h[3] == 0 || std::fread(m.data(), 8, m.size(), f) == m.size());
  }
  std::fclose(f);
  if (ok) {      // dense per-species form; every index is checked here
    mass.assign((size_t)nspec, 0.0);
    for (size_t i = 0; ok && i < j.size(); i++) {
      ok = j[i] >= 1 && j[i] <= nspec && m[i] > 0.0;
      if (ok) mass[(size_t)j[i] - 1] = m[i];
    }
  }
  if (!ok && err) *err = path + ": not a table of molar masses of v_mean";
  return ok;
}

namespace {

struct VmBufs {
  DevBuf<uint32_t> wave_base, recs;
  DevBuf<uint16_t> blk_n;
  int nrounds = 0;
  hipError_t upload(const VmProgram& P) {
    nrounds = P.nbarriers;      // what the executor counts down: the rounds that end in a barrier
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
  DevBuf<uint16_t> schur_cells;
  // Update_RCONST_x on the device (rates.hip): present for the mechanisms whose table and rate-law functions exist
  bool rates_ready = false;
  int rates_nenv = 0;
  DevBuf<double> rates_consts, s_env;
  DevBuf<int32_t> rates_offs, rates_words, rates_fslot;
  int lu_scale_slots = 0;
  // the hand-over halves of x_drive on the device (pack.hip): tables of the mechanism + the model's species maps
  bool pack_ready = false, maps_ready = false;
  PackTable pack_tab;
  DevBuf<int32_t> pk_pack, pk_fix, pk_unpack, pk_slot_id, pk_slot_first, pk_terms, pk_words, pk_acc, pk_envc, pk_aptr, pk_afac;
  DevBuf<int32_t> map_gas_m2k, map_gas_k2m, map_rad_m2k, map_rad_k2m;
  int map_j1 = 0, map_j5 = 0;
  DevBuf<double> d_env, d_rct;      // scratch of mistra_chem_drive_device (grow-only)
  // mistra_chem_drive (host buffers): ONE device block and ONE pinned host mirror hold everything that crosses PCIe for a batch of layers
  // (model slabs, scal, env, budgets in; slabs, budgets, exit data out), a private stream carries the two copies and the kernel chain
  double* drv_dev = nullptr;
  double* drv_host = nullptr;
  size_t drv_cap = 0;               // doubles
  hipStream_t drv_stream = nullptr;
  struct PendingDrive {             // a column step issued by mistra_chem_drive_begin and not yet fetched by mistra_chem_drive_end
    bool active = false;
    std::vector<int32_t> layer, level;
    double *s1 = nullptr, *s3 = nullptr, *sl1 = nullptr, *sion1 = nullptr, *bg = nullptr, *bgs = nullptr, *t_h = nullptr, *c_packed = nullptr;
    int32_t *ierr = nullptr, *stats = nullptr;
    int nrxn = 0;
    size_t o_s1 = 0, o_s3 = 0, o_sl1 = 0, o_si = 0, o_bg = 0, o_bgs = 0, o_th = 0, o_hl = 0, o_int = 0, o_cp = 0;
  } pend;
  // fast_k_mt_a / fast_k_mt_t (aer, tot): the exchanged species and the size-axis limits of the last call
  bool kmt_ready = false;
  KmtTable kmt_tab;
  DevBuf<int32_t> kmt_lex;
  // henry_x / equil_co_x (aer, tot)
  bool liq_ready = false;
  LiqTable liq_tab;
  DevBuf<int32_t> lq_hkind, lq_eof, lq_foff, lq_boff, lq_fkind, lq_farg;
  DevBuf<double> lq_ha0, lq_hb0, lq_fa, lq_fb;
  // st_coeff_x (aer, tot): one table per setting of the two namelist switches
  bool stc_ready = false;
  StcoeffTable stc_tab;
  DevBuf<double> stc_consts[4];
  DevBuf<int32_t> stc_offs[4], stc_words[4];
  // v_mean_x (aer, tot)
  bool vmean_ready = false;
  VmeanTable vmean_tab;
  DevBuf<double> vm_mass;
  // staging for the host-buffer entry point (grow-only)
  DevBuf<double> s_var, s_fix, s_rct, s_out, s_th;
  DevBuf<int32_t> s_ierr, s_stats, s_sing;
  // where the zero-pivot rows of the LAST host-buffer call of this slot are (mistra_chem_singular_rows): cells [sing_start,
  // sing_start + sing_count) of the caller's batch in s_sing, or the one cell of the COMMON-block call in one_sing
  size_t sing_start = 0, sing_count = 0;
  bool sing_one = false;
  int32_t one_sing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // one-cell calls (the Fortran shim): one contiguous device block in, one out, pinned host mirrors, a private stream
  double* one_dev = nullptr;      // [C(NSPEC) | RCONST(NREACT)]  then  [VAR out | Texit Hexit | 8 stats + ierr as int32]
  double* one_host = nullptr;
  hipStream_t one_stream = nullptr;
  void release() {
    consts.release(); fun_fac.release(); jac_fac.release(); jvs_pos.release(); zero_pos.release(); diag_pos.release();
    vdot.release(); jvs.release(); lu.release(); solve_head_fwd.release(); solve_head_bwd.release();
    tail_fwd.release(); tail_bwd.release(); lu_scale.release();
    pk_pack.release(); pk_fix.release(); pk_unpack.release(); pk_slot_id.release(); pk_slot_first.release(); pk_terms.release(); pk_words.release();
    pk_acc.release(); pk_envc.release(); pk_aptr.release(); pk_afac.release(); map_gas_m2k.release(); map_gas_k2m.release(); map_rad_m2k.release();
    map_rad_k2m.release(); d_env.release(); d_rct.release(); pack_ready = maps_ready = false;
    kmt_lex.release(); kmt_ready = false;
    lq_hkind.release(); lq_eof.release(); lq_foff.release(); lq_boff.release(); lq_fkind.release(); lq_farg.release(); lq_ha0.release(); lq_hb0.release();
    lq_fa.release(); lq_fb.release(); liq_ready = false;
    vm_mass.release(); vmean_ready = false;
    for (int i = 0; i < 4; i++) { stc_consts[i].release(); stc_offs[i].release(); stc_words[i].release(); }
    stc_ready = false;
    dense_rows.release(); schur_cells.release(); rates_consts.release(); rates_offs.release(); rates_words.release(); rates_fslot.release(); s_env.release(); rates_ready = false;
    s_var.release(); s_fix.release(); s_rct.release(); s_out.release(); s_th.release(); s_ierr.release(); s_stats.release(); s_sing.release();
    sing_count = 0; sing_one = false;
    if (drv_dev) (void)hipFree(drv_dev);
    if (drv_host) (void)hipHostFree(drv_host);
    if (drv_stream) (void)hipStreamDestroy(drv_stream);
    drv_dev = drv_host = nullptr; drv_cap = 0; drv_stream = nullptr;
    if (one_dev) (void)hipFree(one_dev);
    if (one_host) (void)hipHostFree(one_host);
    if (one_stream) (void)hipStreamDestroy(one_stream);
    one_dev = one_host = nullptr;
    one_stream = nullptr;
    pend = PendingDrive{};      // (a step issued and never fetched dies with its buffers)
    ready = false;
  }
};

// One DeviceState per GPU the library was initialised on (mistra_chem_init: one; mistra_chem_init_devices: several).
// Slot 0 is the primary device: the one-cell Fortran entry points and mistra_chem_describe use it.
struct DeviceState {
  int id = -1;
  MechState mech[3];
  bool lds_configured[3] = {false, false, false};     // hipFuncAttributeMaxDynamicSharedMemorySize is per device
  void release() {
    if (id >= 0) (void)hipSetDevice(id);
    for (auto& m : mech) m.release();
    for (bool& c : lds_configured) c = false;
    id = -1;
  }
};

std::mutex g_mu;
bool g_inited = false;
int g_max_steps = 100000;      // Max_no_steps (gas.f:1042); only mistra_chem_debug_set_max_steps changes it
std::vector<DeviceState> g_devs;

// host-buffer entries of the liq_parm kernels: arenas on the primary device (see DevBlock below)
struct LiqStage {
  char *dev = nullptr, *host = nullptr;
  size_t cap = 0;
  hipStream_t st = nullptr;
  std::vector<std::pair<const char*, size_t>> pinned;      // caller ranges registered by mistra_chem_pin_host
  hipError_t ensure(size_t bytes) {
    if (!st)
      if (hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) return e;
    if (bytes <= cap) return hipSuccess;
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    dev = host = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4;
    if (hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev), want)) return e;
    if (hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&host), want, hipHostMallocDefault)) return e;
    cap = want;
    return hipSuccess;
  }
  bool is_pinned(const void* p, size_t n) const {
    const char* c = static_cast<const char*>(p);
    for (const auto& r : pinned)
      if (c >= r.first && c + n <= r.first + r.second) return true;
    return false;
  }
  void release() {
    for (const auto& r : pinned) (void)hipHostUnregister(const_cast<char*>(r.first));
    pinned.clear();
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    if (st) (void)hipStreamDestroy(st);
    dev = host = nullptr; cap = 0; st = nullptr;
  }
};
LiqStage g_liq;

DeviceState* device_slot(int hip_device) {
  for (auto& d : g_devs)
    if (d.id == hip_device) return &d;
  return nullptr;
}

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

int default_nt(int mech) { return mech == MISTRA_MECH_GAS ? kGasNT : mech == MISTRA_MECH_AER ? kAerNT : kTotNT; }      // the workgroup sizes ros3_kernel.hip instantiates

template <class MT>
bool traits_match(const MechTables& t, int n_jnz, int tail_regs, bool scale_pass, const DenseTail& dense) {
  return dense.nd == MT::DENSE_ND && dense.kb == MT::DENSE_KB && tail_regs == MT::TAIL_REGS && scale_pass == MT::SCALE_PASS && t.nvar == MT::NVAR && t.nfix == MT::NFIX && t.nreact == MT::NREACT && t.nnz == MT::NNZ && t.nb == MT::NB &&
         t.nconst == MT::NCONST && n_jnz == MT::NJNZ;
}

int setup_mech(DeviceState& D, int mech) {
  MechState& S = D.mech[mech];
  std::string err;
  if (!S.tab.load(mech_dir() + "/" + kMechName[mech] + ".mech", &err)) return fail(err);
  S.nt = default_nt(mech);
  const bool nt_ok = (mech == MISTRA_MECH_GAS && S.nt == kGasNT) || (mech == MISTRA_MECH_AER && S.nt == kAerNT) ||
                     (mech == MISTRA_MECH_TOT && S.nt == kTotNT);
  if (!nt_ok) return fail(std::string("no kernel instantiated for workgroup size ") + std::to_string(S.nt) + " of " + kMechName[mech]);
  // LDS byte address of the A/B product array for this <mechanism, workgroup size> (the gather-sum tables hold addresses)
  const uint32_t ab_base = 8u * (uint32_t)(mech == MISTRA_MECH_GAS   ? LdsLayout<GasTraits, kGasNT>::AB
                                           : mech == MISTRA_MECH_AER ? LdsLayout<AerTraits, kAerNT>::AB
                                                                     : LdsLayout<TotTraits, kTotNT>::AB);
  KernelSchedule K;
  try {
    const int max_temps = mech == MISTRA_MECH_GAS ? GasTraits::MAX_TEMPS : mech == MISTRA_MECH_AER ? AerTraits::MAX_TEMPS : TotTraits::MAX_TEMPS;
    const DenseConfig dc = dense_config(S.tab);
    const uint32_t jb_base = 8u * (uint32_t)(mech == MISTRA_MECH_GAS   ? LdsLayout<GasTraits, kGasNT>::JB
                                             : mech == MISTRA_MECH_AER ? LdsLayout<AerTraits, kAerNT>::JB
                                                                       : LdsLayout<TotTraits, kTotNT>::JB);
    K = build_kernel_schedule(S.tab, S.nt, ab_base, max_temps, dc.nd, dc.kb, jb_base);
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
  HIP_TRY(S.schur_cells.upload(K.dense.schur_cells));
  {   // optional: the rate table (gas today)
    RatesTable T;
    std::string rerr;
    if (T.load(mech_dir() + "/" + kMechName[mech] + ".rates", &rerr)) {
      if (T.nreact != S.tab.nreact) return fail(std::string(kMechName[mech]) + ".rates does not belong to this mechanism");
      HIP_TRY(S.rates_consts.upload(T.consts));
      HIP_TRY(S.rates_offs.upload(T.offs));
      HIP_TRY(S.rates_words.upload(T.words));
      HIP_TRY(S.rates_fslot.upload(T.fslot));
      S.rates_nenv = T.nenv;
      S.rates_ready = true;
    }
  }
  {   // the drivers' hand-over tables
    std::string perr;
    if (S.pack_tab.load(mech_dir() + "/" + kMechName[mech] + ".pack", &perr)) {
      const PackTable& T = S.pack_tab;
      if (T.nvar != S.tab.nvar || T.nfix != S.tab.nfix) return fail(std::string(kMechName[mech]) + ".pack does not belong to this mechanism");
      for (size_t q = 0; q < T.terms.size() / 3; q++)
        if (T.terms[3 * q + 1] >= S.tab.nreact) return fail(std::string(kMechName[mech]) + ".pack: a budget term names a reaction the mechanism does not have");
      for (int i = 0; i < T.n_envc(); i++)
        if (S.rates_ready && T.envc[(size_t)2 * i] >= S.rates_nenv) return fail(std::string(kMechName[mech]) + ".pack: a concentration slot lies outside the rate evaluator's input");
      HIP_TRY(S.pk_pack.upload(T.pack)); HIP_TRY(S.pk_fix.upload(T.fix)); HIP_TRY(S.pk_unpack.upload(T.unpack));
      HIP_TRY(S.pk_slot_id.upload(T.slot_id)); HIP_TRY(S.pk_slot_first.upload(T.slot_first)); HIP_TRY(S.pk_terms.upload(T.terms));
      HIP_TRY(S.pk_words.upload(T.term_words)); HIP_TRY(S.pk_acc.upload(T.acc)); HIP_TRY(S.pk_envc.upload(T.envc));
      HIP_TRY(S.pk_aptr.upload(S.tab.a_ptr)); HIP_TRY(S.pk_afac.upload(S.tab.a_fac));
      S.pack_ready = true;
    }
  }
  {   // species list of the mass-transfer routines (aer, tot)
    std::string kerr;
    if (S.kmt_tab.load(mech_dir() + "/" + kMechName[mech] + ".kmt", &kerr)) {
      for (int32_t i : S.kmt_tab.lex)
        if (i < 1 || i > S.tab.nvar + S.tab.nfix) return fail(std::string(kMechName[mech]) + ".kmt does not belong to this mechanism");
      HIP_TRY(S.kmt_lex.upload(S.kmt_tab.lex));
      S.kmt_ready = true;
    }
  }
  {   // Henry and equilibrium constants (aer, tot)
    std::string lerr;
    LiqTable& T = S.liq_tab;
    if (T.load(mech_dir() + "/" + kMechName[mech] + ".liq", &lerr)) {
      if (T.nspec != S.tab.nvar + S.tab.nfix) return fail(std::string(kMechName[mech]) + ".liq does not belong to this mechanism");
      HIP_TRY(S.lq_hkind.upload(T.h_kind)); HIP_TRY(S.lq_eof.upload(T.e_of)); HIP_TRY(S.lq_foff.upload(T.foff)); HIP_TRY(S.lq_boff.upload(T.boff));
      HIP_TRY(S.lq_fkind.upload(T.fkind)); HIP_TRY(S.lq_farg.upload(T.farg)); HIP_TRY(S.lq_ha0.upload(T.h_a0)); HIP_TRY(S.lq_hb0.upload(T.h_b0));
      HIP_TRY(S.lq_fa.upload(T.fa)); HIP_TRY(S.lq_fb.upload(T.fb));
      S.liq_ready = true;
    }
  }
  {   // accommodation coefficients (aer, tot)
    std::string serr;
    StcoeffTable& T = S.stc_tab;
    if (T.load(mech_dir() + "/" + kMechName[mech] + ".stcoeff", &serr)) {
      if (T.v[0].nreact != S.tab.nvar + S.tab.nfix) return fail(std::string(kMechName[mech]) + ".stcoeff does not belong to this mechanism");
      for (int i = 0; i < 4; i++) {
        HIP_TRY(S.stc_consts[i].upload(T.v[i].consts)); HIP_TRY(S.stc_offs[i].upload(T.v[i].offs)); HIP_TRY(S.stc_words[i].upload(T.v[i].words));
      }
      S.stc_ready = true;
    }
  }
  {   // mean molecular speeds (aer, tot)
    std::string verr;
    VmeanTable& T = S.vmean_tab;
    if (T.load(mech_dir() + "/" + kMechName[mech] + ".vmean", &verr)) {
      if (T.nspec != S.tab.nvar + S.tab.nfix) return fail(std::string(kMechName[mech]) + ".vmean does not belong to this mechanism");
      HIP_TRY(S.vm_mass.upload(T.mass));
      S.vmean_ready = true;
    }
  }
  S.lu_scale_slots = K.lu_scale.nslots;
  S.ready = true;
  return 0;
}

int launch(DeviceState& D, int mech, const KernelArgs& a, hipStream_t stream) {
  hipError_t e = hipErrorInvalidValue;
  if (mech == MISTRA_MECH_GAS) e = launch_ros3<GasTraits, kGasNT>(a, stream, &D.lds_configured[mech]);
  else if (mech == MISTRA_MECH_AER) e = launch_ros3<AerTraits, kAerNT>(a, stream, &D.lds_configured[mech]);
  else e = launch_ros3<TotTraits, kTotNT>(a, stream, &D.lds_configured[mech]);
  if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
  return 0;
}

KernelArgs make_args(const MechState& S, int ncell, const double* var_in, const double* fix, const double* rconst, double tin,
                     double tout, double* var_out, int32_t* ierr, int32_t* stats, double* th) {
  KernelArgs a;
  a.var_in = var_in; a.fix = fix; a.rconst = rconst; a.var_out = var_out; a.ierr = ierr; a.stats = stats;
  a.texit_hexit = th; a.h_last = nullptr; a.hstart = nullptr; a.prof = nullptr; a.dump = nullptr; a.sing_rows = nullptr; a.n_temps = S.n_temps; a.max_steps = g_max_steps; a.tin = tin; a.tout = tout; a.ncell = ncell;
  a.consts = S.consts.p; a.fun_fac = S.fun_fac.p; a.jac_fac = S.jac_fac.p; a.jvs_pos = S.jvs_pos.p;
  a.zero_pos = S.zero_pos.p; a.diag_pos = S.diag_pos.p;
  a.vdot = S.vdot.dev(); a.jvs = S.jvs.dev(); a.lu = S.lu.dev();
#ifdef MISTRA_DIAG_ENV      // diagnostic builds only (tools/diag_dense.sh env): never in the product library
  if (const char* cut = std::getenv("MISTRA_DIAG_LU_ROUNDS"))      // timing diagnostic (tools/profile_lu_rounds.py): results are garbage
    a.lu.nrounds = std::max(1, std::min(a.lu.nrounds, std::atoi(cut)));
#endif
  a.solve_head_fwd = S.solve_head_fwd.dev(); a.solve_head_bwd = S.solve_head_bwd.dev();
  a.tail = TailDev{S.tail_fwd.p, S.tail_bwd.p};
  a.lu_scale = ScaleDev{S.lu_scale.p, S.lu_scale_slots, S.lu_scale_slots + VM_LOOKAHEAD_ROWS};
  a.dense = DenseDev{S.dense_rows.p, S.schur_cells.p};
  return a;
}

int check_call(int mech, int ncell) {
  if (mech < 0 || mech > 2) return fail("unknown mechanism id");
  if (ncell < 0) return fail("negative cell count");
  if (!g_inited || g_devs.empty() || !g_devs[0].mech[mech].ready) return fail("mistra_chem_init has not been called (or failed)");
  return 0;
}

int init_locked(int n, const int* ids) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return fail("no HIP device available (this library has no CPU path)");
  if (n <= 0 || n > 64) return fail("device count out of range");      // (more slots than GPUs is legal: a device may be listed more than once)
  std::vector<int> want((size_t)n);
  for (int i = 0; i < n; i++) {
    want[(size_t)i] = ids ? ids[i] : i;
    if (want[(size_t)i] < 0 || want[(size_t)i] >= count) return fail("device index out of range");
    // (a device may be listed more than once: every listing is a slot of its own — tables, staging buffers, host thread —
    //  so two blocks of one batch overlap their copies with each other's kernel on that GPU; include/mistra_chem.h)
  }
  bool same = g_inited && g_devs.size() == want.size();
  for (size_t i = 0; same && i < want.size(); i++) same = g_devs[i].id == want[i];
  if (same) {
    HIP_TRY(hipSetDevice(g_devs[0].id));
    return 0;
  }
  if (!g_devs.empty() && g_devs[0].id >= 0) (void)hipSetDevice(g_devs[0].id);
  g_liq.release();
  for (auto& d : g_devs) d.release();
  g_devs.clear();
  g_devs.resize(want.size());
  g_inited = false;
  for (size_t i = 0; i < want.size(); i++) {
    HIP_TRY(hipSetDevice(want[i]));
    g_devs[i].id = want[i];
    for (int mech = 0; mech < 3; mech++)
      if (int rc = setup_mech(g_devs[i], mech)) return rc;
  }
  HIP_TRY(hipSetDevice(g_devs[0].id));      // the calling thread ends up on the primary device
  g_inited = true;
  return 0;
}

// one device's share of a host-buffer call: upload, integrate, download (synchronous on that device)
// `env` != nullptr: the caller hands over the rate evaluator's inputs instead of the rate constants (rconst is then ignored): RCONST
// is made on the device and never crosses PCIe
int integrate_host_on(DeviceState& D, int mech, int ncell, const double* var_in, const double* fix, const double* rconst, double tin,
                      double tout, double* var_out, int32_t* ierr, int32_t* stats, double* t_h, size_t batch_start = 0,
                      const double* env = nullptr) {
  HIP_TRY(hipSetDevice(D.id));
  MechState& S = D.mech[mech];
  const size_t nv = (size_t)kDims[mech][0], nf = (size_t)kDims[mech][1], nr = (size_t)kDims[mech][2], nc = (size_t)ncell;
  HIP_TRY(S.s_var.reserve(nc * nv));
  HIP_TRY(S.s_fix.reserve(nc * nf));
  HIP_TRY(S.s_rct.reserve(nc * nr));
  HIP_TRY(S.s_ierr.reserve(nc));
  HIP_TRY(S.s_stats.reserve(nc * 8));
  HIP_TRY(S.s_sing.reserve(nc * 8));
  if (t_h) HIP_TRY(S.s_th.reserve(nc * 3));
  HIP_TRY(hipMemcpy(S.s_var.p, var_in, nc * nv * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.s_fix.p, fix, nc * nf * sizeof(double), hipMemcpyHostToDevice));
  if (env) {
    if (!S.rates_ready) return fail(std::string("no device rate table for the ") + kMechName[mech] + " mechanism");
    const size_t ne = (size_t)S.rates_nenv;
    HIP_TRY(S.s_env.reserve(nc * ne));
    HIP_TRY(hipMemcpy(S.s_env.p, env, nc * ne * sizeof(double), hipMemcpyHostToDevice));
    const RatesDev R{S.rates_consts.p, S.rates_offs.p, S.rates_words.p, S.rates_fslot.p, S.tab.nreact, S.rates_nenv};
    hipError_t e = launch_update_rconst(R, S.s_env.p, S.s_rct.p, ncell, nullptr);
    if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
  } else {
    HIP_TRY(hipMemcpy(S.s_rct.p, rconst, nc * nr * sizeof(double), hipMemcpyHostToDevice));
  }
  KernelArgs a = make_args(S, ncell, S.s_var.p, S.s_fix.p, S.s_rct.p, tin, tout, S.s_var.p, S.s_ierr.p, S.s_stats.p, t_h ? S.s_th.p : nullptr);
  if (t_h) a.h_last = S.s_th.p + 2 * nc;
  a.sing_rows = S.s_sing.p;      // stays on the device: fetched by mistra_chem_singular_rows, i.e. only when a cell reports Nsng > 0
  S.sing_start = batch_start; S.sing_count = nc; S.sing_one = false;
  // diagnostic builds only (-DMISTRA_DIAG_ENV, tools/diag_dense.sh env): MISTRA_CHEM_PROFILE=1 prints where wave 0 of the
  // workgroups spent its cycles (mean over the cells of the call).  The product library does not read the environment here.
  DevBuf<unsigned long long> prof;
#ifdef MISTRA_DIAG_ENV
  const bool profile = std::getenv("MISTRA_CHEM_PROFILE") != nullptr;
#else
  const bool profile = false;
#endif
  if (profile) {
    HIP_TRY(prof.reserve(nc * kProfSlots));
    a.prof = prof.p;
  }
  if (int rc = launch(D, mech, a, nullptr)) return rc;
  HIP_TRY(hipDeviceSynchronize());
  if (profile) {
    std::vector<unsigned long long> h(nc * kProfSlots);
    HIP_TRY(hipMemcpy(h.data(), prof.p, nc * kProfSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[kProfSlots] = {0};
    for (size_t c = 0; c < nc; c++)
      for (int k = 0; k < kProfSlots; k++) sum[k] += (double)h[c * kProfSlots + k];
    const char* names[kProfSlots] = {"fun", "jac", "prepare", "lu", "solve(rest)", "norm", "other", "total", "solve_head_fwd", "solve_tail", "solve_head_bwd", "lu_scale", "lu_dense", "jac_products", "jac_sums", "fun_products"};
    std::fprintf(stderr, "[mistra_chem profile] %s, %zu cells, mean shader-clock ticks per cell:", kMechName[mech], nc);
    for (int k = 0; k < kProfSlots; k++) std::fprintf(stderr, " %s=%.0f (%.1f%%)", names[k], sum[k] / nc, 100.0 * sum[k] / sum[7]);
    std::fprintf(stderr, "\n");
    prof.release();
  }
  HIP_TRY(hipMemcpy(var_out, S.s_var.p, nc * nv * sizeof(double), hipMemcpyDeviceToHost));
  if (ierr) HIP_TRY(hipMemcpy(ierr, S.s_ierr.p, nc * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (stats) HIP_TRY(hipMemcpy(stats, S.s_stats.p, nc * 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (t_h) {      // device: [ncell][2] (T, Hexit) then [ncell] (H)  ->  caller: [ncell][3]
    std::vector<double> h(nc * 3);
    HIP_TRY(hipMemcpy(h.data(), S.s_th.p, nc * 3 * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t c = 0; c < nc; c++) { t_h[3 * c] = h[2 * c]; t_h[3 * c + 1] = h[2 * c + 1]; t_h[3 * c + 2] = h[2 * nc + c]; }
  }
  return 0;
}

}  // namespace

static int lazy_init();

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
  return init_locked(1, &device);
}

int mistra_chem_init_devices(int n_devices, const int* device_ids) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_err.clear();
  return init_locked(n_devices, device_ids);
}

int mistra_chem_device_count(void) { return g_inited ? (int)g_devs.size() : 0; }

void mistra_chem_finalize(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_devs.empty() && g_devs[0].id >= 0) (void)hipSetDevice(g_devs[0].id);
  g_liq.release();
  for (auto& d : g_devs) d.release();
  g_devs.clear();
  g_inited = false;
}

const char* mistra_chem_describe(int mech) {
  if (mech < 0 || mech > 2 || !g_inited || g_devs.empty() || !g_devs[0].mech[mech].ready) return "";
  return g_devs[0].mech[mech].text.c_str();
}

int mistra_chem_integrate_device(int mech, int ncell, const double* d_var_in, const double* d_fix, const double* d_rconst,
                                 double tin, double tout, double* d_var_out, int32_t* d_ierr, int32_t* d_stats,
                                 double* d_texit_hexit, void* hip_stream) {
  return mistra_chem_integrate_device_hstart(mech, ncell, d_var_in, d_fix, d_rconst, tin, tout, d_var_out, d_ierr, d_stats, d_texit_hexit,
                                             nullptr, hip_stream);
}

int mistra_chem_integrate_device_hstart(int mech, int ncell, const double* d_var_in, const double* d_fix, const double* d_rconst,
                                        double tin, double tout, double* d_var_out, int32_t* d_ierr, int32_t* d_stats,
                                        double* d_texit_hexit, const double* d_hstart, void* hip_stream) {
  if (int rc = check_call(mech, ncell)) return rc;
  if (ncell == 0) return 0;
  if (!d_var_in || !d_fix || !d_rconst || !d_var_out || !d_ierr || !d_stats) return fail("null device pointer");
  // the buffers decide the device: the call runs where d_var_in lives, which must be one of the initialised devices
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_var_in) != hipSuccess) return fail("d_var_in is not a device pointer");
  DeviceState* D = device_slot(attr.device);
  if (!D) return fail("the buffers live on device " + std::to_string(attr.device) + ", which mistra_chem_init(_devices) did not set up");
  HIP_TRY(hipSetDevice(D->id));
  KernelArgs a = make_args(D->mech[mech], ncell, d_var_in, d_fix, d_rconst, tin, tout, d_var_out, d_ierr, d_stats, d_texit_hexit);
  a.hstart = d_hstart;
  return launch(*D, mech, a, static_cast<hipStream_t>(hip_stream));
}

int mistra_chem_rates_env_size(int mech) {
  if (mech < 0 || mech > 2 || !g_inited || g_devs.empty() || !g_devs[0].mech[mech].rates_ready) return 0;
  return g_devs[0].mech[mech].rates_nenv;
}

int mistra_chem_update_rconst_device(int mech, int ncell, const double* d_env, double* d_rconst, void* hip_stream) {
  if (int rc = check_call(mech, ncell)) return rc;
  if (ncell == 0) return 0;
  if (!d_env || !d_rconst) return fail("null device pointer");
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_env) != hipSuccess) return fail("d_env is not a device pointer");
  DeviceState* D = device_slot(attr.device);
  if (!D) return fail("the buffers live on a device mistra_chem_init(_devices) did not set up");
  MechState& S = D->mech[mech];
  if (!S.rates_ready) return fail(std::string("no device rate table for the ") + kMechName[mech] + " mechanism");
  HIP_TRY(hipSetDevice(D->id));
  const RatesDev R{S.rates_consts.p, S.rates_offs.p, S.rates_words.p, S.rates_fslot.p, S.tab.nreact, S.rates_nenv};
  hipError_t e = launch_update_rconst(R, d_env, d_rconst, ncell, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
  return 0;
}

int mistra_chem_update_rconst(int mech, int ncell, const double* env, double* rconst) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, ncell)) return rc;
  if (ncell == 0) return 0;
  if (!env || !rconst) return fail("null host pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  MechState& S = D.mech[mech];
  if (!S.rates_ready) return fail(std::string("no device rate table for the ") + kMechName[mech] + " mechanism");
  HIP_TRY(hipSetDevice(D.id));
  const size_t nc = (size_t)ncell, ne = (size_t)S.rates_nenv, nr = (size_t)S.tab.nreact;
  HIP_TRY(S.s_env.reserve(nc * ne));
  HIP_TRY(S.s_rct.reserve(nc * nr));
  HIP_TRY(hipMemcpy(S.s_env.p, env, nc * ne * sizeof(double), hipMemcpyHostToDevice));
  if (int rc = mistra_chem_update_rconst_device(mech, ncell, S.s_env.p, S.s_rct.p, nullptr)) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(rconst, S.s_rct.p, nc * nr * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// ---- the hand-over halves of x_drive (pack.hip)
static int pack_dev(int mech, const void* any_device_ptr, DeviceState** Dout, PackDev* P) {
  if (int rc = check_call(mech, 1)) return rc;
  hipPointerAttribute_t attr;
  if (!any_device_ptr || hipPointerGetAttributes(&attr, any_device_ptr) != hipSuccess) return fail("not a device pointer");
  DeviceState* D = device_slot(attr.device);
  if (!D) return fail("the buffers live on a device mistra_chem_init(_devices) did not set up");
  MechState& S = D->mech[mech];
  if (!S.pack_ready) return fail(std::string("no hand-over table for the ") + kMechName[mech] + " mechanism");
  if (!S.maps_ready) return fail("mistra_chem_set_species_maps has not been called for this mechanism");
  HIP_TRY(hipSetDevice(D->id));
  const PackTable& T = S.pack_tab;
  *P = PackDev{S.pk_pack.p, S.pk_fix.p, S.pk_unpack.p, S.pk_slot_id.p, S.pk_slot_first.p, S.pk_terms.p, S.pk_words.p, S.pk_acc.p, S.pk_envc.p,
               T.n_pack(), T.n_fix(), T.n_unpack(), T.n_slots(), (int)T.terms.size() / 3, (int)T.term_words.size(), (int)T.acc.size() / 2, T.n_envc(),
               T.nvar, T.nfix, S.tab.nreact, T.j2, T.j6, T.nkc, T.preclamp,
               S.map_gas_m2k.p, S.map_gas_k2m.p, S.map_rad_m2k.p, S.map_rad_k2m.p, S.map_j1, S.map_j5, S.pk_aptr.p, S.pk_afac.p, S.consts.p};
  *Dout = D;
  return 0;
}
#define LAUNCH_TRY(expr)                                                                                 \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e_));          \
  } while (0)

int mistra_chem_drive_dims(int mech, int* j2, int* j6, int* nkc, int* nbgs) {
  if (int rc = check_call(mech, 1)) return rc;
  const MechState& S = g_devs[0].mech[mech];
  if (!S.pack_ready) return fail(std::string("no hand-over table for the ") + kMechName[mech] + " mechanism");
  if (j2) *j2 = S.pack_tab.j2;
  if (j6) *j6 = S.pack_tab.j6;
  if (nkc) *nkc = S.pack_tab.nkc;
  if (nbgs) *nbgs = kBudSlots;
  return 0;
}

int mistra_chem_set_species_maps(int mech, int j1, const int32_t* gas_m2k, const int32_t* gas_k2m, int j5, const int32_t* rad_m2k,
                                 const int32_t* rad_k2m) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, 1)) return rc;
  if (j1 < 0 || j5 < 0 || (j1 > 0 && (!gas_m2k || !gas_k2m)) || (j5 > 0 && (!rad_m2k || !rad_k2m))) return fail("bad species maps");
  const int nvar = kDims[mech][0];
  std::lock_guard<std::mutex> lock(g_mu);
  // what the kernels rely on: every map entry names a VARIABLE species, and no species is written twice by the pack (neither by two
  // map entries nor by a map entry and one of the driver's explicit liquid-phase assignments): the reference assigns in sequence
  // and the later one wins, the kernel assigns in parallel
  std::vector<char> seen((size_t)nvar, 0);
  for (int j = 0; j < j1 + j5; j++) {
    const int32_t* m = j < j1 ? gas_m2k + 2 * j : rad_m2k + 2 * (j - j1);
    const int c = m[0], src = m[1], lim = j < j1 ? j1 : j5;
    if (c < 1 || c > nvar || src < 1 || src > lim) return fail("species map entry out of range");
    if (seen[(size_t)c - 1]++) return fail("a species is mapped twice");
    const int back = j < j1 ? gas_k2m[src - 1] : rad_k2m[src - 1];
    if (back != c) return fail("gas_m2k / gas_k2m (rad_m2k / rad_k2m) are not inverse to each other");
  }
  for (auto& D : g_devs) {
    MechState& S = D.mech[mech];
    if (!S.pack_ready) return fail(std::string("no hand-over table for the ") + kMechName[mech] + " mechanism");
    for (int i = 0; i < S.pack_tab.n_pack(); i++) {
      const int c0 = S.pack_tab.pack[(size_t)4 * i];
      if (c0 < nvar && seen[(size_t)c0]) return fail("a species of the gas maps is also packed from sl1 / sion1");
    }
    HIP_TRY(hipSetDevice(D.id));
    HIP_TRY(S.map_gas_m2k.upload(std::vector<int32_t>(gas_m2k, gas_m2k + 2 * (size_t)j1)));
    HIP_TRY(S.map_gas_k2m.upload(std::vector<int32_t>(gas_k2m, gas_k2m + (size_t)j1)));
    HIP_TRY(S.map_rad_m2k.upload(std::vector<int32_t>(rad_m2k, rad_m2k + 2 * (size_t)j5)));
    HIP_TRY(S.map_rad_k2m.upload(std::vector<int32_t>(rad_k2m, rad_k2m + (size_t)j5)));
    S.map_j1 = j1; S.map_j5 = j5;
    S.maps_ready = true;
  }
  (void)hipSetDevice(g_devs[0].id);
  return 0;
}

int mistra_chem_pack_device(int mech, int ncell, const double* d_s1, const double* d_s3, double* d_sl1, double* d_sion1, const double* d_scal,
                            double* d_var, double* d_fix, void* hip_stream) {
  if (ncell == 0) return 0;
  if (!d_s1 || !d_s3 || !d_sl1 || !d_sion1 || !d_scal || !d_var || !d_fix) return fail("null device pointer");
  DeviceState* D; PackDev P;
  if (int rc = pack_dev(mech, d_var, &D, &P)) return rc;
  LAUNCH_TRY(launch_pack(P, ncell, d_s1, d_s3, d_sl1, d_sion1, d_scal, d_var, d_fix, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

int mistra_chem_unpack_device(int mech, int ncell, const double* d_var, double* d_s1, double* d_s3, double* d_sl1, double* d_sion1, void* hip_stream) {
  if (ncell == 0) return 0;
  if (!d_var || !d_s1 || !d_s3 || !d_sl1 || !d_sion1) return fail("null device pointer");
  DeviceState* D; PackDev P;
  if (int rc = pack_dev(mech, d_var, &D, &P)) return rc;
  LAUNCH_TRY(launch_unpack(P, ncell, d_var, d_s1, d_s3, d_sl1, d_sion1, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

int mistra_chem_budgets_device(int mech, int ncell, const double* d_var, const double* d_fix, const double* d_rconst, double dt, double* d_bg,
                               double* d_bgs, void* hip_stream) {
  if (ncell == 0) return 0;
  if (!d_var || !d_fix || !d_rconst) return fail("null device pointer");
  DeviceState* D; PackDev P;
  if (int rc = pack_dev(mech, d_var, &D, &P)) return rc;
  LAUNCH_TRY(launch_budgets(P, ncell, d_var, d_fix, d_rconst, dt, d_bg, d_bgs, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

int mistra_chem_rates_env_from_c_device(int mech, int ncell, const double* d_var, const double* d_fix, double* d_env, void* hip_stream) {
  if (ncell == 0) return 0;
  if (!d_var || !d_fix || !d_env) return fail("null device pointer");
  DeviceState* D; PackDev P;
  if (int rc = pack_dev(mech, d_var, &D, &P)) return rc;
  if (!D->mech[mech].rates_ready) return fail("no device rate table for this mechanism");
  LAUNCH_TRY(launch_env_from_c(P, ncell, D->mech[mech].rates_nenv, d_var, d_fix, d_env, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

int mistra_chem_drive_device(int mech, int ncell, double* d_s1, double* d_s3, double* d_sl1, double* d_sion1, const double* d_scal, double* d_env,
                             double* d_var, double* d_fix, double tin, double dt, int32_t* d_ierr, int32_t* d_stats, double* d_texit_hexit,
                             double* d_bg, double* d_bgs, void* hip_stream) {
  if (ncell == 0) return 0;
  if (!d_s1 || !d_s3 || !d_sl1 || !d_sion1 || !d_scal || !d_env || !d_var || !d_fix || !d_ierr || !d_stats) return fail("null device pointer");
  DeviceState* D; PackDev P;
  if (int rc = pack_dev(mech, d_var, &D, &P)) return rc;
  MechState& S = D->mech[mech];
  if (!S.rates_ready) return fail("no device rate table for this mechanism");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  HIP_TRY(S.d_rct.reserve((size_t)ncell * (size_t)S.tab.nreact));      // (grow-only: the first call of a size allocates, i.e. synchronises)
  LAUNCH_TRY(launch_pack(P, ncell, d_s1, d_s3, d_sl1, d_sion1, d_scal, d_var, d_fix, st));
  LAUNCH_TRY(launch_env_from_c(P, ncell, S.rates_nenv, d_var, d_fix, d_env, st));
  if (int rc = mistra_chem_update_rconst_device(mech, ncell, d_env, S.d_rct.p, hip_stream)) return rc;
  if (int rc = mistra_chem_integrate_device(mech, ncell, d_var, d_fix, S.d_rct.p, tin, tin + dt, d_var, d_ierr, d_stats, d_texit_hexit, hip_stream)) return rc;
  if (d_bg || d_bgs) LAUNCH_TRY(launch_budgets(P, ncell, d_var, d_fix, S.d_rct.p, dt, d_bg, d_bgs, st));
  LAUNCH_TRY(launch_unpack(P, ncell, d_var, d_s1, d_s3, d_sl1, d_sion1, st));
  return 0;
}

// x_drive for a batch of layers from the model's own arrays in HOST memory (include/mistra_chem.h).  The layers' slabs are gathered into one
// pinned block, go up in one copy, the device chain of mistra_chem_drive_device runs on a private stream, everything the model gets back
// comes down in one copy and is scattered into the model arrays.  Primary device only: a column step is 148 layers.
int mistra_chem_drive_begin(int mech, int nlayer, const int32_t* layer, int n, double* s1, double* s3, double* sl1, double* sion1, const double* scal,
                            const double* env, double tin, double dt, int32_t* ierr, int32_t* stats, double* t_h, double* bg, int nrxn,
                            const int32_t* bg_level, double* bgs, double* c_packed) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, nlayer)) return rc;
  if (nlayer == 0) return 0;
  if (!layer || !s1 || !s3 || !sl1 || !sion1 || !scal || !env) return fail("null host pointer");
  if (bg && (!bg_level || nrxn < kDims[mech][2])) return fail("bg needs bg_level and nrxn >= NREACT");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  MechState& S = D.mech[mech];
  if (!S.pack_ready) return fail(std::string("no hand-over table for the ") + kMechName[mech] + " mechanism");
  if (!S.maps_ready) return fail("mistra_chem_set_species_maps has not been called for this mechanism");
  if (!S.rates_ready) return fail("no device rate table for this mechanism");
  if (S.pend.active) return fail("mistra_chem_drive_begin: the mechanism's previous step has not been fetched (mistra_chem_drive_end)");
  for (int i = 0; i < nlayer; i++) {
    if (layer[i] < 1 || layer[i] > n) return fail("layer index out of range");
    if (bg && (bg_level[i] < 0)) return fail("bg_level out of range");
  }
  HIP_TRY(hipSetDevice(D.id));
  const PackTable& T = S.pack_tab;
  const size_t nl = (size_t)nlayer, j1 = (size_t)S.map_j1, j5 = (size_t)S.map_j5, nsl = (size_t)T.j2 * T.nkc, nsi = (size_t)T.j6 * T.nkc;
  const size_t nv = (size_t)kDims[mech][0], nf = (size_t)kDims[mech][1], nr = (size_t)kDims[mech][2], ne = (size_t)S.rates_nenv, nb = 2 * (size_t)kBudSlots;
  // block layout (doubles): in/out part first — s1 | s3 | sl1 | sion1 | bg | bgs — then in-only — scal | env — then out-only — th(2) | hlast | int32 ierr, stats
  size_t off = 0;
  auto take = [&](size_t count) { const size_t at = off; off += (count + 31) & ~(size_t)31; return at; };      // 256-byte aligned sub-blocks
  const size_t o_s1 = take(nl * j1), o_s3 = take(nl * j5), o_sl1 = take(nl * nsl), o_si = take(nl * nsi), o_bg = take(bg ? nl * 2 * nr : 0), o_bgs = take(bgs ? nl * nb : 0);
  const size_t io_end = off;
  const size_t o_scal = take(nl * 6), o_env = take(nl * ne);
  const size_t in_end = off;
  const size_t o_th = take(nl * 2), o_hl = take(nl), o_int = take((nl * 9 + 1) / 2), o_cp = take(c_packed ? nl * (nv + nf) : 0);
  const size_t out_end = off;
  const size_t o_var = take(nl * nv), o_fix = take(nl * nf), o_rct = take(nl * nr);      // device only
  if (off > S.drv_cap) {
    if (S.drv_dev) (void)hipFree(S.drv_dev);
    if (S.drv_host) (void)hipHostFree(S.drv_host);
    S.drv_dev = S.drv_host = nullptr; S.drv_cap = 0;
    const size_t cap = off + off / 2;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&S.drv_dev), cap * sizeof(double)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S.drv_host), cap * sizeof(double), hipHostMallocDefault));
    S.drv_cap = cap;
  }
  if (!S.drv_stream) HIP_TRY(hipStreamCreateWithFlags(&S.drv_stream, hipStreamNonBlocking));
  double* H = S.drv_host;
  double* Dv = S.drv_dev;
  // ---- gather the layers (the model arrays hold layer k at stride j1 / j5 / j2*nkc / j6*nkc; bg at stride 2*nrxn per level; bgs at 2*122 per layer)
  for (size_t i = 0; i < nl; i++) {
    const size_t k = (size_t)layer[i] - 1;
    std::memcpy(H + o_s1 + i * j1, s1 + k * j1, j1 * sizeof(double));
    std::memcpy(H + o_s3 + i * j5, s3 + k * j5, j5 * sizeof(double));
    std::memcpy(H + o_sl1 + i * nsl, sl1 + k * nsl, nsl * sizeof(double));
    std::memcpy(H + o_si + i * nsi, sion1 + k * nsi, nsi * sizeof(double));
    if (bg) {
      if (bg_level[i] > 0) std::memcpy(H + o_bg + i * 2 * nr, bg + ((size_t)bg_level[i] - 1) * 2 * (size_t)nrxn, 2 * nr * sizeof(double));
      else std::memset(H + o_bg + i * 2 * nr, 0, 2 * nr * sizeof(double));
    }
    if (bgs) std::memcpy(H + o_bgs + i * nb, bgs + k * nb, nb * sizeof(double));
  }
  std::memcpy(H + o_scal, scal, nl * 6 * sizeof(double));
  std::memcpy(H + o_env, env, nl * ne * sizeof(double));
  hipStream_t st = S.drv_stream;
  HIP_TRY(hipMemcpyAsync(Dv, H, in_end * sizeof(double), hipMemcpyHostToDevice, st));
  // KPP's dummy product species are not set by the drivers; the reference carries over what the previous LAYER left in COMMON /GDATA_x/
  // (INTEGRATION.md §4): a batch gives every layer zeros
  HIP_TRY(hipMemsetAsync(Dv + o_var, 0, nl * nv * sizeof(double), st));
  int32_t* d_int = reinterpret_cast<int32_t*>(Dv + o_int);      // [nl] ierr, then [nl][8] stats
  PackDev P{S.pk_pack.p, S.pk_fix.p, S.pk_unpack.p, S.pk_slot_id.p, S.pk_slot_first.p, S.pk_terms.p, S.pk_words.p, S.pk_acc.p, S.pk_envc.p,
            T.n_pack(), T.n_fix(), T.n_unpack(), T.n_slots(), (int)T.terms.size() / 3, (int)T.term_words.size(), (int)T.acc.size() / 2, T.n_envc(),
            T.nvar, T.nfix, S.tab.nreact, T.j2, T.j6, T.nkc, T.preclamp,
            S.map_gas_m2k.p, S.map_gas_k2m.p, S.map_rad_m2k.p, S.map_rad_k2m.p, S.map_j1, S.map_j5, S.pk_aptr.p, S.pk_afac.p, S.consts.p};
  LAUNCH_TRY(launch_pack(P, nlayer, Dv + o_s1, Dv + o_s3, Dv + o_sl1, Dv + o_si, Dv + o_scal, Dv + o_var, Dv + o_fix, st));
  if (c_packed) {
    HIP_TRY(hipMemcpy2DAsync(Dv + o_cp, (nv + nf) * sizeof(double), Dv + o_var, nv * sizeof(double), nv * sizeof(double), nl, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(Dv + o_cp + nv, (nv + nf) * sizeof(double), Dv + o_fix, nf * sizeof(double), nf * sizeof(double), nl, hipMemcpyDeviceToDevice, st));
  }
  LAUNCH_TRY(launch_env_from_c(P, nlayer, S.rates_nenv, Dv + o_var, Dv + o_fix, Dv + o_env, st));
  {
    const RatesDev R{S.rates_consts.p, S.rates_offs.p, S.rates_words.p, S.rates_fslot.p, S.tab.nreact, S.rates_nenv};
    LAUNCH_TRY(launch_update_rconst(R, Dv + o_env, Dv + o_rct, nlayer, st));
  }
  {
    HIP_TRY(S.s_sing.reserve(nl * 8));
    KernelArgs a = make_args(S, nlayer, Dv + o_var, Dv + o_fix, Dv + o_rct, tin, tin + dt, Dv + o_var, d_int, d_int + nl, Dv + o_th);
    a.h_last = Dv + o_hl;
    a.sing_rows = S.s_sing.p;
    S.sing_start = 0; S.sing_count = nl; S.sing_one = false;
    if (int rc = launch(D, mech, a, st)) return rc;
  }
  if (bg || bgs) LAUNCH_TRY(launch_budgets(P, nlayer, Dv + o_var, Dv + o_fix, Dv + o_rct, dt, bg ? Dv + o_bg : nullptr, bgs ? Dv + o_bgs : nullptr, st));
  LAUNCH_TRY(launch_unpack(P, nlayer, Dv + o_var, Dv + o_s1, Dv + o_s3, Dv + o_sl1, Dv + o_si, st));
  // what comes back: the in/out part and the out-only part (two copies: the in-only part between them stays up)
  HIP_TRY(hipMemcpyAsync(H, Dv, io_end * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(H + in_end, Dv + in_end, (out_end - in_end) * sizeof(double), hipMemcpyDeviceToHost, st));
  MechState::PendingDrive& Q = S.pend;
  Q.layer.assign(layer, layer + nl);
  if (bg) Q.level.assign(bg_level, bg_level + nl); else Q.level.clear();
  Q.s1 = s1; Q.s3 = s3; Q.sl1 = sl1; Q.sion1 = sion1; Q.bg = bg; Q.bgs = bgs; Q.t_h = t_h; Q.c_packed = c_packed; Q.ierr = ierr; Q.stats = stats; Q.nrxn = nrxn;
  Q.o_s1 = o_s1; Q.o_s3 = o_s3; Q.o_sl1 = o_sl1; Q.o_si = o_si; Q.o_bg = o_bg; Q.o_bgs = o_bgs; Q.o_th = o_th; Q.o_hl = o_hl; Q.o_int = o_int; Q.o_cp = o_cp;
  Q.active = true;
  return 0;
}

int mistra_chem_drive_end(int mech) {
  if (int rc = check_call(mech, 1)) return rc;
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  MechState& S = D.mech[mech];
  MechState::PendingDrive& Q = S.pend;
  if (!Q.active) return fail("mistra_chem_drive_end: nothing issued for this mechanism");
  Q.active = false;
  HIP_TRY(hipSetDevice(D.id));
  HIP_TRY(hipStreamSynchronize(S.drv_stream));
  const PackTable& T = S.pack_tab;
  const size_t nl = Q.layer.size(), j1 = (size_t)S.map_j1, j5 = (size_t)S.map_j5, nsl = (size_t)T.j2 * T.nkc, nsi = (size_t)T.j6 * T.nkc;
  const size_t nv = (size_t)kDims[mech][0], nf = (size_t)kDims[mech][1], nr = (size_t)kDims[mech][2], nb = 2 * (size_t)kBudSlots;
  const double* H = S.drv_host;
  double *s1 = Q.s1, *s3 = Q.s3, *sl1 = Q.sl1, *sion1 = Q.sion1, *bg = Q.bg, *bgs = Q.bgs, *t_h = Q.t_h, *c_packed = Q.c_packed;
  int32_t *ierr = Q.ierr, *stats = Q.stats;
  const int nrxn = Q.nrxn;
  const int32_t *layer = Q.layer.data(), *bg_level = Q.level.data();
  const size_t o_s1 = Q.o_s1, o_s3 = Q.o_s3, o_sl1 = Q.o_sl1, o_si = Q.o_si, o_bg = Q.o_bg, o_bgs = Q.o_bgs, o_th = Q.o_th, o_hl = Q.o_hl, o_int = Q.o_int, o_cp = Q.o_cp;
  // ---- scatter
  const int32_t* h_int = reinterpret_cast<const int32_t*>(H + o_int);
  for (size_t i = 0; i < nl; i++) {
    const size_t k = (size_t)layer[i] - 1;
    std::memcpy(s1 + k * j1, H + o_s1 + i * j1, j1 * sizeof(double));
    std::memcpy(s3 + k * j5, H + o_s3 + i * j5, j5 * sizeof(double));
    std::memcpy(sl1 + k * nsl, H + o_sl1 + i * nsl, nsl * sizeof(double));
    std::memcpy(sion1 + k * nsi, H + o_si + i * nsi, nsi * sizeof(double));
    if (bg && bg_level[i] > 0) std::memcpy(bg + ((size_t)bg_level[i] - 1) * 2 * (size_t)nrxn, H + o_bg + i * 2 * nr, 2 * nr * sizeof(double));
    if (bgs) std::memcpy(bgs + k * nb, H + o_bgs + i * nb, nb * sizeof(double));
    if (ierr) ierr[i] = h_int[i];
    if (stats) std::memcpy(stats + i * 8, h_int + nl + i * 8, 8 * sizeof(int32_t));
    if (t_h) { t_h[3 * i] = H[o_th + 2 * i]; t_h[3 * i + 1] = H[o_th + 2 * i + 1]; t_h[3 * i + 2] = H[o_hl + i]; }
  }
  if (c_packed) std::memcpy(c_packed, H + o_cp, nl * (nv + nf) * sizeof(double));
  return 0;
}

int mistra_chem_drive(int mech, int nlayer, const int32_t* layer, int n, double* s1, double* s3, double* sl1, double* sion1, const double* scal,
                      const double* env, double tin, double dt, int32_t* ierr, int32_t* stats, double* t_h, double* bg, int nrxn,
                      const int32_t* bg_level, double* bgs, double* c_packed) {
  if (int rc = mistra_chem_drive_begin(mech, nlayer, layer, n, s1, s3, sl1, sion1, scal, env, tin, dt, ierr, stats, t_h, bg, nrxn, bg_level, bgs, c_packed)) return rc;
  if (nlayer == 0) return 0;
  return mistra_chem_drive_end(mech);
}

int mistra_chem_fast_k_mt_device(int mech, int nlayer, const double* d_ff, const double* d_rq, const int32_t* kw, int nkw, int ka, int ifeed,
                                 int nkc_l, const double* d_cw, const double* d_cm, const double* d_freep, const double* d_alpha,
                                 const double* d_vmean, double* d_xkmt, const double* d_t, const double* d_p, double* d_vt, void* hip_stream) {
  if (int rc = check_call(mech, 1)) return rc;
  if (nlayer == 0) return 0;
  if (!d_ff || !d_rq || !kw || !d_cw || !d_cm || !d_freep || !d_alpha || !d_vmean || !d_xkmt) return fail("null pointer");
  if (d_vt && (!d_t || !d_p)) return fail("the sedimentation velocity needs the layers' temperature and pressure (d_t, d_p)");
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_xkmt) != hipSuccess) return fail("d_xkmt is not a device pointer");
  DeviceState* D = device_slot(attr.device);
  if (!D) return fail("the buffers live on a device mistra_chem_init(_devices) did not set up");
  MechState& S = D->mech[mech];
  if (!S.kmt_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no mass-transfer routine (fast_k_mt_a: aer, fast_k_mt_t: tot)");
  const KmtTable& T = S.kmt_tab;
  if (nkw != T.nka || T.nka > kKmtMaxNka) return fail("kw does not have nka entries");
  if (ka < 0 || ka > T.nka || nkc_l < 1 || nkc_l > T.nkc) return fail("ka / nkc_l out of range");
  KmtDev K{S.kmt_lex.p, {0}, T.nx, T.nka, T.nkt, T.nkc, S.tab.nvar + S.tab.nfix, ka, ifeed, nkc_l};
  for (int i = 0; i < T.nka; i++) {
    if (kw[i] < 0 || kw[i] > T.nkt) return fail("kw out of range");      // the kernel's loop limits: checked here, on the host
    K.kw[i] = kw[i];
  }
  HIP_TRY(hipSetDevice(D->id));
  LAUNCH_TRY(launch_fast_k_mt(K, nlayer, d_ff, d_rq, d_cw, d_cm, d_freep, d_alpha, d_vmean, d_xkmt, d_t, d_p, d_vt, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

namespace {
// the device slot the buffer lives on and its table of Henry / equilibrium constants
int liq_state(int mech, const void* d_out, DeviceState** D, MechState** S) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_out) != hipSuccess) return fail("the output is not a device pointer");
  *D = device_slot(attr.device);
  if (!*D) return fail("the buffers live on a device mistra_chem_init(_devices) did not set up");
  *S = &(*D)->mech[mech];
  if (!(*S)->liq_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no liquid-phase routines (henry_a / equil_co_a: aer, henry_t / equil_co_t: tot)");
  return 0;
}
LiqDev liq_dev(const MechState& S) {
  const LiqTable& T = S.liq_tab;
  return LiqDev{S.lq_hkind.p, S.lq_eof.p, S.lq_foff.p, S.lq_boff.p, S.lq_fkind.p, S.lq_farg.p, S.lq_ha0.p, S.lq_hb0.p, S.lq_fa.p, S.lq_fb.p,
                T.nspec, T.nkc_eq, T.henry_tref, T.henry_fct, T.equil_tref};
}
}  // namespace

int mistra_chem_henry_device(int mech, int nlayer, const double* d_tt, double* d_henry, void* hip_stream) {
  if (int rc = check_call(mech, 1)) return rc;
  if (nlayer == 0) return 0;
  if (nlayer < 0) return fail("nlayer < 0");
  if (!d_tt || !d_henry) return fail("null pointer");
  DeviceState* D;
  MechState* S;
  if (int rc = liq_state(mech, d_henry, &D, &S)) return rc;
  HIP_TRY(hipSetDevice(D->id));
  LAUNCH_TRY(launch_henry(liq_dev(*S), nlayer, d_tt, d_henry, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

int mistra_chem_st_coeff_device(int mech, int nlayer, int lp_joyce14bc, int lp_buxmann15alph, const double* d_env, double* d_alpha, void* hip_stream) {
  if (int rc = check_call(mech, 1)) return rc;
  if (nlayer == 0) return 0;
  if (nlayer < 0) return fail("nlayer < 0");
  if (!d_env || !d_alpha) return fail("null pointer");
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_alpha) != hipSuccess) return fail("the output is not a device pointer");
  DeviceState* D = device_slot(attr.device);
  if (!D) return fail("the buffers live on a device mistra_chem_init(_devices) did not set up");
  const MechState& S = D->mech[mech];
  if (!S.stc_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no st_coeff routine (st_coeff_a: aer, st_coeff_t: tot)");
  HIP_TRY(hipSetDevice(D->id));
  const int v = (lp_joyce14bc ? 1 : 0) + (lp_buxmann15alph ? 2 : 0);
  const RatesDev R{S.stc_consts[v].p, S.stc_offs[v].p, S.stc_words[v].p, nullptr, S.stc_tab.v[v].nreact, S.stc_tab.v[v].nenv};
  LAUNCH_TRY(launch_update_rconst(R, d_env, d_alpha, nlayer, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

int mistra_chem_v_mean_device(int mech, int nlayer, const double* d_tt, double* d_vmean, void* hip_stream) {
  if (int rc = check_call(mech, 1)) return rc;
  if (nlayer == 0) return 0;
  if (nlayer < 0) return fail("nlayer < 0");
  if (!d_tt || !d_vmean) return fail("null pointer");
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, d_vmean) != hipSuccess) return fail("the output is not a device pointer");
  DeviceState* D = device_slot(attr.device);
  if (!D) return fail("the buffers live on a device mistra_chem_init(_devices) did not set up");
  const MechState& S = D->mech[mech];
  if (!S.vmean_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no v_mean routine (v_mean_a: aer, v_mean_t: tot)");
  HIP_TRY(hipSetDevice(D->id));
  LAUNCH_TRY(launch_v_mean(S.vm_mass.p, S.vmean_tab.nspec, S.vmean_tab.coef, nlayer, d_tt, d_vmean, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

int mistra_chem_equil_co_device(int mech, int nlayer, int nkc, int j6, const double* d_tt, const double* d_conv2, const double* d_xgamma,
                                double* d_xkef, double* d_xkeb, void* hip_stream) {
  if (int rc = check_call(mech, 1)) return rc;
  if (nlayer == 0) return 0;
  if (nlayer < 0) return fail("nlayer < 0");
  if (!d_tt || !d_conv2 || !d_xgamma || !d_xkef || !d_xkeb) return fail("null pointer");
  DeviceState* D;
  MechState* S;
  if (int rc = liq_state(mech, d_xkef, &D, &S)) return rc;
  const LiqTable& T = S->liq_tab;
  if (nkc < T.nkc_eq) return fail("nkc is smaller than the number of bins the routine sets");
  for (size_t i = 0; i < T.fkind.size(); i++)
    if (T.fkind[i] == 3 && T.farg[i] > j6) return fail("j6 is smaller than an activity-coefficient index the routine reads");
  HIP_TRY(hipSetDevice(D->id));
  LAUNCH_TRY(launch_equil_co(liq_dev(*S), nlayer, nkc, j6, d_tt, d_conv2, d_xgamma, d_xkef, d_xkeb, static_cast<hipStream_t>(hip_stream)));
  return 0;
}

// ---- host-buffer forms of the liq_parm kernels: what a Fortran caller reaches (shim/mistra_kpp_liq.f90).  One grow-only device arena and a pinned host
//      arena of the same layout on the primary device, a private stream: a call gathers its inputs into the pinned arena (or, for caller memory that
//      mistra_chem_pin_host registered, leaves them where they are), sends them up, runs the kernel on the same stream and fetches its outputs; one
//      stream synchronisation per call, no allocation once the arenas have grown to the column's size.
namespace {
struct DevBlock {      // the arrays of one call as 256-byte aligned sub-blocks of the arenas
  std::vector<std::pair<size_t, size_t>> parts;      // (offset, bytes)
  std::vector<const void*> src;                      // per part: where its input comes from (nullptr: not an input)
  std::vector<void*> dst;                            // per part: where its output goes (nullptr: not an output)
  size_t cap = 0;
  size_t add(size_t bytes) {
    const size_t at = cap;
    parts.emplace_back(at, bytes); src.push_back(nullptr); dst.push_back(nullptr);
    cap += (bytes + 255) & ~(size_t)255;
    return parts.size() - 1;
  }
  hipError_t alloc() { return g_liq.ensure(cap ? cap : 256); }
  double* dptr(size_t i) const { return reinterpret_cast<double*>(g_liq.dev + parts[i].first); }
  hipError_t up(size_t i, const void* from) { src[i] = from; return hipSuccess; }
  // every transfer costs ~15 us of latency in the stream whatever its size: neighbouring staged parts travel as ONE copy (the padding between them with
  // them), parts inside a registered caller range go straight from / to the caller's memory
  hipError_t transfer(bool upward) {
    size_t i = 0;
    while (i < parts.size()) {
      const void* p = upward ? src[i] : dst[i];
      if (!p) { i++; continue; }
      if (g_liq.is_pinned(p, parts[i].second)) {
        hipError_t e = upward ? hipMemcpyAsync(g_liq.dev + parts[i].first, p, parts[i].second, hipMemcpyHostToDevice, g_liq.st)
                              : hipMemcpyAsync(const_cast<void*>(p), g_liq.dev + parts[i].first, parts[i].second, hipMemcpyDeviceToHost, g_liq.st);
        if (e != hipSuccess) return e;
        i++;
        continue;
      }
      size_t j = i;      // the run of staged parts i..j-1
      while (j < parts.size() && (upward ? src[j] : dst[j]) && !g_liq.is_pinned(upward ? src[j] : dst[j], parts[j].second)) {
        if (upward) std::memcpy(g_liq.host + parts[j].first, src[j], parts[j].second);
        j++;
      }
      const size_t lo = parts[i].first, hi = parts[j - 1].first + parts[j - 1].second;
      hipError_t e = upward ? hipMemcpyAsync(g_liq.dev + lo, g_liq.host + lo, hi - lo, hipMemcpyHostToDevice, g_liq.st)
                            : hipMemcpyAsync(g_liq.host + lo, g_liq.dev + lo, hi - lo, hipMemcpyDeviceToHost, g_liq.st);
      if (e != hipSuccess) return e;
      i = j;
    }
    return hipSuccess;
  }
  hipError_t send() { return transfer(true); }      // after the last up(), before the kernel
  hipStream_t stream() const { return g_liq.st; }
  hipError_t down(size_t i, void* to) { dst[i] = to; return hipSuccess; }
  hipError_t finish() {       // after the last down(): fetch, wait, hand the staged outputs to the caller
    if (hipError_t e = transfer(false)) return e;
    if (hipError_t e = hipStreamSynchronize(g_liq.st)) return e;
    for (size_t i = 0; i < parts.size(); i++)
      if (dst[i] && !g_liq.is_pinned(dst[i], parts[i].second)) std::memcpy(dst[i], g_liq.host + parts[i].first, parts[i].second);
    return hipSuccess;
  }
};
}  // namespace

int mistra_chem_pin_host(void* p, size_t bytes) {
  if (int rc = lazy_init()) return rc;
  if (!p || !bytes) return fail("mistra_chem_pin_host: null range");
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_liq.is_pinned(p, bytes)) return 0;
  const char* c = static_cast<const char*>(p);
  for (const auto& r : g_liq.pinned)
    if (c < r.first + r.second && r.first < c + bytes) return fail("mistra_chem_pin_host: the range overlaps one registered before");
  HIP_TRY(hipSetDevice(g_devs[0].id));
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
  g_liq.pinned.emplace_back(c, bytes);
  return 0;
}

int mistra_chem_unpin_host(void* p) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (size_t i = 0; i < g_liq.pinned.size(); i++)
    if (g_liq.pinned[i].first == static_cast<const char*>(p)) {
      HIP_TRY(hipHostUnregister(p));
      g_liq.pinned.erase(g_liq.pinned.begin() + (long)i);
      return 0;
    }
  return fail("mistra_chem_unpin_host: not a registered range");
}

int mistra_chem_fast_k_mt(int mech, int nlayer, const double* ff, const double* rq, const int32_t* kw, int nkw, int ka, int ifeed, int nkc_l,
                          const double* cw, const double* cm, const double* freep, const double* alpha, const double* vmean, double* xkmt,
                          const double* t, const double* p, double* vt) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, nlayer)) return rc;
  if (nlayer == 0) return 0;
  if (!ff || !rq || !kw || !cw || !cm || !freep || !alpha || !vmean || !xkmt) return fail("null pointer");
  if (vt && (!t || !p)) return fail("the sedimentation velocity needs the layers' temperature and pressure (t, p)");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  const MechState& S = D.mech[mech];
  if (!S.kmt_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no mass-transfer routine (fast_k_mt_a: aer, fast_k_mt_t: tot)");
  const KmtTable& T = S.kmt_tab;
  HIP_TRY(hipSetDevice(D.id));
  const size_t nl = (size_t)nlayer, grid = (size_t)T.nka * T.nkt, nspec = (size_t)(S.tab.nvar + S.tab.nfix), nkc = (size_t)T.nkc, d8 = sizeof(double);
  DevBlock B;
  const size_t i_ff = B.add(nl * grid * d8), i_rq = B.add(grid * d8), i_cw = B.add(nl * nkc * d8), i_cm = B.add(nl * nkc * d8), i_fp = B.add(nl * d8),
               i_al = B.add(nl * nspec * d8), i_vm = B.add(nl * nspec * d8), i_xk = B.add(nl * nkc * nspec * d8), i_t = B.add(nl * d8), i_p = B.add(nl * d8),
               i_vt = B.add(nl * nkc * d8);
  HIP_TRY(B.alloc());
  HIP_TRY(B.up(i_ff, ff)); HIP_TRY(B.up(i_rq, rq)); HIP_TRY(B.up(i_cw, cw)); HIP_TRY(B.up(i_cm, cm)); HIP_TRY(B.up(i_fp, freep)); HIP_TRY(B.up(i_al, alpha));
  HIP_TRY(B.up(i_vm, vmean)); HIP_TRY(B.up(i_xk, xkmt));
  if (vt) { HIP_TRY(B.up(i_t, t)); HIP_TRY(B.up(i_p, p)); HIP_TRY(B.up(i_vt, vt)); }
  HIP_TRY(B.send());
  if (int rc = mistra_chem_fast_k_mt_device(mech, nlayer, B.dptr(i_ff), B.dptr(i_rq), kw, nkw, ka, ifeed, nkc_l, B.dptr(i_cw), B.dptr(i_cm),
                                            B.dptr(i_fp), B.dptr(i_al), B.dptr(i_vm), B.dptr(i_xk), vt ? B.dptr(i_t) : nullptr,
                                            vt ? B.dptr(i_p) : nullptr, vt ? B.dptr(i_vt) : nullptr, B.stream()))
    return rc;
  HIP_TRY(B.down(i_xk, xkmt));
  if (vt) HIP_TRY(B.down(i_vt, vt));
  HIP_TRY(B.finish());
  return 0;
}

int mistra_chem_henry(int mech, int nlayer, const double* tt, double* henry) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, nlayer)) return rc;
  if (nlayer == 0) return 0;
  if (!tt || !henry) return fail("null pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  const MechState& S = D.mech[mech];
  if (!S.liq_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no liquid-phase routines (henry_a: aer, henry_t: tot)");
  HIP_TRY(hipSetDevice(D.id));
  const size_t nl = (size_t)nlayer, nspec = (size_t)S.liq_tab.nspec;
  DevBlock B;
  const size_t i_t = B.add(nl * sizeof(double)), i_h = B.add(nl * nspec * sizeof(double));
  HIP_TRY(B.alloc());
  HIP_TRY(B.up(i_t, tt));
  HIP_TRY(B.send());
  if (int rc = mistra_chem_henry_device(mech, nlayer, B.dptr(i_t), B.dptr(i_h), B.stream())) return rc;
  HIP_TRY(B.down(i_h, henry));
  HIP_TRY(B.finish());
  return 0;
}

int mistra_chem_dry_rates(int gas, int nlayer, const double* tt, const double* freep, const double* rcd, const double* vmean4, double* xkmtd, double* xeq,
                          double* henry4) {
  if (int rc0 = lazy_init()) return rc0;
  if (nlayer == 0) return 0;
  if (nlayer < 0) return fail("nlayer < 0");
  if (!tt || !freep || !rcd || !xkmtd || !xeq || (gas ? !henry4 : !vmean4)) return fail("null pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  HIP_TRY(hipSetDevice(D.id));
  const size_t nl = (size_t)nlayer, d8 = sizeof(double);
  DevBlock B;
  const size_t i_t = B.add(nl * d8), i_f = B.add(nl * d8), i_r = B.add(nl * 2 * d8), i_v = B.add(nl * 4 * d8), i_x = B.add(nl * 8 * d8), i_q = B.add(nl * d8),
               i_h = B.add(nl * 4 * d8);
  HIP_TRY(B.alloc());
  HIP_TRY(B.up(i_t, tt)); HIP_TRY(B.up(i_f, freep)); HIP_TRY(B.up(i_r, rcd));
  if (gas) HIP_TRY(B.up(i_h, henry4)); else HIP_TRY(B.up(i_v, vmean4));
  const DryRatesArgs A{nlayer, gas ? 1 : 0, B.dptr(i_t), B.dptr(i_f), B.dptr(i_r), B.dptr(i_v), B.dptr(i_x), B.dptr(i_q), B.dptr(i_h)};
  HIP_TRY(B.send());
  LAUNCH_TRY(launch_dry_rates(A, B.stream()));
  HIP_TRY(B.down(i_x, xkmtd)); HIP_TRY(B.down(i_q, xeq));
  if (gas) HIP_TRY(B.down(i_h, henry4));
  HIP_TRY(B.finish());
  return 0;
}

int mistra_chem_cw_rc(int nlayer, int nkt, int nka, int dry, const double* ff, const double* rq, const double* e, const int32_t* kw, int ka, int ifeed,
                      const double* feu, const int32_t* cloud, const double* crys4, double* rc, double* cw, double* cm, double* conv2, int32_t* below) {
  if (int rc0 = lazy_init()) return rc0;
  if (nlayer == 0) return 0;
  if (nlayer < 0 || nkt < 1 || nka < 1 || nkt > 2048 || nka > 4096 || ka < 0 || ka > nka) return fail("cw_rc: bad dimensions (nkt <= 2048, nka <= 4096)");
  if (!ff || !rq || !kw || !rc || !cw || (!dry && (!e || !feu || !cloud || !crys4 || !cm || !conv2))) return fail("null pointer");
  for (int i = 0; i < nka; i++)
    if (kw[i] < 0 || kw[i] > nkt) return fail("kw out of range");      // the kernel's loop limits: checked here, on the host
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  HIP_TRY(hipSetDevice(D.id));
  const size_t nl = (size_t)nlayer, grid = (size_t)nka * nkt, nb = dry ? 2 : 4, d8 = sizeof(double);
  DevBlock B;
  const size_t i_ff = B.add(nl * grid * d8), i_rq = B.add(grid * d8), i_e = B.add((size_t)nkt * d8), i_kw = B.add((size_t)nka * 4), i_feu = B.add(nl * d8),
               i_cl = B.add(nl * 4 * 4), i_rc = B.add(nl * nb * d8), i_cw = B.add(nl * nb * d8), i_cm = B.add(nl * nb * d8), i_cv = B.add(nl * nb * d8),
               i_bl = B.add(nl * 4);
  HIP_TRY(B.alloc());
  HIP_TRY(B.up(i_ff, ff)); HIP_TRY(B.up(i_rq, rq)); HIP_TRY(B.up(i_kw, kw));
  if (!dry) { HIP_TRY(B.up(i_e, e)); HIP_TRY(B.up(i_feu, feu)); HIP_TRY(B.up(i_cl, cloud)); }
  CwRcArgs A{};
  A.nlayer = nlayer; A.nkt = nkt; A.nka = nka; A.ka = ka; A.ial = ifeed == 2 ? 2 : 1; A.dry = dry ? 1 : 0;
  if (!dry) { A.xcryssulf = crys4[0]; A.xcrysss = crys4[1]; A.xdelisulf = crys4[2]; A.xdeliss = crys4[3]; }
  A.kw = reinterpret_cast<const int32_t*>(B.dptr(i_kw)); A.ff = B.dptr(i_ff); A.rq = B.dptr(i_rq); A.e = B.dptr(i_e); A.feu = B.dptr(i_feu);
  A.cloud = reinterpret_cast<const int32_t*>(B.dptr(i_cl)); A.rc = B.dptr(i_rc); A.cw = B.dptr(i_cw); A.cm = B.dptr(i_cm); A.conv2 = B.dptr(i_cv);
  A.below = reinterpret_cast<int32_t*>(B.dptr(i_bl));
  HIP_TRY(B.send());
  LAUNCH_TRY(launch_cw_rc(A, B.stream()));
  HIP_TRY(B.down(i_rc, rc)); HIP_TRY(B.down(i_cw, cw));
  if (!dry) { HIP_TRY(B.down(i_cm, cm)); HIP_TRY(B.down(i_cv, conv2)); if (below) HIP_TRY(B.down(i_bl, below)); }
  HIP_TRY(B.finish());
  return 0;
}

int mistra_chem_st_coeff(int mech, int nlayer, int lp_joyce14bc, int lp_buxmann15alph, const double* env, double* alpha) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, nlayer)) return rc;
  if (nlayer == 0) return 0;
  if (!env || !alpha) return fail("null pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  const MechState& S = D.mech[mech];
  if (!S.stc_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no st_coeff routine (st_coeff_a: aer, st_coeff_t: tot)");
  HIP_TRY(hipSetDevice(D.id));
  const size_t nl = (size_t)nlayer, nspec = (size_t)S.stc_tab.v[0].nreact, nenv = (size_t)S.stc_tab.v[0].nenv;
  DevBlock B;
  const size_t i_e = B.add(nl * nenv * sizeof(double)), i_a = B.add(nl * nspec * sizeof(double));
  HIP_TRY(B.alloc());
  HIP_TRY(B.up(i_e, env));
  HIP_TRY(B.send());
  if (int rc = mistra_chem_st_coeff_device(mech, nlayer, lp_joyce14bc, lp_buxmann15alph, B.dptr(i_e), B.dptr(i_a), B.stream())) return rc;
  HIP_TRY(B.down(i_a, alpha));
  HIP_TRY(B.finish());
  return 0;
}

int mistra_chem_v_mean(int mech, int nlayer, const double* tt, double* vmean) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, nlayer)) return rc;
  if (nlayer == 0) return 0;
  if (!tt || !vmean) return fail("null pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  const MechState& S = D.mech[mech];
  if (!S.vmean_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no v_mean routine (v_mean_a: aer, v_mean_t: tot)");
  HIP_TRY(hipSetDevice(D.id));
  const size_t nl = (size_t)nlayer, nspec = (size_t)S.vmean_tab.nspec;
  DevBlock B;
  const size_t i_t = B.add(nl * sizeof(double)), i_v = B.add(nl * nspec * sizeof(double));
  HIP_TRY(B.alloc());
  HIP_TRY(B.up(i_t, tt));
  HIP_TRY(B.send());
  if (int rc = mistra_chem_v_mean_device(mech, nlayer, B.dptr(i_t), B.dptr(i_v), B.stream())) return rc;
  HIP_TRY(B.down(i_v, vmean));
  HIP_TRY(B.finish());
  return 0;
}

int mistra_chem_equil_co(int mech, int nlayer, int nkc, int j6, const double* tt, const double* conv2, const double* xgamma, double* xkef, double* xkeb) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, nlayer)) return rc;
  if (nlayer == 0) return 0;
  if (!tt || !conv2 || !xgamma || !xkef || !xkeb || nkc < 1 || j6 < 1) return fail("null pointer or bad dimensions");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  const MechState& S = D.mech[mech];
  if (!S.liq_ready) return fail(std::string("the ") + kMechName[mech] + " mechanism has no liquid-phase routines (equil_co_a: aer, equil_co_t: tot)");
  HIP_TRY(hipSetDevice(D.id));
  const size_t nl = (size_t)nlayer, nspec = (size_t)S.liq_tab.nspec, d8 = sizeof(double);
  DevBlock B;
  const size_t i_t = B.add(nl * d8), i_c = B.add(nl * nkc * d8), i_g = B.add(nl * nkc * j6 * d8), i_f = B.add(nl * nkc * nspec * d8), i_b = B.add(nl * nkc * nspec * d8);
  HIP_TRY(B.alloc());
  HIP_TRY(B.up(i_t, tt)); HIP_TRY(B.up(i_c, conv2)); HIP_TRY(B.up(i_g, xgamma)); HIP_TRY(B.up(i_f, xkef)); HIP_TRY(B.up(i_b, xkeb));
  HIP_TRY(B.send());
  if (int rc = mistra_chem_equil_co_device(mech, nlayer, nkc, j6, B.dptr(i_t), B.dptr(i_c), B.dptr(i_g), B.dptr(i_f), B.dptr(i_b), B.stream()))
    return rc;
  HIP_TRY(B.down(i_f, xkef)); HIP_TRY(B.down(i_b, xkeb));
  HIP_TRY(B.finish());
  return 0;
}

int mistra_chem_debug_set_max_steps(int max_steps) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_max_steps = max_steps > 0 ? max_steps : 100000;
  return 0;
}

int mistra_chem_debug_first_step(int mech, int ncell, const double* var_in, const double* fix, const double* rconst, double tin,
                                 double tout, double* dump) {
  if (int rc = check_call(mech, ncell)) return rc;
  if (ncell == 0) return 0;
  if (!var_in || !fix || !rconst || !dump) return fail("null host pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  DeviceState& D = g_devs[0];
  HIP_TRY(hipSetDevice(D.id));
  MechState& S = D.mech[mech];
  const size_t nv = (size_t)kDims[mech][0], nf = (size_t)kDims[mech][1], nr = (size_t)kDims[mech][2], nc = (size_t)ncell;
  const size_t per = 5 * nv + 2 * (size_t)kDims[mech][3] + 2;
  DevBuf<double> d_dump;
  HIP_TRY(S.s_var.reserve(nc * nv));
  HIP_TRY(S.s_fix.reserve(nc * nf));
  HIP_TRY(S.s_rct.reserve(nc * nr));
  HIP_TRY(S.s_ierr.reserve(nc));
  HIP_TRY(S.s_stats.reserve(nc * 8));
  HIP_TRY(d_dump.reserve(nc * per));
  HIP_TRY(hipMemset(d_dump.p, 0, nc * per * sizeof(double)));
  HIP_TRY(hipMemcpy(S.s_var.p, var_in, nc * nv * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.s_fix.p, fix, nc * nf * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S.s_rct.p, rconst, nc * nr * sizeof(double), hipMemcpyHostToDevice));
  KernelArgs a = make_args(S, ncell, S.s_var.p, S.s_fix.p, S.s_rct.p, tin, tout, S.s_var.p, S.s_ierr.p, S.s_stats.p, nullptr);
  a.dump = d_dump.p;
  if (int rc = launch(D, mech, a, nullptr)) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(dump, d_dump.p, nc * per * sizeof(double), hipMemcpyDeviceToHost));
  d_dump.release();
  return 0;
}

int mistra_chem_integrate(int mech, int ncell, const double* var_in, const double* fix, const double* rconst, double tin,
                          double tout, double* var_out, int32_t* ierr, int32_t* stats) {
  return mistra_chem_integrate_ex(mech, ncell, var_in, fix, rconst, tin, tout, var_out, ierr, stats, nullptr);
}

// A Fortran caller has no init hook: the first call brings the library up on device MISTRA_CHEM_DEVICE (default 0), or on the
// first MISTRA_CHEM_DEVICES GPUs of the node when that is set.
static int lazy_init() {
  if (g_inited) return 0;
  if (const char* n = std::getenv("MISTRA_CHEM_DEVICES")) return mistra_chem_init_devices(std::atoi(n), nullptr);
  const char* dev = std::getenv("MISTRA_CHEM_DEVICE");
  return mistra_chem_init(dev ? std::atoi(dev) : 0);
}

static int integrate_host(int mech, int ncell, const double* var_in, const double* fix, const double* rconst, const double* env, double tin,
                          double tout, double* var_out, int32_t* ierr, int32_t* stats, double* t_h);

int mistra_chem_integrate_ex(int mech, int ncell, const double* var_in, const double* fix, const double* rconst, double tin,
                             double tout, double* var_out, int32_t* ierr, int32_t* stats, double* t_h) {
  return integrate_host(mech, ncell, var_in, fix, rconst, nullptr, tin, tout, var_out, ierr, stats, t_h);
}

int mistra_chem_integrate_env_ex(int mech, int ncell, const double* var_in, const double* fix, const double* env, double tin, double tout,
                                 double* var_out, int32_t* ierr, int32_t* stats, double* t_h) {
  return integrate_host(mech, ncell, var_in, fix, nullptr, env, tin, tout, var_out, ierr, stats, t_h);
}

static int integrate_host(int mech, int ncell, const double* var_in, const double* fix, const double* rconst, const double* env, double tin,
                          double tout, double* var_out, int32_t* ierr, int32_t* stats, double* t_h) {
  if (int rc = lazy_init()) return rc;
  if (int rc = check_call(mech, ncell)) return rc;
  if (ncell == 0) return 0;
  if (!var_in || !fix || !var_out || (!rconst && !env)) return fail("null host pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  const size_t nv = (size_t)kDims[mech][0], nf = (size_t)kDims[mech][1], nr = (size_t)kDims[mech][2];
  const size_t ne = env ? (size_t)g_devs[0].mech[mech].rates_nenv : 0;
  const int ndev = (int)g_devs.size();
  for (auto& d : g_devs) d.mech[mech].sing_count = 0;
  if (ndev == 1 || ncell < 2 * ndev) {
    int rc = integrate_host_on(g_devs[0], mech, ncell, var_in, fix, rconst, tin, tout, var_out, ierr, stats, t_h, 0, env);
    (void)hipSetDevice(g_devs[0].id);
    return rc;
  }
  // several devices: contiguous blocks of cells, one host thread per device (cells are independent, kpp.f90:4310-4470:
  // nothing is exchanged, every device uploads, integrates and downloads its own block)
  std::vector<int> rcs((size_t)ndev, 0);
  std::vector<std::string> errs((size_t)ndev);
  std::vector<std::thread> workers;
  for (int d = 0; d < ndev; d++) {
    const int per = ncell / ndev, rem = ncell % ndev;
    const size_t start = (size_t)d * per + (size_t)std::min(d, rem);
    const int count = per + (d < rem ? 1 : 0);
    workers.emplace_back([=, &rcs, &errs]() {
      rcs[(size_t)d] = integrate_host_on(g_devs[(size_t)d], mech, count, var_in + start * nv, fix + start * nf, rconst ? rconst + start * nr : nullptr, tin, tout,
                                         var_out + start * nv, ierr ? ierr + start : nullptr, stats ? stats + start * 8 : nullptr, t_h ? t_h + start * 3 : nullptr, start,
                                         env ? env + start * ne : nullptr);
      if (rcs[(size_t)d]) errs[(size_t)d] = g_err;      // g_err is thread-local: carry the text to the caller's thread
    });
  }
  for (auto& w : workers) w.join();
  (void)hipSetDevice(g_devs[0].id);
  for (int d = 0; d < ndev; d++)
    if (rcs[(size_t)d]) return fail("device " + std::to_string(g_devs[(size_t)d].id) + ": " + errs[(size_t)d]);
  return 0;
}

// text of ros_ErrorMsg_x (gas.f:1474-1509) for an error code
static const char* ros_error_text(int code) {
  switch (code) {
    case -1: return "--> Improper value for maximal no of steps";
    case -2: return "--> Selected Rosenbrock method not implemented";
    case -3: return "--> Hmin/Hmax/Hstart must be positive";
    case -4: return "--> FacMin/FacMax/FacRej must be positive";
    case -5: return "--> Improper tolerance values";
    case -6: return "--> No of steps exceeds maximum bound";
    case -7: return "--> Step size too small: T + 10*H = T or H < Roundoff";
    case -8: return "--> Matrix is repeatedly singular";
  }
  return nullptr;
}

int mistra_chem_integrate_common_status(int mech, void* gdata, double* tin, double* tout, int32_t* ierr_out, double* t_err,
                                        double* h_err, int32_t* nsng) {
  if (int rc = lazy_init()) return rc;
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
  DeviceState& D = g_devs[0];
  MechState& S = D.mech[mech];
  {
    // One call = one cell: what costs here is synchronisation, not bytes.  /GDATA_x/ holds C and RCONST back to back, so
    // the inputs go up in ONE copy from a pinned mirror and everything the kernel writes comes back in ONE, on a private
    // stream with a single wait (seven blocking calls before: 340 us per gas call, of which the kernel is a fraction).
    std::lock_guard<std::mutex> lock(g_mu);
    HIP_TRY(hipSetDevice(D.id));
    const size_t n_in = (size_t)(nv + nf + nr), n_out = (size_t)nv + 2 + 1 + 5 + 4;   // VAR | Texit Hexit | H at exit | 9 int32 in 5 doubles | 8 zero-pivot rows in 4
    if (!S.one_dev) {
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&S.one_dev), (n_in + n_out) * sizeof(double)));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S.one_host), (n_in + n_out) * sizeof(double), hipHostMallocDefault));
      HIP_TRY(hipStreamCreateWithFlags(&S.one_stream, hipStreamNonBlocking));
    }
    std::memcpy(S.one_host, c, n_in * sizeof(double));
    HIP_TRY(hipMemcpyAsync(S.one_dev, S.one_host, n_in * sizeof(double), hipMemcpyHostToDevice, S.one_stream));
    double* d_out = S.one_dev + n_in;
    int32_t* d_stats = reinterpret_cast<int32_t*>(d_out + nv + 3);
    KernelArgs a = make_args(S, 1, S.one_dev, S.one_dev + nv, S.one_dev + nv + nf, *tin, *tout, d_out, d_stats + 8, d_stats, d_out + nv);
    a.h_last = d_out + nv + 2;
    a.sing_rows = d_stats + 10;
    if (int rc = launch(D, mech, a, S.one_stream)) return rc;
    double* h_out = S.one_host + n_in;
    HIP_TRY(hipMemcpyAsync(h_out, d_out, n_out * sizeof(double), hipMemcpyDeviceToHost, S.one_stream));
    HIP_TRY(hipStreamSynchronize(S.one_stream));
    std::memcpy(c, h_out, (size_t)nv * sizeof(double));
    const int32_t* st = reinterpret_cast<const int32_t*>(h_out + nv + 3);
    std::memcpy(&ierr, st + 8, sizeof ierr);
    if (ierr_out) *ierr_out = ierr;
    if (nsng) *nsng = st[7];
    for (auto& d : g_devs) d.mech[mech].sing_count = 0;
    std::memcpy(S.one_sing, st + 10, sizeof S.one_sing);
    S.sing_one = true;
    if (t_err) *t_err = h_out[nv];         // T when the integrator returned
    if (h_err) *h_err = h_out[nv + 2];     // H when the integrator returned (what ros_ErrorMsg_x prints)
    *tin = h_out[nv];          // TIN = RPAR(11), exit time
    *stepmin = h_out[nv + 1];  // STEPMIN = RPAR(12), last step
  }
  return 0;
}

int mistra_chem_singular_rows(int mech, int cell, int32_t* rows8) {
  if (int rc = check_call(mech, 1)) return rc;
  if (!rows8 || cell < 0) return fail("bad argument");
  std::lock_guard<std::mutex> lock(g_mu);
  for (auto& D : g_devs) {
    MechState& S = D.mech[mech];
    if (S.sing_one && cell == 0 && &D == &g_devs[0]) {
      std::memcpy(rows8, S.one_sing, sizeof S.one_sing);
      return 0;
    }
    if (!S.sing_one && S.sing_count > 0 && (size_t)cell >= S.sing_start && (size_t)cell < S.sing_start + S.sing_count) {
      HIP_TRY(hipSetDevice(D.id));
      HIP_TRY(hipMemcpy(rows8, S.s_sing.p + ((size_t)cell - S.sing_start) * 8, 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
      (void)hipSetDevice(g_devs[0].id);
      return 0;
    }
  }
  return fail("no host-buffer integration of this mechanism holds that cell");
}

int mistra_chem_integrate_common(int mech, void* gdata, double* tin, double* tout) {
  int32_t ierr = 1, nsng = 0;
  double t_err = 0.0, h_err = 0.0;
  const double tin_in = tin ? *tin : 0.0;
  if (int rc = mistra_chem_integrate_common_status(mech, gdata, tin, tout, &ierr, &t_err, &h_err, &nsng)) return rc;
  if (nsng > 0) {      // ros_PrepareMatrix_x's warning, one per failed decomposition (gas.f:1456)
    int32_t rows[8] = {0};
    (void)mistra_chem_singular_rows(mech, 0, rows);
    for (int i = 0; i < nsng; i++) std::printf(" Warning: LU Decomposition returned ising =  %d\n", rows[i < 8 ? i : 7]);
  }
  if (ierr < 0) {   // the reference prints and continues (ros_ErrorMsg_x gas.f:1474-1509, INTEGRATE_x gas.f:764-767);
                    // a Fortran caller gets the same lines from unit 6 through shim/mistra_kpp_shim.f90
    const char sfx = "gat"[mech];
    std::printf(" Forced exit from Rosenbrock_%c due to the following error:\n", sfx);
    if (const char* txt = ros_error_text(ierr)) std::printf(" %s\n", txt);
    else std::printf("       Unknown Error code: %4d\n", ierr);
    std::printf("        T=%15.7E and H=%15.7E\n", t_err, h_err);
    std::printf(" Rosenbrock: Unsucessful step at T= %g  (IERR= %d )\n", tin_in, ierr);
  }
  return 0;
}

}  // extern "C"
