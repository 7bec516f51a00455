// Hand-over tables of the reference's per-layer drivers (mistra_amd/mech/<mech>.pack, written by tools/extract_pack.py) and the
// device kernels that do what gas_drive / aer_drive / tot_drive do around Update_RCONST_x + INTEGRATE_x (gas.f:60-217 | aer.f:59-246 |
// tot.f:59-982 with aer_mk.dat / aer_km.dat; budgets bud_x.f / bud_s_x.f): SURVEY.md §8 f2.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace mistra {

struct PackTable {
  int nvar = 0, nfix = 0, j2 = 0, j6 = 0, nkc = 0, preclamp = 0;
  std::vector<int32_t> pack, fix, unpack;                 // [n][4], [n][3], [n][4]  (see tools/extract_pack.py: write_binary)
  std::vector<int32_t> slot_id, slot_first, terms, term_words, acc, envc;
  int n_pack() const { return (int)pack.size() / 4; }
  int n_fix() const { return (int)fix.size() / 3; }
  int n_unpack() const { return (int)unpack.size() / 4; }
  int n_slots() const { return (int)slot_id.size(); }
  int n_envc() const { return (int)envc.size() / 2; }
  bool load(const std::string& path, std::string* err);
};

constexpr int kBudSlots = 122;      // common /budgs/ bgs(2,122,n)   (bud_s_g.f:63)

struct PackDev {                    // device copy of one mechanism's tables + the run-time species maps of module gas_common
  const int32_t *pack, *fix, *unpack, *slot_id, *slot_first, *terms, *term_words, *acc, *envc;
  int n_pack, n_fix, n_unpack, n_slots, n_terms, n_words, n_acc, n_envc;
  int nvar, nfix, nreact, j2, j6, nkc, preclamp;
  // gas_m2k_x(1:2, 1:j1) as the Fortran holds it (C index, s1 index; 1-based), gas_k2m_x(1:j1) (C index of s1(j)); likewise rad_* for s3
  const int32_t *gas_m2k, *gas_k2m, *rad_m2k, *rad_k2m;
  int j1, j5;
  const int32_t *a_ptr, *a_fac;     // Fun_x's rate products A(i) = RCT(i) * prod X[a_fac] (mechanism table): bg(1,i,kl) of bud_x
  const double* consts;             // X = [VAR | FIX | consts]
};

hipError_t launch_pack(const PackDev& P, int ncell, const double* s1, const double* s3, double* sl1, double* sion1, const double* scal,
                       double* var, double* fix, hipStream_t stream);
hipError_t launch_unpack(const PackDev& P, int ncell, const double* var, double* s1, double* s3, double* sl1, double* sion1, hipStream_t stream);
hipError_t launch_budgets(const PackDev& P, int ncell, const double* var, const double* fix, const double* rconst, double dt, double* bg,
                          double* bgs, hipStream_t stream);
hipError_t launch_env_from_c(const PackDev& P, int ncell, int nenv, const double* var, const double* fix, double* env, hipStream_t stream);

}  // namespace mistra

// ---- mass-transfer coefficients (kpp.f90: fast_k_mt_a 2683-2947, fast_k_mt_t 2421-2676; SURVEY.md §8 f3, first slice)
namespace mistra {
struct KmtTable {
  int nx = 0, nka = 0, nkt = 0, nkc = 0;
  std::vector<int32_t> lex;        // C index (1-based) of each exchanged species
  bool load(const std::string& path, std::string* err);
};
constexpr int kKmtMaxNka = 96;
struct KmtDev {
  const int32_t* lex;              // [nx], 1-based
  int32_t kw[kKmtMaxNka];          // [nka], as COMMON /blck06/ holds it (1-based jt limits): travels with the launch — calls on different
                                   // streams cannot overwrite each other's limits, and nothing is staged or waited for
  int nx, nka, nkt, nkc, nspec, ka, ifeed, nkc_l;
};
hipError_t launch_fast_k_mt(const KmtDev& K, int nlayer, const double* ff, const double* rq, const double* cw, const double* cm, const double* freep,
                            const double* alpha, const double* vmean, double* xkmt, const double* tt, const double* pp, double* vt, hipStream_t stream);
}  // namespace mistra

// ---- Henry constants and equilibrium rate constants (kpp.f90: henry_a 1914-2145, henry_t 1676-1907, equil_co_a 3162-3363,
//      equil_co_t 2954-3155).  Tables: mistra_amd/mech/<mech>.liq, written by tools/extract_liq.py.
namespace mistra {
struct LiqTable {
  int nspec = 0, nh = 0, ne = 0, nkc_eq = 0, nfac = 0;
  double henry_tref = 0, henry_fct = 0, equil_tref = 0;
  // dense per-species forms the kernels index by species: henry kind (-1 not set, 0 number, 1 temperature law), a0, b0;
  // equil entry of the species (-1: the routine does not set it)
  std::vector<int32_t> h_kind, e_of, foff, boff, fkind, farg;
  std::vector<double> h_a0, h_b0, fa, fb;
  bool load(const std::string& path, std::string* err);
};
struct LiqDev {
  const int32_t *h_kind, *e_of, *foff, *boff, *fkind, *farg;
  const double *h_a0, *h_b0, *fa, *fb;
  int nspec, nkc_eq;
  double henry_tref, henry_fct, equil_tref;
};
hipError_t launch_henry(const LiqDev& L, int nlayer, const double* tt, double* henry, hipStream_t stream);
// ---- liquid water content, mean radius and the chemistry switches of the four particle bins (kpp.f90: cw_rc 2152-2414; dry_cw_rc 4580-4690 for the
//      layers above the chemistry levels): moments of the two-dimensional particle spectrum ff, summed in the reference's own order
struct CwRcArgs {
  int nlayer, nkt, nka, ka, ial;      // ial: first dry-aerosol class that counts (2 where ifeed = 2, else 1)
  int dry;                            // 0: cw_rc (four bins, cm and conv2 with the crystallisation / deliquescence switches); 1: dry_cw_rc (rcd, cwd of bins 1, 2)
  double xcryssulf, xcrysss, xdelisulf, xdeliss;
  const int32_t* kw;                  // [nka]
  const double *ff, *rq, *e, *feu;    // ff [nlayer][nka][nkt], rq [nka][nkt], e [nkt], feu [nlayer]
  const int32_t* cloud;               // [nlayer][nkc]: cloud(kc,k) of /kpp_l1/ as 0 | 1 (bins 1, 2 are read)
  double *rc, *cw, *cm, *conv2;       // [nlayer][nkc] each (dry: rc, cw are [nlayer][2] = rcd, cwd; cm, conv2 unused)
  int32_t* below;                     // [nlayer]: 1 where feu < min(xcryssulf, xcrysss) (the reference prints a line for those below the inversion)
};
hipError_t launch_cw_rc(const CwRcArgs& A, hipStream_t stream);
// ---- uptake on the dry aerosol: dry_rates_g (kpp.f90:4697-4853), dry_rates_a (:4860-5073), dry_rates_t (:5079-5198).  Four species (HNO3, N2O5, NH3,
//      H2SO4, the routines' idr list), two bins; the caller gathers and scatters by species index, so the kernel is the same for the three mechanisms.
struct DryRatesArgs {
  int nlayer, gas;                 // gas != 0: dry_rates_g — the routine's own mean molecular speeds and the Henry constant of HNO3
  const double *tt, *freep, *rcd;  // [nlayer], [nlayer], [nlayer][2]
  const double* vmean4;            // [nlayer][4]: vmean(idr(l),k) (aer, tot); unused for gas
  double *xkmtd, *xeq;             // [nlayer][2][4] = xkmtd(idr(l),kc,k); [nlayer] = xeq(ind_HNO3,k)
  double* henry4;                  // gas: [nlayer][4] = henry(idr(l),k), in/out
};
hipError_t launch_dry_rates(const DryRatesArgs& A, hipStream_t stream);
// ---- mean molecular speeds (kpp.f90: v_mean_a 1472-1670, v_mean_t 1268-1465).  Table: mistra_amd/mech/<mech>.vmean (tools/extract_vmean.py).
struct VmeanTable {
  int nspec = 0;
  double coef = 0;
  std::vector<double> mass;      // [nspec]: molar mass in kg/mol of the species the routine sets, 0 = not set (vmean stays 0)
  bool load(const std::string& path, std::string* err);
};
hipError_t launch_v_mean(const double* mass, int nspec, double coef, int nlayer, const double* tt, double* vmean, hipStream_t stream);
hipError_t launch_equil_co(const LiqDev& L, int nlayer, int nkc, int j6, const double* tt, const double* conv2, const double* xgamma, double* xkef,
                           double* xkeb, hipStream_t stream);
}  // namespace mistra
