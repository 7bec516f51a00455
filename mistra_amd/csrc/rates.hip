// Update_RCONST_x on the device: the rate constants of a batch of cells from each cell's meteorology,
// switches, photolysis rates and the few concentrations the rate laws read, instead of 8*NREACT bytes per cell from the host.
//
// The generated Update_RCONST_x (gas.f:275-666 | aer.f:304-1364 | tot.f:1040-2768) is one assignment per reaction,
//     RCONST(i) = <product of switches, literals, rate-law calls, array elements>
// tools/extract_rates.py turns those into postfix programs (mistra_amd/mech/<mech>.rates); this file holds the evaluator
// and the rate-law functions of kpp.f90:7127-8601 that the gas mechanism calls (aer / tot add 13 more: next step).
// Arithmetic follows the reference statement by statement: Fortran evaluation order, integer arguments converted where
// the reference converts them, DEFAULT-REAL literals as the double nearest their float32 — `300.` is exact,
// `8.314` is 8.31400012969970703, `0.21` is 0.209999993443489075, `10**(-6.16)` is the single-precision power
// 6.91831189669755986e-07 (what flang folds it to; tools/extract_rates.py documents the check).  exp, pow and log10 are the
// device library's: they differ from the host libm's in the last place, which is the whole of the stated tolerance
// (tests/test_gpu_rates.py: 1e-13 relative).  Built with -ffp-contract=off like the rest.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rates.hpp"

namespace mistra {

namespace {

constexpr double kR8314 = 8.31400012969970703;          // 8.314  as a default-real literal
constexpr double k021 = 0.209999993443489075;           // 0.21
constexpr double kTenPowM616 = 6.91831189669755986e-07; // 10**(-6.16): INTEGER ** REAL is evaluated in single precision

struct Cb1 { double aircc, te, h2oppm, pk; };           // COMMON /cb_1/ (kpp.f90:7140)

__device__ __forceinline__ double fmax_fortran(double a, double b) { return (a > b || b != b) ? a : b; }

// kpp.f90:7127  farr=a*exp(b/te), b INTEGER
__device__ double farr(const Cb1& c, double a, double b) { return a * exp(b / c.te); }
// kpp.f90:7149  farr_sp=a*((te/b)**c)*exp(d/te), b and d INTEGER
__device__ double farr_sp(const Cb1& c, double a, double b, double cc, double d) { return (a * pow(c.te / b, cc)) * exp(d / c.te); }
// Troe fall-off shared by ATK_3 (kpp.f90:7171), ATK_3f (:7301), fbck (:7355), fbckJ (:7383), fbck2 (:7411); tref = 300. | 298.
__device__ double troe(const Cb1& c, double a1, double a2, double b1, double b2, double fc, double tref) {
  const double a0 = (a1 * c.aircc) * pow(c.te / tref, a2);
  const double b0 = b1 * pow(c.te / tref, b2);
  const double l = log10(a0 / b0);
  return (a0 / (1.0 + a0 / b0)) * pow(fc, 1.0 / (1.0 + l * l));
}
// kpp.f90:7327  sHNO3: func(a,b) = a*exp(b*tte), tte = 1./te, b INTEGER
__device__ double shno3(const Cb1& c, double a1, double b1, double a2, double b2, double a3, double b3) {
  const double tte = 1.0 / c.te;
  const double f1 = a1 * exp(b1 * tte), f2 = a2 * exp(b2 * tte), f3 = a3 * exp(b3 * tte);
  return f1 + ((f3 * c.aircc) / (1.0 + (f3 * c.aircc) / f2));
}
// kpp.f90:7483  sp_23
__device__ double sp_23(const Cb1& c, double a1, double b1, double a2, double b2, double a3, double b3) {
  const double tte = 1.0 / c.te;
  const double f1 = a1 * exp(b1 * tte), f2 = (a2 * c.aircc) * exp(b2 * tte);
  const double f3 = (((a3 * c.aircc) * c.h2oppm) * 1.0e-6) * exp(b3 * tte);
  return (f1 + f2) * (1.0 + f3);
}
// kpp.f90:7540  fcn: x2=8.314*te; xmg=pk/x2; fcn=10**(-6.16)*exp(-90.7d3/x2)*xmg*x1
__device__ double fcn(const Cb1& c, double x1) {
  const double x2 = kR8314 * c.te, xmg = c.pk / x2;
  return ((kTenPowM616 * exp(-90.7e3 / x2)) * xmg) * x1;
}
// kpp.f90:8351  DMS_add
__device__ double dms_add(const Cb1& c) {
  const double o2 = k021 * c.aircc, tte = 1.0 / c.te;
  return ((9.5e-39 * exp(5270.0 * tte)) * o2) / (1.0 + (7.5e-29 * exp(5610.0 * tte)) * o2);
}
// kpp.f90:8198  fdhetg(na, nb): uptake on dry aerosol; env slots of what it reads: tools/extract_rates.py ENV["gas"]
__device__ double fdhetg(const double* e, int na, int nb) {
  constexpr int YCWD = 9, YXK = 61, YHENRY = 69, YXEQ = 70, C_HNO3 = 71, C_HNO3L = 72;
  const double ycwd = e[YCWD + na - 1];
  if (nb == 1) {
    const double yx = e[YXK + 0 + na - 1];
    const double x1 = yx * ycwd;
    const double caq = ((e[C_HNO3L + na - 1] * 1.5e3) * 1.0e-2) / (e[YXEQ] + 1.0e-2);
    double x2 = 0.0;
    if (e[C_HNO3] != 0.0 && e[YHENRY] != 0.0) x2 = ((-yx) / (e[C_HNO3] * e[YHENRY])) * caq;
    return fmax_fortran(0.0, x1 + x2);
  }
  return e[YXK + 2 * (nb - 1) + na - 1] * ycwd;        // N2O5, NH3, H2SO4
}

__global__ __launch_bounds__(256) void update_rconst_kernel(const RatesDev R, const double* __restrict__ env, double* __restrict__ rconst,
                                                            int ncell) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)ncell * R.nreact) return;
  const int cell = (int)(gid / R.nreact), r = (int)(gid % R.nreact);
  const double* e = env + (size_t)cell * R.nenv;
  const Cb1 cb{e[0], e[1], e[2], e[3]};
  double st[12];
  int sp = 0;
  for (int w = R.offs[r]; w < R.offs[r + 1]; w++) {
    const int word = R.words[w], op = word & 0xFF, arg = word >> 8;
    switch (op) {
      case 0: st[sp++] = R.consts[arg]; break;
      case 1: st[sp++] = e[arg]; break;
      case 2: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
      case 3: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
      case 4: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
      case 5: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
      case 6: st[sp - 1] = -st[sp - 1]; break;
      default: {      // call: arguments are the top of the stack, first argument deepest
        double v = 0.0;
        switch (arg) {
          case 0: sp -= 2; v = farr(cb, st[sp], st[sp + 1]); break;
          case 1: sp -= 4; v = farr_sp(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3]); break;
          case 2: sp -= 5; v = troe(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3], st[sp + 4], 300.0); break;       // ATK_3
          case 3: sp -= 5; v = troe(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3], st[sp + 4], 298.0); break;       // ATK_3f
          case 4: sp -= 6; v = shno3(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3], st[sp + 4], st[sp + 5]); break;
          case 5: sp -= 7; v = troe(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3], st[sp + 4], 300.0) / (st[sp + 5] * exp(st[sp + 6] / cb.te)); break;   // fbck
          case 6: sp -= 6; v = troe(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3], 0.6, 300.0) / (st[sp + 4] * exp(st[sp + 5] / cb.te)); break;        // fbckJ
          case 7: {   // fbck2 (kpp.f90:7411): ak=5.44d-9, bk=14192.d0; 0 where ck = 0
            sp -= 6;
            const double x1 = troe(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3], st[sp + 4], 300.0), ck = st[sp + 5];
            v = ck != 0.0 ? x1 / (((((5.44e-9 * exp(14192.0 / cb.te)) * kR8314) / 101325.0) * cb.te) / ck) : 0.0;
            break;
          }
          case 8: sp -= 2; v = st[sp] * (1.0 + cb.aircc / st[sp + 1]); break;                                         // sp_17
          case 9: sp -= 6; v = sp_23(cb, st[sp], st[sp + 1], st[sp + 2], st[sp + 3], st[sp + 4], st[sp + 5]); break;
          case 10: sp -= 1; v = fcn(cb, st[sp]); break;
          case 11: v = dms_add(cb); break;
          case 12: sp -= 2; v = fdhetg(e, (int)st[sp], (int)st[sp + 1]); break;
        }
        st[sp++] = v;
      }
    }
  }
  rconst[(size_t)cell * R.nreact + r] = st[0];
}

}  // namespace

hipError_t launch_update_rconst(const RatesDev& R, const double* d_env, double* d_rconst, int ncell, hipStream_t stream) {
  if (ncell <= 0) return hipSuccess;
  const long long total = (long long)ncell * R.nreact;
  hipLaunchKernelGGL(update_rconst_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, R, d_env, d_rconst, ncell);
  return hipGetLastError();
}

}  // namespace mistra
