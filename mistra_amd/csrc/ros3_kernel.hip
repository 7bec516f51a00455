// One workgroup integrates one grid cell: the whole Ros3 Rosenbrock step loop of the reference
// (INTEGRATE_x -> Rosenbrock_x -> RosenbrockIntegrator_x, gas.f:710-1337 | aer.f:1408-2035 | tot.f:2812-3439)
// runs inside the kernel with the cell's state on chip:
//
//   LDS        M  = [ Ghimj (LU_NONZERO) | XS (NVAR) ]   matrix being factorised / solve vector   (LDS VM memory)
//              X  = [ V (NVAR) | F (NFIX) | consts ]     extended species vector read by Fun/Jac products
//              AB = A(NREACT) or B(NB)                   rate products, then Jacobian products
//   registers  one species per lane-slot: Y, Ynew, Fcn0, Fcn, K1..K3 (thread t owns species q*NT+t);
//              RCONST of the reactions the thread owns; the structurally non-zero entries of Jac0
//   HBM        read VAR, FIX, RCONST once (coalesced, cell-major), write VAR once; schedule words stream from L2
//
// gfx950 only.  Built with -ffp-contract=off: every multiply and add/subtract rounds once, as in the reference
// built without FMA contraction.  MFMA (v_mfma_f64_16x16x4_f64) is used in ONE place: the last 64x64 block of the tot
// mechanism's LU, which fill-in makes completely dense (dense_lu below); everything else is sparse gather arithmetic.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "kernel_args.hpp"
#include "ros3_kernel.hpp"

namespace mistra {

namespace {

constexpr uint16_t kPosDiag = 0x8000, kPosNone = 0xFFFF;
constexpr int kRingSlots = 8;  // 16-byte table loads in flight per lane (schedule.cpp appends 2x that many rows of slack)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // plain vector types load from any address space
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class T>
using gptr = const T __attribute__((address_space(1)))*;
template <class T>
using gptr_mut = T __attribute__((address_space(1)))*;
template <class T>
__device__ __forceinline__ gptr<T> G_(const T* p) { return (gptr<T>)p; }
template <class T>
__device__ __forceinline__ gptr_mut<T> GM_(T* p) { return (gptr_mut<T>)p; }

// LDS accesses by 32-bit LDS address.  A `double*` that crosses a (non-inlined) function boundary is a generic pointer:
// every access through it pays a generic->LDS conversion with a null check (4 VALU + a scalar load per operand).
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ double lds_ld(uint32_t addr) { return *(const lds_f64*)(uintptr_t)addr; }
__device__ __forceinline__ void lds_st(uint32_t addr, double v) { *(lds_f64*)(uintptr_t)addr = v; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. wait for the
// schedule-table prefetches that are deliberately kept in flight across rounds.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Ros3_x (gas.f:1596-1626)
constexpr double kRosA1 = 1.0;
constexpr double kRosC1 = -0.10156171083877702091975600115545e+01;
constexpr double kRosC2 = 0.40759956452537699824805835358067e+01;
constexpr double kRosC3 = 0.92076794298330791242156818474003e+01;
constexpr double kRosM1 = 0.1e+01;
constexpr double kRosM2 = 0.61697947043828245592553615689730e+01;
constexpr double kRosM3 = -0.42772256543218573326238373806514e+00;
constexpr double kRosE1 = 0.5e+00;
constexpr double kRosE2 = -0.29079558716805469821718236208017e+01;
constexpr double kRosE3 = 0.22354069897811569627360909276199e+00;
constexpr double kRosGamma1 = 0.43586652150845899941601945119356e+00;
[[maybe_unused]] constexpr double kRosGamma2 = 0.24291996454816804366592249683314e+00;
[[maybe_unused]] constexpr double kRosGamma3 = 0.21851380027664058511513169485832e+01;
constexpr double kRosElo = 3.0;

// Err**(1/ros_ELO) of the step-size controller (gas.f:1303).  Not inlined: the library routine's two dozen polynomial
// coefficients were hoisted out of the step loop as live registers, and the 128-register kernel spilled them to scratch and
// read them back every step (most of its HBM-side traffic).  The 256-register kernel keeps the inlined form (0.7 % faster there).
__device__ __attribute__((noinline)) double err_root(double err) { return pow(err, 1.0 / kRosElo); }

// A value every lane of the wave holds identically (time, step size: sums reduced in a fixed order, the same in all lanes),
// moved to scalar registers: the register-starved kernels spilled these to scratch at the head of the step loop.
__device__ __forceinline__ double wave_uniform(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}

// A constant formed in scalar registers where it is used: hoisted out of the step loop, 0.1 became a vector-register pair the
// register-starved kernels kept in scratch and re-read three times per step.
__device__ __forceinline__ double scalar_const(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}

// Sum over the wave, the same value in every lane: six pairing steps in which both partners add the same two numbers (so all lanes
// agree bit for bit).  Inside a lane row the partner's value comes by DPP (quad swaps, half-row mirror, row mirror: ~10 cycles a step),
// across lane rows by v_permlane16/32_swap; __shfl_xor went through ds_bpermute, an LDS round trip per step (six of them: ~800 cycles of
// every step's error norm).
template <int CTRL>
__device__ __forceinline__ double dpp_partner(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_partner<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_partner<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_partner<0x141>(v);      // row_half_mirror: quads 0 <-> 1, 2 <-> 3
  v += dpp_partner<0x140>(v);      // row_mirror: lanes 0-7 <-> 8-15
  {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    // permlane16_swap(a, b) from a = b: a = [r0 r0 r2 r2], b = [r1 r1 r3 r3] (lane rows): their sum pairs rows 0+1 and 2+3 in every lane
    const auto l = __builtin_amdgcn_permlane16_swap((uint32_t)u, (uint32_t)u, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((uint32_t)(u >> 32), (uint32_t)(u >> 32), false, false);
    v = __builtin_bit_cast(double, (uint64_t)l[0] | ((uint64_t)h[0] << 32)) + __builtin_bit_cast(double, (uint64_t)l[1] | ((uint64_t)h[1] << 32));
  }
  {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    // permlane32_swap(a, b) from a = b: a = [lower lower], b = [upper upper]
    const auto l = __builtin_amdgcn_permlane32_swap((uint32_t)u, (uint32_t)u, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((uint32_t)(u >> 32), (uint32_t)(u >> 32), false, false);
    v = __builtin_bit_cast(double, (uint64_t)l[0] | ((uint64_t)h[0] << 32)) + __builtin_bit_cast(double, (uint64_t)l[1] | ((uint64_t)h[1] << 32));
  }
  return v;
}

__device__ __forceinline__ double fmin_f(double a, double b) { return (a < b || b != b) ? a : b; }   // Fortran MIN
__device__ __forceinline__ double fmax_f(double a, double b) { return (a > b || b != b) ? a : b; }   // Fortran MAX

// ---- table look-ahead ring of the tail chain and the gather-sum machine: eight 16-byte loads in flight per lane.
// hipcc's own s_waitcnt placement falls back to vmcnt(0) around branches and barriers, which would serialise every
// table row behind a full memory round trip, so the loads are issued from asm and counted by hand.  A compiler-visible
// VGPR destination would be unsafe (the compiler may copy an asm output before the data lands, cdna_hip_programming.md
// §5.7 item 1).  The ring therefore lives in fixed registers named only by the two statements below; the consume
// statement waits and copies out in ONE asm (§5.7 form i).  Loads are in flight only inside the two non-inlined functions
// that use the ring (each drains it before returning), and those functions' own values sit below the ring (they need
// < 64 registers; mistra_amd/build.py checks the generated ISA before it links), so nothing of the compiler's can be hit by a landing
// load; the callers see the blocks as ordinary call-clobbered registers.  Two placements (LOW):
//   false  v192-199, v208-215, v224-231, v240-247: four caller-saved blocks at the top of the file, nothing to save — for
//          kernels that run two waves per SIMD and have 256 registers (tot: one cell fills a CU's LDS anyway);
//   true   v64-71, v80-87, v96-103, v112-119: four caller-saved blocks again, for the kernels held to 128 registers (FOUR waves
//          per SIMD — aer: two cells per CU instead of one, +39 %; gas: eight cells per CU); the functions' own values stay below v64
//          (the build checks it).  (Until round 2 this ring was v96-127: two of those blocks are callee-saved, every call
//          of gsum_run / tail_solve stored and reloaded 16 registers per lane — most of the aer kernel's HBM-side traffic.)
// Loads return in issue order, hence "at most PENDING outstanding" means the oldest one — the slot about to be
// consumed — has landed.  (An earlier version kept the ring in AGPRs: any AGPR use halves the compiler's VGPR budget
// on gfx950, which cost the kernel ~100 spilled registers.)
template <bool LOW, int K, int BYTE_OFFSET = 0>
__device__ __forceinline__ void vm_ring_load(gptr<u32x4> p) {
#define MISTRA_RING_LOAD(R0, R1, R2, R3)                                                                              \
  asm volatile("global_load_dwordx4 v[" #R0 ":" #R3 "], %0, off offset:%1" : : "v"(p), "n"(BYTE_OFFSET)             \
               : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3)
  if constexpr (LOW) {
    if constexpr (K == 0) MISTRA_RING_LOAD(64, 65, 66, 67);
    else if constexpr (K == 1) MISTRA_RING_LOAD(68, 69, 70, 71);
    else if constexpr (K == 2) MISTRA_RING_LOAD(80, 81, 82, 83);
    else if constexpr (K == 3) MISTRA_RING_LOAD(84, 85, 86, 87);
    else if constexpr (K == 4) MISTRA_RING_LOAD(96, 97, 98, 99);
    else if constexpr (K == 5) MISTRA_RING_LOAD(100, 101, 102, 103);
    else if constexpr (K == 6) MISTRA_RING_LOAD(112, 113, 114, 115);
    else MISTRA_RING_LOAD(116, 117, 118, 119);
  } else {
    if constexpr (K == 0) MISTRA_RING_LOAD(192, 193, 194, 195);
    else if constexpr (K == 1) MISTRA_RING_LOAD(196, 197, 198, 199);
    else if constexpr (K == 2) MISTRA_RING_LOAD(208, 209, 210, 211);
    else if constexpr (K == 3) MISTRA_RING_LOAD(212, 213, 214, 215);
    else if constexpr (K == 4) MISTRA_RING_LOAD(224, 225, 226, 227);
    else if constexpr (K == 5) MISTRA_RING_LOAD(228, 229, 230, 231);
    else if constexpr (K == 6) MISTRA_RING_LOAD(240, 241, 242, 243);
    else MISTRA_RING_LOAD(244, 245, 246, 247);
  }
#undef MISTRA_RING_LOAD
}

// The same load with the table's base in scalar registers and the lane's 32-bit byte offset in ONE vector register: the row a load
// fetches is base + ROW KiB.  Rows 0..3 go into the instruction's immediate offset (13 bits signed on gfx950), rows 4..8 take the
// second base, 4 KiB further on.  (With per-lane 64-bit pointers the compiler kept one 64-bit row offset per load in scalar
// registers — enough of them to reach the callee-saved ones, whose save area cost tail_solve_columns a scratch store and reload
// per call: aer's last per-step HBM traffic.)
struct RingBase { uint64_t b0, b1; };      // b1 = b0 + 4096
__device__ __forceinline__ RingBase ring_base(const void* p) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  uint32_t lo, hi;
  // (a scalar register written by a vector instruction may not feed a memory instruction's address for 5 wait states, and the
  // compiler's hazard pass does not look inside the asm statements that issue the loads: the wait is spelled out here)
  asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 4" : "=s"(lo), "=s"(hi) : "v"((uint32_t)u), "v"((uint32_t)(u >> 32)));
  const uint64_t b = (uint64_t)lo | ((uint64_t)hi << 32);
  return RingBase{b, b + 4096};
}
__device__ __forceinline__ void ring_advance(RingBase& r, uint32_t bytes) { r.b0 += bytes; r.b1 += bytes; }
template <bool LOW, int K, int ROW>
__device__ __forceinline__ void vm_ring_load_s(const RingBase& r, uint32_t voff) {
  static_assert(ROW >= 0 && ROW <= 7, "row within the two 4-KiB windows");
#define MISTRA_RING_LOAD_S(R0, R1, R2, R3)                                                                                    \
  asm volatile("global_load_dwordx4 v[" #R0 ":" #R3 "], %0, %1 offset:%2" : : "v"(voff), "s"(ROW < 4 ? r.b0 : r.b1), "n"((ROW % 4) * 1024) \
               : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3)
  if constexpr (LOW) {
    if constexpr (K == 0) MISTRA_RING_LOAD_S(64, 65, 66, 67);
    else if constexpr (K == 1) MISTRA_RING_LOAD_S(68, 69, 70, 71);
    else if constexpr (K == 2) MISTRA_RING_LOAD_S(80, 81, 82, 83);
    else if constexpr (K == 3) MISTRA_RING_LOAD_S(84, 85, 86, 87);
    else if constexpr (K == 4) MISTRA_RING_LOAD_S(96, 97, 98, 99);
    else if constexpr (K == 5) MISTRA_RING_LOAD_S(100, 101, 102, 103);
    else if constexpr (K == 6) MISTRA_RING_LOAD_S(112, 113, 114, 115);
    else MISTRA_RING_LOAD_S(116, 117, 118, 119);
  } else {
    if constexpr (K == 0) MISTRA_RING_LOAD_S(192, 193, 194, 195);
    else if constexpr (K == 1) MISTRA_RING_LOAD_S(196, 197, 198, 199);
    else if constexpr (K == 2) MISTRA_RING_LOAD_S(208, 209, 210, 211);
    else if constexpr (K == 3) MISTRA_RING_LOAD_S(212, 213, 214, 215);
    else if constexpr (K == 4) MISTRA_RING_LOAD_S(224, 225, 226, 227);
    else if constexpr (K == 5) MISTRA_RING_LOAD_S(228, 229, 230, 231);
    else if constexpr (K == 6) MISTRA_RING_LOAD_S(240, 241, 242, 243);
    else MISTRA_RING_LOAD_S(244, 245, 246, 247);
  }
#undef MISTRA_RING_LOAD_S
}

// ... and with the row's base handed over as it stands (a compile-time offset from a table's start: two scalar additions)
template <bool LOW, int K, int IMM>
__device__ __forceinline__ void vm_ring_load_at(uint64_t sbase, uint32_t voff) {
  static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate offset");
#define MISTRA_RING_LOAD_AT(R0, R1, R2, R3)                                                                                   \
  asm volatile("global_load_dwordx4 v[" #R0 ":" #R3 "], %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "n"(IMM)                \
               : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3)
  if constexpr (LOW) {
    if constexpr (K == 0) MISTRA_RING_LOAD_AT(64, 65, 66, 67);
    else if constexpr (K == 1) MISTRA_RING_LOAD_AT(68, 69, 70, 71);
    else if constexpr (K == 2) MISTRA_RING_LOAD_AT(80, 81, 82, 83);
    else if constexpr (K == 3) MISTRA_RING_LOAD_AT(84, 85, 86, 87);
    else if constexpr (K == 4) MISTRA_RING_LOAD_AT(96, 97, 98, 99);
    else if constexpr (K == 5) MISTRA_RING_LOAD_AT(100, 101, 102, 103);
    else if constexpr (K == 6) MISTRA_RING_LOAD_AT(112, 113, 114, 115);
    else MISTRA_RING_LOAD_AT(116, 117, 118, 119);
  } else {
    if constexpr (K == 0) MISTRA_RING_LOAD_AT(192, 193, 194, 195);
    else if constexpr (K == 1) MISTRA_RING_LOAD_AT(196, 197, 198, 199);
    else if constexpr (K == 2) MISTRA_RING_LOAD_AT(208, 209, 210, 211);
    else if constexpr (K == 3) MISTRA_RING_LOAD_AT(212, 213, 214, 215);
    else if constexpr (K == 4) MISTRA_RING_LOAD_AT(224, 225, 226, 227);
    else if constexpr (K == 5) MISTRA_RING_LOAD_AT(228, 229, 230, 231);
    else if constexpr (K == 6) MISTRA_RING_LOAD_AT(240, 241, 242, 243);
    else MISTRA_RING_LOAD_AT(244, 245, 246, 247);
  }
#undef MISTRA_RING_LOAD_AT
}

template <bool LOW, int K, int PENDING = 7>
__device__ __forceinline__ u32x4 vm_ring_take() {
  uint32_t x, y, z, w;
#define MISTRA_RING_TAKE(R0, R1, R2, R3)                                                                              \
  asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1                                   \
               "\n\tv_mov_b32 %2, v" #R2 "\n\tv_mov_b32 %3, v" #R3                                                    \
               : "=v"(x), "=v"(y), "=v"(z), "=v"(w) : "n"(PENDING) : "memory")
  if constexpr (LOW) {
    if constexpr (K == 0) MISTRA_RING_TAKE(64, 65, 66, 67);
    else if constexpr (K == 1) MISTRA_RING_TAKE(68, 69, 70, 71);
    else if constexpr (K == 2) MISTRA_RING_TAKE(80, 81, 82, 83);
    else if constexpr (K == 3) MISTRA_RING_TAKE(84, 85, 86, 87);
    else if constexpr (K == 4) MISTRA_RING_TAKE(96, 97, 98, 99);
    else if constexpr (K == 5) MISTRA_RING_TAKE(100, 101, 102, 103);
    else if constexpr (K == 6) MISTRA_RING_TAKE(112, 113, 114, 115);
    else MISTRA_RING_TAKE(116, 117, 118, 119);
  } else {
    if constexpr (K == 0) MISTRA_RING_TAKE(192, 193, 194, 195);
    else if constexpr (K == 1) MISTRA_RING_TAKE(196, 197, 198, 199);
    else if constexpr (K == 2) MISTRA_RING_TAKE(208, 209, 210, 211);
    else if constexpr (K == 3) MISTRA_RING_TAKE(212, 213, 214, 215);
    else if constexpr (K == 4) MISTRA_RING_TAKE(224, 225, 226, 227);
    else if constexpr (K == 5) MISTRA_RING_TAKE(228, 229, 230, 231);
    else if constexpr (K == 6) MISTRA_RING_TAKE(240, 241, 242, 243);
    else MISTRA_RING_TAKE(244, 245, 246, 247);
  }
#undef MISTRA_RING_TAKE
  return u32x4{x, y, z, w};
}
// The table rows of the tail chain hold two 16-bit Ghimj cell numbers per word (low half: the rows of register 0, high half: of
// register 1).  This takes slot K out of the ring and turns the halves that are wanted straight into LDS byte addresses, one
// v_lshlrev_b32_sdwa per cell reading the ring register itself: copied out first and decoded by the compiler it was 4 moves, 4 ands and
// 8 shifts per slot, a fifth of the instructions of a chain that is bound by instruction issue (a lone wave: one instruction per
// 4-7 cycles whatever it is).  three: a register holding 3 (the shift count; SDWA takes no literal).
template <bool LOW, int K, int PENDING, bool LO, bool HI>
__device__ __forceinline__ void vm_ring_take_cells(uint32_t* lo, uint32_t* hi, uint32_t three) {
  static_assert(LO || HI, "nothing to take");
#define MISTRA_SDWA(D, R, W) "\n\tv_lshlrev_b32_sdwa %" #D ", %[three], v" #R " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_" #W
#define MISTRA_TAKE_CELLS(R0, R1, R2, R3)                                                                                              \
  if constexpr (LO && HI)                                                                                                              \
    asm volatile("s_waitcnt vmcnt(%[pend])" MISTRA_SDWA(0, R0, 0) MISTRA_SDWA(1, R1, 0) MISTRA_SDWA(2, R2, 0) MISTRA_SDWA(3, R3, 0)    \
                 MISTRA_SDWA(4, R0, 1) MISTRA_SDWA(5, R1, 1) MISTRA_SDWA(6, R2, 1) MISTRA_SDWA(7, R3, 1)                              \
                 : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3])      \
                 : [three] "v"(three), [pend] "n"(PENDING) : "memory");                                                              \
  else if constexpr (LO)                                                                                                               \
    asm volatile("s_waitcnt vmcnt(%[pend])" MISTRA_SDWA(0, R0, 0) MISTRA_SDWA(1, R1, 0) MISTRA_SDWA(2, R2, 0) MISTRA_SDWA(3, R3, 0)    \
                 : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]) : [three] "v"(three), [pend] "n"(PENDING) : "memory");      \
  else                                                                                                                                 \
    asm volatile("s_waitcnt vmcnt(%[pend])" MISTRA_SDWA(0, R0, 1) MISTRA_SDWA(1, R1, 1) MISTRA_SDWA(2, R2, 1) MISTRA_SDWA(3, R3, 1)    \
                 : "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]) : [three] "v"(three), [pend] "n"(PENDING) : "memory");
  if constexpr (LOW) {
    if constexpr (K == 0) { MISTRA_TAKE_CELLS(64, 65, 66, 67) }
    else if constexpr (K == 1) { MISTRA_TAKE_CELLS(68, 69, 70, 71) }
    else if constexpr (K == 2) { MISTRA_TAKE_CELLS(80, 81, 82, 83) }
    else if constexpr (K == 3) { MISTRA_TAKE_CELLS(84, 85, 86, 87) }
    else if constexpr (K == 4) { MISTRA_TAKE_CELLS(96, 97, 98, 99) }
    else if constexpr (K == 5) { MISTRA_TAKE_CELLS(100, 101, 102, 103) }
    else if constexpr (K == 6) { MISTRA_TAKE_CELLS(112, 113, 114, 115) }
    else { MISTRA_TAKE_CELLS(116, 117, 118, 119) }
  } else {
    if constexpr (K == 0) { MISTRA_TAKE_CELLS(192, 193, 194, 195) }
    else if constexpr (K == 1) { MISTRA_TAKE_CELLS(196, 197, 198, 199) }
    else if constexpr (K == 2) { MISTRA_TAKE_CELLS(208, 209, 210, 211) }
    else if constexpr (K == 3) { MISTRA_TAKE_CELLS(212, 213, 214, 215) }
    else if constexpr (K == 4) { MISTRA_TAKE_CELLS(224, 225, 226, 227) }
    else if constexpr (K == 5) { MISTRA_TAKE_CELLS(228, 229, 230, 231) }
    else if constexpr (K == 6) { MISTRA_TAKE_CELLS(240, 241, 242, 243) }
    else { MISTRA_TAKE_CELLS(244, 245, 246, 247) }
  }
#undef MISTRA_TAKE_CELLS
#undef MISTRA_SDWA
}
static_assert(kRingSlots == 8, "the ring helpers above are written for 8 slots");

// ---- the LDS VM executor (schedule.hpp), hand-scheduled: one asm statement holds the whole program loop; its instruction
// stream is generated (tools/gen_vm_asm.py -> vm_exec_asm.inc, where the pipeline and the wait counts are explained).
//   * records land straight in VGPRs: a ring of N 8-register slots in caller-saved blocks (v48-55, v64-71, ...: N = MT::VM_SLOTS), named only inside this statement, so no compiler copy can get between a load
//     and its counted wait (cdna_hip_programming.md §5.7); a slot is refilled (row + N) behind its record's store;
//   * d0, d2..d7 of a record are LDS byte addresses as they stand (M starts at LDS address 0, checked at kernel entry);
//     every mark sits on d1: one v_readfirstlane per record, one scalar test for "any mark" on the main line;
//   * round 3: the six operand gathers of record S+1 are issued at the head of record S (two operand register sets), so a
//     lone wave no longer waits out an LDS round trip per record: measured ~250 cycles per record row before, the record's
//     own instruction issue now.  The arithmetic and its order are unchanged (bit-identical results);
//   * marked rows (aux operand: scale by a pivot reciprocal, or publish one; end of round; null rows) run out of line.
// A continuation record reloads its target: its lane's previous record stored it, and LDS is in-order within a wave.
// The IEEE reciprocal is the sequence hipcc emits for 1.0/x (v_div_scale, v_rcp, two Newton steps, v_div_fmas, v_div_fixup).
#include "vm_exec_asm.inc"

// UPR: updates per record — 2: (a, r, u) triples, the LU program; 3: (a, u) pairs, the triangular sweeps (schedule.hpp)
template <int NT, int SLOTS, int UPR = 2>
__device__ __attribute__((noinline)) void vm_run(const VmDev P, uint32_t row0, int lane) {
  static_assert(SLOTS == 4 || SLOTS == 6 || SLOTS == 8, "ring depths vm_exec_asm.inc is generated for");
  static_assert(UPR == 2 || (UPR == 3 && SLOTS == 4), "executor variants vm_exec_asm.inc is generated for");
  const uint64_t recs = reinterpret_cast<uint64_t>(P.recs);      // the same in every lane: move it to SGPRs
  const uint64_t base = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)recs) |
                        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(recs >> 32)) << 32);
  // row0: the wave's first record row (P.wave_base[wave]; the kernel fetched it when it started — read here it was a memory round
  // trip in front of the first table load, on every call)
  uint32_t va = row0 * 2048u + (uint32_t)lane * 16u;     // this lane's first record (planar rows: lo plane, hi plane + 1024), bytes
  uint32_t vb = va + 4096u;                                                          // rows +2, +3
  int rounds = __builtin_amdgcn_readfirstlane(P.nrounds);
  double acc, a1A, r1A, u1A, a2A, r2A, u2A, a1B, r1B, u1B, a2B, r2B, u2B, sc;
  uint32_t tg, ax, t, flA, flB, tmp;
  uint64_t sv, sm;
#define MISTRA_VM_OPERANDS                                                                                                                             \
  [acc] "=&v"(acc), [a1A] "=&v"(a1A), [r1A] "=&v"(r1A), [u1A] "=&v"(u1A), [a2A] "=&v"(a2A), [r2A] "=&v"(r2A), [u2A] "=&v"(u2A), [a1B] "=&v"(a1B),      \
      [r1B] "=&v"(r1B), [u1B] "=&v"(u1B), [a2B] "=&v"(a2B), [r2B] "=&v"(r2B), [u2B] "=&v"(u2B), [sc] "=&v"(sc), [tg] "=&v"(tg), [ax] "=&v"(ax),          \
      [t] "=&v"(t), [flA] "=&s"(flA), [flB] "=&s"(flB), [tmp] "=&s"(tmp), [sv] "=&s"(sv), [sm] "=&s"(sm), [rounds] "+s"(rounds), [va] "+v"(va),         \
      [vb] "+v"(vb)
  if constexpr (UPR == 3) {
    asm volatile(MISTRA_VM_ASM_N4_SWEEP : MISTRA_VM_OPERANDS : [base] "s"(base) : "memory", "vcc", "scc", MISTRA_VM_CLOBBER_N4);
  } else if constexpr (SLOTS == 8) {
    uint32_t vc = va + 8192u, vd = va + 12288u;                                      // rows +4, +5 and +6, +7
    asm volatile(MISTRA_VM_ASM_N8 : MISTRA_VM_OPERANDS, [vc] "+v"(vc), [vd] "+v"(vd) : [base] "s"(base) : "memory", "vcc", "scc", MISTRA_VM_CLOBBER_N8);
  } else if constexpr (SLOTS == 6) {
    uint32_t vc = va + 8192u;
    asm volatile(MISTRA_VM_ASM_N6 : MISTRA_VM_OPERANDS, [vc] "+v"(vc) : [base] "s"(base) : "memory", "vcc", "scc", MISTRA_VM_CLOBBER_N6);
  } else {
    asm volatile(MISTRA_VM_ASM_N4 : MISTRA_VM_OPERANDS : [base] "s"(base) : "memory", "vcc", "scc", MISTRA_VM_CLOBBER_N4);
  }
#undef MISTRA_VM_OPERANDS
}

// ---- tail chain of the triangular solves (schedule.hpp: TailSolve), run by ONE wave: lane l holds rows h+l and h+64+l of the
//      solution in registers (x[0], x[1]); lane row r (16 lanes) of register q = 16x16 block row 4q + r of the tail triangle.
//      Matrix entries are gathered from LDS through per-column index tables streamed through the look-ahead ring (one 16-byte
//      slot = 4 columns, a 16-column block = 4 slots; the backward tables follow the forward ones through the same ring, so the
//      stream is primed once per solve).  Per block column J (forward; backward runs the blocks and the columns the other way):
//        1. the block's own 16 unknowns, in their lane row: x(i) -= L(i,j) x(j), j ascending — the pivot value reaches the other
//           lanes of its lane row inside the multiply-add itself (DPP row_newbcast on v_fmac_f64: measured 10 cycles per dependent
//           step with one wait state, 14 with the two the ISA manual asks for; the v_readlane -> SGPR -> VALU round trip that all
//           columns but the dense block's forward ones took until round 3 costs ~57);
//        2. the 16 finished values go to all four lane rows (gfx950 v_permlane16_swap / v_permlane32_swap);
//        3. the rows below take their 16 terms, j ascending, again by DPP: the lane rows below in the same register, and every
//           lane row of the other register.
//      Every row still receives its terms in ascending (backward: descending) column order with the operands and the fused
//      multiply-add of the column-by-column chain the emulator states (tests/emu/schedule_emu.cpp): the results are bit-identical.
//      Operands that are not in the pattern (and every diagonal) point at the 0.0 cell.
template <int ROW>
__device__ __forceinline__ double lane_row_to_all(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
  // permlane16_swap(a, b): a.row1 <-> b.row0, a.row3 <-> b.row2.   From a = b = v: a = [v0 v0 v2 v2], b = [v1 v1 v3 v3]
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const uint32_t a = l16[ROW & 1], b = h16[ROW & 1];
  // permlane32_swap(a, b): a.upper32 <-> b.lower32.   From a = b: a = [lower lower], b = [upper upper]
  const auto l32 = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  const auto h32 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  return __builtin_bit_cast(double, (uint64_t)l32[(ROW >> 1) & 1] | ((uint64_t)h32[(ROW >> 1) & 1] << 32));
}
// The phases are ONE asm statement each: between separate statements the compiler adds wait states of its own.  Step j of the
// diagonal block needs no lane selection: every operand that is not a strictly-lower (forward) / strictly-upper (backward) entry of
// the block points at the 0.0 cell, so the lanes i <= j (i >= j) of the lane row add an exact zero, as the column-by-column chain
// of the emulator does; row_mask confines the writes to the block's lane row.  The DPP read of the value the previous step wrote
// needs two wait states, the hardware does not interlock it (tools/ubench/dpp.hip: 10 cycles per dependent step with one, wrong
// with none).
#define MISTRA_TAIL_COPS                                                                                                          \
  [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]),     \
      [c8] "v"(c[8]), [c9] "v"(c[9]), [c10] "v"(c[10]), [c11] "v"(c[11]), [c12] "v"(c[12]), [c13] "v"(c[13]), [c14] "v"(c[14]), [c15] "v"(c[15])
#define MISTRA_DIAG_STEP(J, RM) "s_nop 1\n\tv_fmac_f64_dpp %[x], -%[x], %[c" #J "] row_newbcast:" #J " row_mask:" RM " bank_mask:0xf\n\t"
#define MISTRA_DIAG_FWD(RM)                                                                                                     \
  asm volatile(MISTRA_DIAG_STEP(0, RM) MISTRA_DIAG_STEP(1, RM) MISTRA_DIAG_STEP(2, RM) MISTRA_DIAG_STEP(3, RM) MISTRA_DIAG_STEP(4, RM)   \
               MISTRA_DIAG_STEP(5, RM) MISTRA_DIAG_STEP(6, RM) MISTRA_DIAG_STEP(7, RM) MISTRA_DIAG_STEP(8, RM) MISTRA_DIAG_STEP(9, RM)   \
               MISTRA_DIAG_STEP(10, RM) MISTRA_DIAG_STEP(11, RM) MISTRA_DIAG_STEP(12, RM) MISTRA_DIAG_STEP(13, RM) MISTRA_DIAG_STEP(14, RM) \
               "s_nop 0"                                                                                                        \
               : [x] "+v"(x) : MISTRA_TAIL_COPS)
#define MISTRA_DIAG_BWD(RM)                                                                                                     \
  asm volatile(MISTRA_DIAG_STEP(15, RM) MISTRA_DIAG_STEP(14, RM) MISTRA_DIAG_STEP(13, RM) MISTRA_DIAG_STEP(12, RM) MISTRA_DIAG_STEP(11, RM) \
               MISTRA_DIAG_STEP(10, RM) MISTRA_DIAG_STEP(9, RM) MISTRA_DIAG_STEP(8, RM) MISTRA_DIAG_STEP(7, RM) MISTRA_DIAG_STEP(6, RM)     \
               MISTRA_DIAG_STEP(5, RM) MISTRA_DIAG_STEP(4, RM) MISTRA_DIAG_STEP(3, RM) MISTRA_DIAG_STEP(2, RM) MISTRA_DIAG_STEP(1, RM)      \
               "s_nop 0"                                                                                                        \
               : [x] "+v"(x) : MISTRA_TAIL_COPS)
#define MISTRA_UPD_STEP(J, ROWMASK) "v_fmac_f64_dpp %[x], -%[xb], %[c" #J "] row_newbcast:" #J " row_mask:" ROWMASK " bank_mask:0xf\n\t"
#define MISTRA_UPDATE_FWD(ROWMASK)                                                                                             \
  asm volatile("s_nop 1\n\t"                                                                                                   \
               MISTRA_UPD_STEP(0, ROWMASK) MISTRA_UPD_STEP(1, ROWMASK) MISTRA_UPD_STEP(2, ROWMASK) MISTRA_UPD_STEP(3, ROWMASK)   \
               MISTRA_UPD_STEP(4, ROWMASK) MISTRA_UPD_STEP(5, ROWMASK) MISTRA_UPD_STEP(6, ROWMASK) MISTRA_UPD_STEP(7, ROWMASK)   \
               MISTRA_UPD_STEP(8, ROWMASK) MISTRA_UPD_STEP(9, ROWMASK) MISTRA_UPD_STEP(10, ROWMASK) MISTRA_UPD_STEP(11, ROWMASK) \
               MISTRA_UPD_STEP(12, ROWMASK) MISTRA_UPD_STEP(13, ROWMASK) MISTRA_UPD_STEP(14, ROWMASK)                            \
               "v_fmac_f64_dpp %[x], -%[xb], %[c15] row_newbcast:15 row_mask:" ROWMASK " bank_mask:0xf"                         \
               : [x] "+v"(x) : [xb] "v"(xb), MISTRA_TAIL_COPS)
#define MISTRA_UPDATE_BWD(ROWMASK)                                                                                             \
  asm volatile("s_nop 1\n\t"                                                                                                   \
               MISTRA_UPD_STEP(15, ROWMASK) MISTRA_UPD_STEP(14, ROWMASK) MISTRA_UPD_STEP(13, ROWMASK) MISTRA_UPD_STEP(12, ROWMASK) \
               MISTRA_UPD_STEP(11, ROWMASK) MISTRA_UPD_STEP(10, ROWMASK) MISTRA_UPD_STEP(9, ROWMASK) MISTRA_UPD_STEP(8, ROWMASK)   \
               MISTRA_UPD_STEP(7, ROWMASK) MISTRA_UPD_STEP(6, ROWMASK) MISTRA_UPD_STEP(5, ROWMASK) MISTRA_UPD_STEP(4, ROWMASK)     \
               MISTRA_UPD_STEP(3, ROWMASK) MISTRA_UPD_STEP(2, ROWMASK) MISTRA_UPD_STEP(1, ROWMASK)                                 \
               "v_fmac_f64_dpp %[x], -%[xb], %[c0] row_newbcast:0 row_mask:" ROWMASK " bank_mask:0xf"                           \
               : [x] "+v"(x) : [xb] "v"(xb), MISTRA_TAIL_COPS)

// step 1 of a block column for the lane row ROW of x
template <int ROW, bool BACKWARD>
__device__ __forceinline__ void tail_diag(double& x, const double (&c)[16]) {
  if constexpr (!BACKWARD) {
    if constexpr (ROW == 0) MISTRA_DIAG_FWD("0x1");
    else if constexpr (ROW == 1) MISTRA_DIAG_FWD("0x2");
    else if constexpr (ROW == 2) MISTRA_DIAG_FWD("0x4");
    else MISTRA_DIAG_FWD("0x8");
  } else {
    if constexpr (ROW == 0) MISTRA_DIAG_BWD("0x1");
    else if constexpr (ROW == 1) MISTRA_DIAG_BWD("0x2");
    else if constexpr (ROW == 2) MISTRA_DIAG_BWD("0x4");
    else MISTRA_DIAG_BWD("0x8");
  }
}
// step 3: the lane rows selected by MASK (a 4-bit row mask) take the block column's 16 terms from the broadcast values xb
template <int MASK, bool BACKWARD>
__device__ __forceinline__ void tail_update(double& x, const double xb, const double (&c)[16]) {
  static_assert(MASK >= 1 && MASK <= 15, "row mask");
#define MISTRA_UPD_CASE(M, S) else if constexpr (MASK == M) { if constexpr (BACKWARD) MISTRA_UPDATE_BWD(S); else MISTRA_UPDATE_FWD(S); }
  if constexpr (MASK == 15) { if constexpr (BACKWARD) MISTRA_UPDATE_BWD("0xf"); else MISTRA_UPDATE_FWD("0xf"); }
  MISTRA_UPD_CASE(14, "0xe") MISTRA_UPD_CASE(12, "0xc") MISTRA_UPD_CASE(8, "0x8") MISTRA_UPD_CASE(1, "0x1") MISTRA_UPD_CASE(3, "0x3") MISTRA_UPD_CASE(7, "0x7")
  else static_assert(MASK == 15, "row mask not instantiated");
#undef MISTRA_UPD_CASE
}
#undef MISTRA_DIAG_STEP
#undef MISTRA_DIAG_FWD
#undef MISTRA_DIAG_BWD
#undef MISTRA_UPD_STEP
#undef MISTRA_UPDATE_FWD
#undef MISTRA_UPDATE_BWD
#undef MISTRA_TAIL_COPS

// One block column, in two halves so that the operand gathers of block b + 1 are in flight while block b computes (a lone wave
// would otherwise sit out the table wait and an LDS round trip in front of every block: measured, the chain then took as long
// as the column-by-column one).  w[0..3]: the block's four table slots (16 columns in chain order: ascending forward,
// descending backward); a word holds the Ghimj cell of (row h + lane, column) in its low half and of (row h + 64 + lane,
// column) in its high half.  (vm_ring_take_cells turns them into LDS addresses as they come out of the ring; a[i]: column i of the
// block in TABLE order, the chain's step j is table column j forward and 15 - j backward.)
// NEXT: takes the next block's table slots and issues the gathers of its operands (into c and w, free by then) — placed between
// the block's last use of c and the other register's update, whose 16 dependent steps cover the gathers' LDS round trip.
// ad: LDS addresses of the other register's operands of this block, in table order (taken from the ring with the block's own).
template <int R, int J, bool BACKWARD, bool LAST, class NEXT>
__device__ __forceinline__ void tail_block(double (&xr)[R], double (&c)[16], const uint32_t (&ad)[16], NEXT&& next) {
  constexpr int REG = J / 4, ROW = J % 4;
  constexpr int OTHER = BACKWARD ? 0 : R - 1;                    // the other register's rows lie wholly below (forward) / above (backward) the block
  constexpr bool HAS_OTHER = !LAST && R == 2 && REG != OTHER;
  double d[16];      // ... their operands: gathered here, used three phases further down
  if constexpr (HAS_OTHER) {
#pragma unroll
    for (int j = 0; j < 16; j++) d[j] = lds_ld(ad[BACKWARD ? 15 - j : j]);
  }
  double x = xr[REG];
  tail_diag<ROW, BACKWARD>(x, c);                                  // 1.
  if constexpr (!LAST) {
    const double xb = lane_row_to_all<ROW>(x);                     // 2.
    constexpr int same = BACKWARD ? (1 << ROW) - 1 : (0xF << (ROW + 1)) & 0xF;      // 3. lane rows above / below in the same register
    if constexpr (same != 0) tail_update<same, BACKWARD>(x, xb, c);
    xr[REG] = x;
    if constexpr (HAS_OTHER) {
      double y = xr[OTHER];
      asm volatile("" : "+v"(y), "+v"(d[0]), "+v"(d[15]));      // (d and y are in registers before the next block's table words overwrite w)
      next();
      tail_update<15, BACKWARD>(y, xb, d);
      xr[OTHER] = y;
    } else {
      next();
    }
  } else {
    xr[REG] = x;
    next();
  }
}

// FWD_BLOCK0: first 16-column block of the forward chain.  0: the whole forward chain; 4R: none (the vector has been
// forward-swept already — stage 1, inside the LU program); 4 of R = 2 with the dense tail block: only the block's own columns,
// whose rows the LU program leaves without exactly those terms (schedule.cpp: lu_entries, dense_h)
template <int R, int FWD_BLOCK0, bool LOW>
__device__ __attribute__((noinline)) void tail_solve(const TailDev T, uint32_t xb, uint32_t rb, int lane) {
  constexpr int NB = 4 * R;                      // 16-column blocks of the tail triangle
  constexpr int NF = NB - FWD_BLOCK0;            // ... of them in the forward chain
  constexpr int NS = NF + NB;                    // blocks of the whole solve, forward then backward
  static_assert(FWD_BLOCK0 >= 0 && FWD_BLOCK0 <= NB && NS <= 16, "first forward block");
  double x[R], rd[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    x[r] = lds_ld(xb + 8 * (r * 64 + lane));
    rd[r] = lds_ld(rb + 8 * (r * 64 + lane));          // R(k) = 1/U(k,k), published by the LU program
  }
  // table rows: one u32x4 per lane and group of 4 columns, 1 KiB per row; the forward tables from block FWD_BLOCK0 on, then the
  // backward ones; past the last block: the tables' slack rows.  Row bases in scalar registers, the lane's offset in one vector register.
  const uint64_t tf = ring_base(T.fwd + FWD_BLOCK0 * 4 * 64 * 4).b0, tb = ring_base(T.bwd).b0;
  const uint32_t voff = 16u * (uint32_t)lane;
  uint32_t three = 3;
  asm volatile("" : "+v"(three));      // (held in one register for the whole solve)
#define MISTRA_TAIL_LOADS(S, K0)                                                                                          \
  {                                                                                                                      \
    const uint64_t base = (S) < NF ? tf + (uint64_t)(S) * 4096u : tb + (uint64_t)((S) - NF) * 4096u;                     \
    vm_ring_load_at<LOW, K0, 0>(base, voff); vm_ring_load_at<LOW, K0 + 1, 1024>(base, voff);                             \
    vm_ring_load_at<LOW, K0 + 2, 2048>(base, voff); vm_ring_load_at<LOW, K0 + 3, 3072>(base, voff);                      \
  }
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // nothing of the caller's may sit between the counted loads
  MISTRA_TAIL_LOADS(0, 0) MISTRA_TAIL_LOADS(1, 4)
  // block S of the stream (S < NF: forward block FWD_BLOCK0 + S; else backward block NB - 1 - (S - NF)): its four slots (blocks
  // alternate between the ring's halves, refilled with block S + 2's) come out of the ring as LDS addresses — of the block's own
  // register's operands (gathered at once, into c) and, where the other register takes part, of its operands (ad, gathered by tail_block)
  double c[16];
  uint32_t ad[16];
#define MISTRA_TAIL_GATHER(S)                                                                                             \
  if constexpr ((S) < NS) {                                                                                              \
    constexpr int K0 = ((S) & 1) * 4;                                                                                    \
    constexpr bool BW = (S) >= NF;                                                                                       \
    constexpr int J = BW ? (NB - 1 - ((S) - NF) + NB) % NB : (FWD_BLOCK0 + (S)) % NB;                                    \
    constexpr int REG = J / 4, OTHER = BW ? 0 : R - 1;                                                                   \
    constexpr bool LAST = BW ? (S) == NS - 1 : FWD_BLOCK0 + (S) == NB - 1;                                               \
    constexpr bool HAS_OTHER = !LAST && R == 2 && REG != OTHER;                                                          \
    constexpr bool LO = REG == 0 || (HAS_OTHER && OTHER == 0), HI = REG == 1 || (HAS_OTHER && OTHER == 1);               \
    uint32_t ac[16];                                                                                                     \
    uint32_t* const alo = REG == 0 ? ac : ad;                                                                            \
    uint32_t* const ahi = REG == 1 ? ac : ad;                                                                            \
    vm_ring_take_cells<LOW, K0, 7, LO, HI>(alo, ahi, three);         vm_ring_take_cells<LOW, K0 + 1, 6, LO, HI>(alo + 4, ahi + 4, three);   \
    vm_ring_take_cells<LOW, K0 + 2, 5, LO, HI>(alo + 8, ahi + 8, three); vm_ring_take_cells<LOW, K0 + 3, 4, LO, HI>(alo + 12, ahi + 12, three); \
    MISTRA_TAIL_LOADS((S) + 2, K0)                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 16; j++) c[j] = lds_ld(ac[BW ? 15 - j : j]);                                   \
  }
  // ... and its arithmetic; between the forward and the backward half: x = R .* x (the backward half runs on the row-scaled
  // triangle U' = D^-1 U that the factorisation leaves in the tail block — schedule.cpp: lu_entries; dense_lu — so there is
  // no quotient on the serial chain)
#define MISTRA_TAIL_STEP(S)                                                                                               \
  if constexpr ((S) < NS) {                                                                                              \
    if constexpr ((S) == NF) {                                                                                           \
      _Pragma("unroll") for (int r = 0; r < R; r++) x[r] = x[r] * rd[r];                                                 \
    }                                                                                                                    \
    auto next = [&]() { MISTRA_TAIL_GATHER((S) + 1) };                                                                   \
    if constexpr ((S) < NF) tail_block<R, (FWD_BLOCK0 + (S)) % NB, false, FWD_BLOCK0 + (S) == NB - 1>(x, c, ad, next);   \
    else tail_block<R, (NB - 1 - ((S) - NF) + NB) % NB, true, (S) == NS - 1>(x, c, ad, next);                            \
  }
  MISTRA_TAIL_GATHER(0)
  MISTRA_TAIL_STEP(0) MISTRA_TAIL_STEP(1) MISTRA_TAIL_STEP(2) MISTRA_TAIL_STEP(3) MISTRA_TAIL_STEP(4) MISTRA_TAIL_STEP(5) MISTRA_TAIL_STEP(6) MISTRA_TAIL_STEP(7)
  MISTRA_TAIL_STEP(8) MISTRA_TAIL_STEP(9) MISTRA_TAIL_STEP(10) MISTRA_TAIL_STEP(11) MISTRA_TAIL_STEP(12) MISTRA_TAIL_STEP(13) MISTRA_TAIL_STEP(14) MISTRA_TAIL_STEP(15)
#undef MISTRA_TAIL_STEP
#undef MISTRA_TAIL_GATHER
#undef MISTRA_TAIL_LOADS
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // the look-ahead loads past the stream's end have landed
#pragma unroll
  for (int r = 0; r < R; r++) lds_st(xb + 8 * (r * 64 + lane), x[r]);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
  return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}

// FWD_FROM: first 64-row block of the forward chain.  0: the whole forward chain; R: none (the vector has been forward-swept
// already — stage 1, inside the LU program).
// The column-by-column form of the chain, kept for the kernels held to 128 registers (aer, gas): the pivot value travels by
// v_readlane (~57 cycles per column); the block form above needs 16 operands per lane in registers at once, more than those
// kernels' functions have below their look-ahead ring.
template <int R, int FWD_FROM, bool LOW>
__device__ __attribute__((noinline)) void tail_solve_columns(const TailDev T, uint32_t xb, uint32_t rb, int lane) {
  constexpr int FWD_END = R;
  constexpr bool FORWARD = FWD_FROM < FWD_END;
  constexpr uint32_t mb = 0;      // M starts at LDS address 0 (checked at kernel entry); xb, rb: LDS addresses of the tail of XS and R
  double x[R], rd[R];
  const uint32_t voff = 16u * (uint32_t)lane;      // the lane's u32x4 within a table row
  static_assert(R == 1 || R == 2, "one or two 64-row registers");
  uint32_t three = 3;
  asm volatile("" : "+v"(three));
#pragma unroll
  for (int r = 0; r < R; r++) x[r] = lds_ld(xb + 8 * (r * 64 + lane));
  // Matrix entries of a 4-column group are read from LDS one group AHEAD of the chain that uses them (two register
  // buffers, alternating): a lone wave would otherwise expose the LDS latency once per group.  Every row slot is read
  // for every group (rows that take no part point at the 0.0 cell); the arithmetic loops keep their exact row ranges.
  // (the group's table words come out of the ring as LDS addresses, vm_ring_take_cells: low halves = register 0's rows, high = register 1's)
#define MISTRA_TAIL_OPERANDS(BUF, K, PENDING)                                           \
  {                                                                                     \
    uint32_t alo[4], ahi[4];                                                            \
    vm_ring_take_cells<LOW, K, PENDING, true, R == 2>(alo, ahi, three);                 \
    _Pragma("unroll") for (int c = 0; c < 4; c++) {                                     \
      BUF[c][0] = lds_ld(mb + alo[c]);                                                  \
      if constexpr (R == 2) BUF[c][1] = lds_ld(mb + ahi[c]);                            \
    }                                                                                   \
  }
  // The compiler would otherwise park the off-chain row updates (and their operands) until the row is next read, sixty
  // columns later: pin every group's results to the end of its group.
#define MISTRA_TAIL_PIN _Pragma("unroll") for (int r = 0; r < R; r++) asm volatile("" : "+v"(x[r]));
  double opa[4][R], opb[4][R];
  // ---- forward: for every tail column q ascending:  x(i) -= L(i,q) * x(q)  for the tail rows i > q
  if constexpr (FORWARD) {
    RingBase tp = ring_base(T.fwd + FWD_FROM * 16 * 64 * 4);      // 16 groups of 4 columns per block, one u32x4 per lane and group
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    vm_ring_load_s<LOW, 0, 0>(tp, voff); vm_ring_load_s<LOW, 1, 1>(tp, voff); vm_ring_load_s<LOW, 2, 2>(tp, voff); vm_ring_load_s<LOW, 3, 3>(tp, voff);
    vm_ring_load_s<LOW, 4, 4>(tp, voff); vm_ring_load_s<LOW, 5, 5>(tp, voff); vm_ring_load_s<LOW, 6, 6>(tp, voff); vm_ring_load_s<LOW, 7, 7>(tp, voff);
    ring_advance(tp, kRingSlots * 1024);
    MISTRA_TAIL_OPERANDS(opa, 0, 7)
    vm_ring_load_s<LOW, 0, 0>(tp, voff);
#pragma unroll
    for (int rq = FWD_FROM; rq < FWD_END; rq++) {
      for (int gb = 0; gb < 16; gb += kRingSlots) {
#define MISTRA_TAIL_FWD(K, CUR, NXT)                                                    \
        {                                                                               \
          MISTRA_TAIL_OPERANDS(NXT, (K + 1) % kRingSlots, 7)                                 \
          if constexpr (K + 1 == kRingSlots) ring_advance(tp, kRingSlots * 1024);            \
          vm_ring_load_s<LOW, (K + 1) % kRingSlots, (K + 1) % kRingSlots>(tp, voff);         \
          _Pragma("unroll") for (int c = 0; c < 4; c++) {                               \
            const double xq = readlane_f64(x[rq], 4 * (gb + K) + c);                    \
            _Pragma("unroll") for (int r = rq; r < R; r++) x[r] = __builtin_fma(-CUR[c][r], xq, x[r]); \
          }                                                                             \
          MISTRA_TAIL_PIN                                                               \
        }
        MISTRA_TAIL_FWD(0, opa, opb) MISTRA_TAIL_FWD(1, opb, opa) MISTRA_TAIL_FWD(2, opa, opb) MISTRA_TAIL_FWD(3, opb, opa)
        MISTRA_TAIL_FWD(4, opa, opb) MISTRA_TAIL_FWD(5, opb, opa) MISTRA_TAIL_FWD(6, opa, opb) MISTRA_TAIL_FWD(7, opb, opa)
#undef MISTRA_TAIL_FWD
      }
    }
  }
  // ---- backward, on the row-scaled triangle U' = D^-1 U that the LU program's last phase leaves in the tail block
  //      (schedule.cpp: lu_entries): x = R .* x, then for every tail column q descending  x(i) -= U'(i,q) * x(q)  for the
  //      tail rows i < q.  No quotient on the serial chain: per column it is readlane -> multiply -> subtract.
  {
    RingBase tp = ring_base(T.bwd);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    vm_ring_load_s<LOW, 0, 0>(tp, voff); vm_ring_load_s<LOW, 1, 1>(tp, voff); vm_ring_load_s<LOW, 2, 2>(tp, voff); vm_ring_load_s<LOW, 3, 3>(tp, voff);
    vm_ring_load_s<LOW, 4, 4>(tp, voff); vm_ring_load_s<LOW, 5, 5>(tp, voff); vm_ring_load_s<LOW, 6, 6>(tp, voff); vm_ring_load_s<LOW, 7, 7>(tp, voff);
    ring_advance(tp, kRingSlots * 1024);
    // R(k) = 1/U(k,k), published by the LU program: fetched here, under the table's first round trip (held through the forward
    // chain it was the register that pushed this function into a callee-saved one, stored to scratch and reloaded on every call)
#pragma unroll
    for (int r = 0; r < R; r++) rd[r] = lds_ld(rb + 8 * (r * 64 + lane));
#pragma unroll
    for (int r = 0; r < R; r++) x[r] = x[r] * rd[r];
    MISTRA_TAIL_OPERANDS(opa, 0, 7)
    vm_ring_load_s<LOW, 0, 0>(tp, voff);
#pragma unroll
    for (int rq = R - 1; rq >= 0; rq--) {
      for (int gb = 0; gb < 16; gb += kRingSlots) {
#define MISTRA_TAIL_BWD(K, CUR, NXT)                                                    \
        {                                                                               \
          MISTRA_TAIL_OPERANDS(NXT, (K + 1) % kRingSlots, 7)                                 \
          if constexpr (K + 1 == kRingSlots) ring_advance(tp, kRingSlots * 1024);            \
          vm_ring_load_s<LOW, (K + 1) % kRingSlots, (K + 1) % kRingSlots>(tp, voff);         \
          _Pragma("unroll") for (int c = 0; c < 4; c++) {                               \
            const double xq = readlane_f64(x[rq], 63 - (4 * (gb + K) + c));             \
            _Pragma("unroll") for (int r = 0; r <= rq; r++) x[r] = __builtin_fma(-CUR[c][r], xq, x[r]); \
          }                                                                             \
          MISTRA_TAIL_PIN                                                               \
        }
        MISTRA_TAIL_BWD(0, opa, opb) MISTRA_TAIL_BWD(1, opb, opa) MISTRA_TAIL_BWD(2, opa, opb) MISTRA_TAIL_BWD(3, opb, opa)
        MISTRA_TAIL_BWD(4, opa, opb) MISTRA_TAIL_BWD(5, opb, opa) MISTRA_TAIL_BWD(6, opa, opb) MISTRA_TAIL_BWD(7, opb, opa)
#undef MISTRA_TAIL_BWD
      }
    }
  }
#undef MISTRA_TAIL_OPERANDS
#undef MISTRA_TAIL_PIN
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
  for (int r = 0; r < R; r++) lds_st(xb + 8 * (r * 64 + lane), x[r]);
}

// ---- the gather-sum machine (schedule.hpp): out = c0*M[i0] + c1*M[i1] + ...  left to right, four terms per table row,
//      rows streamed through the look-ahead ring (4 rows in flight).  acc starts at -0.0 and padding terms are
//      (-0.0f)*(0.0 cell), so terms need no flags and the sums are bit-for-bit the flagged ones.  A row whose first
//      address carries GS_ROW_FLUSH completes the lane's current output: it is stored to the lane's next output cell in
//      LDS (out_addr, then every out_stride bytes) — the caller reads its own cells back, no barrier needed.
//      The instruction stream is generated (tools/gen_gsum_asm.py -> gsum_exec_asm.inc: why, and the pipeline, are described there).
#include "gsum_exec_asm.inc"
template <int NT, bool LOW>
__device__ __attribute__((noinline)) void gsum_run(const GsDev P, uint32_t row0, int rows, int lane, uint32_t out_addr, uint32_t out_stride) {
  // row0, rows: P.wave_base[wave] and P.rows[wave], fetched by the kernel when it started (see vm_run)
  int n = __builtin_amdgcn_readfirstlane(rows);
  const uint64_t recs = reinterpret_cast<uint64_t>(P.recs) + (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)row0) * 2048u;      // the wave's first row: 2 KiB per row
  const uint64_t b0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)recs) |
                      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(recs >> 32)) << 32);
  const uint64_t b1 = b0 + 4096u;
  uint32_t voff = (uint32_t)lane * 32u;      // two uint4 per lane and row: addresses, coefficients
  double acc = -0.0;
  const double mzero = -0.0;
  double xA0, xA1, xA2, xA3, xB0, xB1, xB2, xB3, cA0, cA1, cA2, cA3, cB0, cB1, cB2, cB3;
  uint32_t ad, flA, flB;
#define MISTRA_GSUM_OPERANDS                                                                                                                   \
  [acc] "+&v"(acc), [ad] "=&v"(ad), [xA0] "=&v"(xA0), [xA1] "=&v"(xA1), [xA2] "=&v"(xA2), [xA3] "=&v"(xA3), [xB0] "=&v"(xB0), [xB1] "=&v"(xB1),    \
      [xB2] "=&v"(xB2), [xB3] "=&v"(xB3), [cA0] "=&v"(cA0), [cA1] "=&v"(cA1), [cA2] "=&v"(cA2), [cA3] "=&v"(cA3), [cB0] "=&v"(cB0),               \
      [cB1] "=&v"(cB1), [cB2] "=&v"(cB2), [cB3] "=&v"(cB3), [flA] "=&s"(flA), [flB] "=&s"(flB), [n] "+s"(n), [voff] "+&v"(voff),               \
      [out] "+&v"(out_addr)                                                                                                                    \
      : [b0] "s"(b0), [b1] "s"(b1), [stride] "v"(out_stride), [mzero] "v"(mzero)
  if constexpr (LOW) asm volatile(MISTRA_GSUM_ASM_LOW : MISTRA_GSUM_OPERANDS : "memory", "scc", MISTRA_GSUM_CLOBBER_LOW);
  else asm volatile(MISTRA_GSUM_ASM_HIGH : MISTRA_GSUM_OPERANDS : "memory", "scc", MISTRA_GSUM_CLOBBER_HIGH);
#undef MISTRA_GSUM_OPERANDS
}

// ---- the factorisation's last act (schedule.hpp: ScaleProgram): M[tgt] *= M[aux] for two cells per 16-byte slot,
//      slots streamed through the look-ahead ring.  All cells are distinct: no ordering among lanes.
template <int NT, bool LOW>
__device__ __attribute__((noinline)) void scale_run(const ScaleDev P, int wave, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(P.nslots);
  gptr<u32x4> rp = G_(reinterpret_cast<const u32x4*>(P.recs)) + ((size_t)wave * (size_t)P.wave_slots * 64 + lane);
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // nothing of the caller's may sit between the counted loads
  vm_ring_load<LOW, 0>(rp);       vm_ring_load<LOW, 1>(rp + 64);  vm_ring_load<LOW, 2>(rp + 128); vm_ring_load<LOW, 3>(rp + 192);
  vm_ring_load<LOW, 4>(rp + 256); vm_ring_load<LOW, 5>(rp + 320); vm_ring_load<LOW, 6>(rp + 384); vm_ring_load<LOW, 7>(rp + 448);
  rp += kRingSlots * 64;
  for (int i = 0; i < n; i += kRingSlots) {
    // four slots at a time: sixteen gathers in flight, then the eight products (a lone gather-multiply-store chain per
    // slot would expose the LDS latency sixteen times over)
#define MISTRA_SCALE_GROUP(K)                                                        \
    {                                                                                \
      const u32x4 a0 = vm_ring_take<LOW, K>();     vm_ring_load<LOW, K>(rp + K * 64);         \
      const u32x4 a1 = vm_ring_take<LOW, K + 1>(); vm_ring_load<LOW, K + 1>(rp + (K + 1) * 64); \
      const u32x4 a2 = vm_ring_take<LOW, K + 2>(); vm_ring_load<LOW, K + 2>(rp + (K + 2) * 64); \
      const u32x4 a3 = vm_ring_take<LOW, K + 3>(); vm_ring_load<LOW, K + 3>(rp + (K + 3) * 64); \
      const double v0 = lds_ld(a0.x), f0 = lds_ld(a0.y), v1 = lds_ld(a0.z), f1 = lds_ld(a0.w); \
      const double v2 = lds_ld(a1.x), f2 = lds_ld(a1.y), v3 = lds_ld(a1.z), f3 = lds_ld(a1.w); \
      const double v4 = lds_ld(a2.x), f4 = lds_ld(a2.y), v5 = lds_ld(a2.z), f5 = lds_ld(a2.w); \
      const double v6 = lds_ld(a3.x), f6 = lds_ld(a3.y), v7 = lds_ld(a3.z), f7 = lds_ld(a3.w); \
      lds_st(a0.x, v0 * f0); lds_st(a0.z, v1 * f1); lds_st(a1.x, v2 * f2); lds_st(a1.z, v3 * f3); \
      lds_st(a2.x, v4 * f4); lds_st(a2.z, v5 * f5); lds_st(a3.x, v6 * f6); lds_st(a3.z, v7 * f7); \
    }
    MISTRA_SCALE_GROUP(0) MISTRA_SCALE_GROUP(4)
#undef MISTRA_SCALE_GROUP
    rp += kRingSlots * 64;
  }
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // drain the look-ahead loads before returning
}

// ---- dense tail block (schedule.hpp: DenseTail): the last 64 rows and columns of Ghimj, factorised in registers.
//      Tile (I, J) of the 4x4 grid of 16x16 tiles sits in wave 2I + (J>>1) as accumulator q = J&1 of
//      v_mfma_f64_16x16x4_f64: lane l, element r = row (l>>4) + 4r, column l&15 (cdna_hip_programming.md §3, verified by
//      tools/ubench).  Measured there: 64 cycles per MFMA and wave, the two waves of a SIMD overlapping fully; ~7 cycles per
//      dependent f64 operation, 72 for the IEEE reciprocal, ~30 for a same-wave LDS write -> read (which is why the
//      eliminating wave passes pivot rows lane to lane through LDS: v_readlane into an SGPR and back costs ~60).
#ifdef MISTRA_DIAG_STAMPS      // diagnostic builds only (tools/diag_dense.sh): cycles of wave 0 per section of dense_lu, summed over calls
__device__ unsigned long long g_dense_stamps[16];
#define MISTRA_STAMP(var) const unsigned long long var = clock64();
#define MISTRA_STAMP_ADD(slot, expr) if (threadIdx.x == 0) atomicAdd(&g_dense_stamps[slot], (unsigned long long)(expr));
#else
#define MISTRA_STAMP(var)
#define MISTRA_STAMP_ADD(slot, expr)
#endif
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f64x4 lds_f64x4;
__device__ __forceinline__ f64x4 lds_ld4(uint32_t addr) { return *(const lds_f64x4*)(uintptr_t)addr; }
__device__ __forceinline__ void lds_st4(uint32_t addr, f64x4 v) { *(lds_f64x4*)(uintptr_t)addr = v; }

// Ghimj slot (LDS byte address) of column c of a row's column range, from the row's table entry {first, absent lo, absent hi}
// (schedule.hpp: DenseTail); an absent entry reads as the 0.0 cell `zero`.
__device__ __forceinline__ uint32_t dense_slot(const u32x4 info, const uint32_t c, const uint32_t zero) {
  const uint64_t absent = (uint64_t)info.y | ((uint64_t)info.z << 32);
  const uint32_t idx = info.x + c - (uint32_t)__builtin_popcountll(absent & ((1ull << c) - 1ull));
  return ((absent >> c) & 1ull) ? zero : 8u * idx;
}
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
__device__ __forceinline__ u32x4 lds_ldu4(uint32_t addr) { return *(const lds_u32x4*)(uintptr_t)addr; }

// The finished entries of panel P go from the panel buffers to their Ghimj slots: waves 1 and 3 the L columns (two pivots
// each; wave 1 also the pivots' reciprocals), waves 2 and 5 the (row-scaled) U rows — waves that sit on other SIMDs than
// wave 0.  Runs while wave 0 eliminates inside panel P+1: the buffers of panel P are rewritten by panel P+2, two barriers
// later.
__device__ __forceinline__ bool dense_is_storer(const int wave) { return wave == 1 || wave == 2 || wave == 3 || wave == 5; }
template <class MT, int NT>
__device__ __forceinline__ void dense_store_panel(const int P, const int wave, const int lane) {
  constexpr uint32_t NNZ = MT::NNZ, NVAR = MT::NVAR, H = NVAR - 64, ZERO = 8u * (NNZ + NVAR);
  constexpr uint32_t PB = 8u * (uint32_t)LdsLayout<MT, NT>::PANEL, BC = PB + 8192u, INFO = 8u * (uint32_t)LdsLayout<MT, NT>::DINFO;
  const uint32_t PL = PB + (uint32_t)(P & 1) * 4096u, PU = PL + 2048u;
  const int J0P = 4 * P, khalf = (wave == 3 || wave == 5) ? 2 : 0;
  const f64x4 rcp = lds_ld4(BC + 1024u + 32u * (uint32_t)(P & 1));
  if (wave & 1) {
    if (wave != 5) {
      const f64x4 a = lds_ld4(PL + 32u * lane);
      const u32x4 mine = lds_ldu4(INFO + 16u * lane);
      const double a0 = khalf ? a[2] : a[0], a1 = khalf ? a[3] : a[1];
      const uint32_t at0 = dense_slot(mine, (uint32_t)(J0P + khalf), ZERO), at1 = dense_slot(mine, (uint32_t)(J0P + khalf + 1), ZERO);
      if (lane > J0P + khalf && at0 != ZERO) lds_st(at0, a0);                                               // L(H+lane, H+4P+k)
      if (lane > J0P + khalf + 1 && at1 != ZERO) lds_st(at1, a1);
      if (wave == 1 && lane >= J0P && lane < J0P + 4) {                                                     // R(k) = 1/U(k,k)
        const int k = lane - J0P;
        lds_st(8u * (NNZ + NVAR + 4u + H + (uint32_t)lane), k == 0 ? rcp[0] : k == 1 ? rcp[1] : k == 2 ? rcp[2] : rcp[3]);
      }
      return;
    }
  }
  {
    const f64x4 b = lds_ld4(PU + 32u * lane);
    const double b0 = khalf ? b[2] : b[0], b1 = khalf ? b[3] : b[1], r0 = khalf ? rcp[2] : rcp[0], r1 = khalf ? rcp[3] : rcp[1];
    const uint32_t at0 = dense_slot(lds_ldu4(INFO + 16u * (uint32_t)(J0P + khalf)), (uint32_t)lane, ZERO);
    const uint32_t at1 = dense_slot(lds_ldu4(INFO + 16u * (uint32_t)(J0P + khalf + 1)), (uint32_t)lane, ZERO);
    if (lane >= J0P + khalf && at0 != ZERO) lds_st(at0, lane == J0P + khalf ? b0 : b0 * r0);               // U'(H+4P+k, H+lane), the diagonal unscaled
    if (lane >= J0P + khalf + 1 && at1 != ZERO) lds_st(at1, lane == J0P + khalf + 1 ? b1 : b1 * r1);
  }
}

// One panel of four pivots, 4P .. 4P+3 of the block.
//   1. the owners of block column K = P/4 and block row K put the panel's four columns / rows into LDS: PL[row][4], PU[col][4]
//   2. wave 0 eliminates inside the panel, lane = row for PL and lane = column for PU, pivot by pivot in the order of
//      KppDecomp_x (gas.f:6160-6171), and leaves L, U and the pivots' reciprocals in the panel buffers
//   3. every tile that still has open rows and columns takes the rank-4 update with one MFMA (rows and columns up to the
//      panel's last pivot enter as exact zeros)
// During step 2 of the NEXT panel two otherwise idle waves store the finished L(i,j), U'(j,c) = U(j,c)*R(j), U(j,j), R(j)
// to their Ghimj slots, for the solves (dense_store_panel)
// The two panel buffers alternate, so that step 1 of the next panel can start while slower waves are still in step 3.
// The sixteen panels run as a loop of two copies of eight (MISTRA_DENSE_UNROLL below; fully unrolled, the block's factorisation
// was 39 KB of straight-line code executed once per decomposition, and instruction fetch, not arithmetic, set its pace); the tile
// registers a panel needs are picked with selects on wave-uniform conditions, which the compiler resolves where P's low bits are
// compile-time constants of the copy.
template <class MT, int NT>
__device__ __forceinline__ void dense_panel(f64x4& T0, f64x4& T1, const int P, const int wave, const int lane
#ifdef MISTRA_DIAG_STAMPS
                                            , unsigned long long (&acc)[5]
#endif
) {
  const int K = P >> 2, S = P & 3, J0P = 4 * P, K2 = (J0P + 4) >> 4;
  constexpr uint32_t PB = 8u * (uint32_t)LdsLayout<MT, NT>::PANEL, BC = PB + 8192u;
  const uint32_t PL = PB + (uint32_t)(P & 1) * 4096u, PU = PL + 2048u;
  const int I = wave >> 1, J0 = 2 * (wave & 1);
  const uint32_t lrow = (uint32_t)(lane >> 4), lcol = (uint32_t)(lane & 15);
  MISTRA_STAMP(ta)
  // (the two tiles are two named values and every choice among their registers is a select on a wave-uniform condition:
  //  an array indexed by K or S would be put in scratch memory, a global-memory round trip per access)
  if (I >= K && (wave & 1) == (K >> 1)) {
    const bool odd = K & 1;
    const double t0 = odd ? T1[0] : T0[0], t1 = odd ? T1[1] : T0[1], t2 = odd ? T1[2] : T0[2], t3 = odd ? T1[3] : T0[3];
    if ((int)(lcol >> 2) == S) {
      const uint32_t at = PL + 8u * ((16u * I + lrow) * 4u + (lcol & 3u));
      lds_st(at, t0); lds_st(at + 128u, t1); lds_st(at + 256u, t2); lds_st(at + 384u, t3);
    }
  }
  if (I == K) {
    if (J0 >= K) lds_st(PU + 8u * ((16u * J0 + lcol) * 4u + lrow), S == 0 ? T0[0] : S == 1 ? T0[1] : S == 2 ? T0[2] : T0[3]);
    if (J0 + 1 >= K) lds_st(PU + 8u * ((16u * (J0 + 1) + lcol) * 4u + lrow), S == 0 ? T1[0] : S == 1 ? T1[1] : S == 2 ? T1[2] : T1[3]);
  }
  lds_barrier();
  MISTRA_STAMP(tb)
  if (wave == 0) {
    // A lone wave issues one instruction every 4-7 cycles whatever it is: this block is kept to the eliminations themselves.
    // The panel's 4x4 diagonal block is read by EVERY lane (uniform addresses: broadcast reads) and factorised redundantly in
    // every lane: the dependent chain  pivot -> reciprocal -> multipliers -> next pivot  then runs in registers, without the
    // LDS round trip per pivot that handing rows from lane to lane would cost (~1000 -> ~500 cycles per panel).  The lanes'
    // own panel entries (lane = row: a, lane = column: b) take the same operations with the same operands: where they
    // coincide with the block's entries the values are bit-identical.
    f64x4 a = lds_ld4(PL + 32u * lane), b = lds_ld4(PU + 32u * lane);
    const f64x4 c0 = lds_ld4(PU + 32u * (uint32_t)J0P), c1 = lds_ld4(PU + 32u * (uint32_t)(J0P + 1));      // column j of the block: rows 0..3
    const f64x4 c2 = lds_ld4(PU + 32u * (uint32_t)(J0P + 2)), c3 = lds_ld4(PU + 32u * (uint32_t)(J0P + 3));
    f64x4 rcp;
    rcp[0] = 1.0 / c0[0];
    const double l10 = c0[1] * rcp[0], l20 = c0[2] * rcp[0], l30 = c0[3] * rcp[0];
    const double u01 = c1[0], u02 = c2[0], u03 = c3[0];
    const double d11 = __builtin_fma(-l10, u01, c1[1]), d12 = __builtin_fma(-l10, u02, c2[1]), d13 = __builtin_fma(-l10, u03, c3[1]);
    double d21 = __builtin_fma(-l20, u01, c1[2]), d22 = __builtin_fma(-l20, u02, c2[2]), d23 = __builtin_fma(-l20, u03, c3[2]);
    double d31 = __builtin_fma(-l30, u01, c1[3]), d32 = __builtin_fma(-l30, u02, c2[3]), d33 = __builtin_fma(-l30, u03, c3[3]);
    rcp[1] = 1.0 / d11;
    const double l21 = d21 * rcp[1], l31 = d31 * rcp[1];
    d22 = __builtin_fma(-l21, d12, d22); d23 = __builtin_fma(-l21, d13, d23);
    d32 = __builtin_fma(-l31, d12, d32); d33 = __builtin_fma(-l31, d13, d33);
    rcp[2] = 1.0 / d22;
    const double l32 = d32 * rcp[2];
    d33 = __builtin_fma(-l32, d23, d33);
    rcp[3] = 1.0 / d33;
    // lane = row: L(i, 4P+k) = (entry - earlier multipliers times the pivot rows' entries) * R(k), columns ascending
    a[0] = a[0] * rcp[0];
    a[1] = __builtin_fma(-a[0], u01, a[1]); a[2] = __builtin_fma(-a[0], u02, a[2]); a[3] = __builtin_fma(-a[0], u03, a[3]);
    a[1] = a[1] * rcp[1];
    a[2] = __builtin_fma(-a[1], d12, a[2]); a[3] = __builtin_fma(-a[1], d13, a[3]);
    a[2] = a[2] * rcp[2];
    a[3] = __builtin_fma(-a[2], d23, a[3]);
    a[3] = a[3] * rcp[3];
    // lane = column: U(4P+k, c) = entry - the row's multipliers times the earlier pivot rows, pivots ascending
    b[1] = __builtin_fma(-l10, b[0], b[1]); b[2] = __builtin_fma(-l20, b[0], b[2]); b[3] = __builtin_fma(-l30, b[0], b[3]);
    b[2] = __builtin_fma(-l21, b[1], b[2]); b[3] = __builtin_fma(-l31, b[1], b[3]);
    b[3] = __builtin_fma(-l32, b[2], b[3]);
    lds_st4(PL + 32u * lane, a);      // L(H+lane, H+4P+k) for lane > 4P+k
    lds_st4(PU + 32u * lane, b);      // U(H+4P+k, H+lane) for lane >= 4P+k
    if (lane == 0) lds_st4(BC + 1024u + 32u * (uint32_t)(P & 1), rcp);
  } else if (dense_is_storer(wave) && P > 0) {
    dense_store_panel<MT, NT>(P - 1, wave, lane);      // while wave 0 eliminates: the previous panel's entries go to their Ghimj slots
  }
  MISTRA_STAMP(tc)
  lds_barrier();
  MISTRA_STAMP(td)
  if (I >= K2) {      // (nothing is left open after the last panels: K2 = 4)
    // rows and columns up to the panel's last pivot are finished: their operand is an exact zero, whatever the buffers hold there
    const double aop = (int)(16u * I + lcol) >= J0P + 4 ? -lds_ld(PL + 8u * ((16u * I + lcol) * 4u + lrow)) : 0.0;
    if (J0 >= K2) {
      const double bop = (int)(16u * J0 + lcol) >= J0P + 4 ? lds_ld(PU + 8u * ((16u * J0 + lcol) * 4u + lrow)) : 0.0;
      T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, T0, 0, 0, 0);
    }
    if (J0 + 1 >= K2) {
      const double bop = (int)(16u * (J0 + 1) + lcol) >= J0P + 4 ? lds_ld(PU + 8u * ((16u * (J0 + 1) + lcol) * 4u + lrow)) : 0.0;
      T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, T1, 0, 0, 0);
    }
  }
  MISTRA_STAMP(te)
#ifdef MISTRA_DIAG_STAMPS
  acc[0] += tb - ta; acc[1] += tc - tb; acc[2] += td - tc; acc[3] += te - td; acc[4] += 1;
#endif
}

// The dense tail block after the LU program and the scaling pass: Schur steps, then the block's own factorisation.  Every
// Ghimj slot it touches is found by arithmetic on the row table in LDS: a global load issued here, in the middle of a cell's
// step, waits behind the table streams of the whole chip (measured: 13 000 cycles in front of the first MFMA).
template <class MT, int NT>
__device__ __attribute__((noinline)) void dense_lu(int lane) {
  static_assert(NT == 512 && MT::DENSE_ND == 64, "eight waves, two 16x16 tiles each");
  constexpr int KB = MT::DENSE_KB;
  constexpr uint32_t NNZ = MT::NNZ, NVAR = MT::NVAR, H = NVAR - 64, JM = H - 4 * KB, ZERO = 8u * (NNZ + NVAR);
  constexpr uint32_t INFO = 8u * (uint32_t)LdsLayout<MT, NT>::DINFO, RDIAG = 8u * (NNZ + NVAR + 4u);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t I = (uint32_t)(wave >> 1), J0 = 2u * (uint32_t)(wave & 1), lrow = (uint32_t)(lane >> 4), lcol = (uint32_t)(lane & 15);
  MISTRA_STAMP(t0)
  // ---- the block's slots as the LU program and the scaling pass leave them -> tiles
  f64x4 T0, T1;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const u32x4 info = lds_ldu4(INFO + 16u * (16u * I + lrow + 4u * r));
    T0[r] = lds_ld(dense_slot(info, 16u * J0 + lcol, ZERO));
    T1[r] = lds_ld(dense_slot(info, 16u * J0 + 16u + lcol, ZERO));
  }
  // ---- Schur steps: pivots jm .. h-1, four per MFMA, ascending: D -= W * U' with W the still unscaled L slots of the
  //      block's rows and U' the row-scaled U slots of the pivots' rows (schedule.hpp: DenseTail).  The operands' cells come from
  //      the host-made table in LDS, and everything is fetched in three batches (cells, values, then the MFMAs): taken step by
  //      step, every step waited out three dependent LDS round trips and ~60 instructions of row-table arithmetic (round 2:
  //      7 500 cycles for these 28 MFMAs).
  constexpr uint32_t SCH = 8u * (uint32_t)LdsLayout<MT, NT>::SCHUR;
  typedef __attribute__((address_space(3))) uint16_t lds_u16;
  const uint32_t wtab = SCH + 2u * ((uint32_t)I * KB * 64u + (uint32_t)lane);                                            // + 128 k
  const uint32_t utab = SCH + 2u * (4u * KB * 64u + (uint32_t)(wave & 1) * KB * 128u + (uint32_t)lane);                   // + 256 k (+ 128: second tile)
  uint32_t wslot[KB];
  double wl[KB];
  static_assert(KB % 2 == 0, "two batches of Schur steps");
#pragma unroll
  for (int half = 0; half < 2; half++) {      // two batches of KB/2 steps: one batch of all KB needed more registers than dense_lu can have without saving callee-saved ones
    constexpr int HB = KB / 2;
    uint32_t us0[HB], us1[HB];
#pragma unroll
    for (int kk = 0; kk < HB; kk++) {
      const int k = half * HB + kk;
      wslot[k] = 8u * (uint32_t)*(const lds_u16*)(uintptr_t)(wtab + 128u * k);
      us0[kk] = 8u * (uint32_t)*(const lds_u16*)(uintptr_t)(utab + 256u * k);
      us1[kk] = 8u * (uint32_t)*(const lds_u16*)(uintptr_t)(utab + 256u * k + 128u);
    }
    __builtin_amdgcn_sched_barrier(0);      // (the compiler's scheduler would otherwise weave the batches back into one chain per step)
    double u0[HB], u1[HB];
#pragma unroll
    for (int kk = 0; kk < HB; kk++) {
      wl[half * HB + kk] = lds_ld(wslot[half * HB + kk]);
      u0[kk] = lds_ld(us0[kk]);
      u1[kk] = lds_ld(us1[kk]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < HB; kk++) {
      T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-wl[half * HB + kk], u0[kk], T0, 0, 0, 0);
      T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-wl[half * HB + kk], u1[kk], T1, 0, 0, 0);
    }
  }
  // R(j) of the steps' pivots for the lane's own multipliers (below); the reads run beside the MFMAs
  double rj[KB / 2];
#pragma unroll
  for (int k2 = 0; k2 < KB / 2; k2++) rj[k2] = lds_ld(RDIAG + 8u * (JM + 4u * (2u * k2 + (uint32_t)(wave & 1)) + lrow));
  MISTRA_STAMP(t1)
  lds_barrier();      // both waves of a block row have read the unscaled slots
  {                   // L = W * R(j) for the solves; the two waves of a block row share the work (steps k even / k odd)
    static_assert(KB % 2 == 0, "the two waves of a block row take every other Schur step");
    if (wave & 1) {   // (wave-uniform branch: compile-time register indices on both sides)
#pragma unroll
      for (int k = 1; k < KB; k += 2)
        if (wslot[k] != ZERO) lds_st(wslot[k], wl[k] * rj[k / 2]);
    } else {
#pragma unroll
      for (int k = 0; k < KB; k += 2)
        if (wslot[k] != ZERO) lds_st(wslot[k], wl[k] * rj[k / 2]);
    }
  }
  MISTRA_STAMP(t2)
  MISTRA_STAMP_ADD(0, t1 - t0) MISTRA_STAMP_ADD(1, t2 - t1) MISTRA_STAMP_ADD(9, 1)
  // ---- the block's own factorisation
#ifndef MISTRA_DENSE_UNROLL      // panels per copy of the loop body: with 8 the choices among tile registers (panel within its block column, odd / even
#define MISTRA_DENSE_UNROLL 8      // block column) are made at compile time — 22 selects fewer per panel and wave; measured on one box: 1: 24 990, 2: 24 860, 4: 24 770,
#endif                            // 8: 25 210 timesteps/s (16 = the whole loop unrolled was slower in round 2: instruction fetch)
#ifndef MISTRA_DIAG_DENSE_PANELS      // timing diagnostics only (tools/diag_dense.sh): a library built with fewer panels computes garbage
#define MISTRA_DIAG_DENSE_PANELS 16
#endif
#ifdef MISTRA_DIAG_STAMPS
  unsigned long long acc[5] = {0, 0, 0, 0, 0};
  MISTRA_STAMP(tp0)
#pragma unroll 1
  for (int P = 0; P < MISTRA_DIAG_DENSE_PANELS; P++) dense_panel<MT, NT>(T0, T1, P, wave, lane, acc);
  MISTRA_STAMP(tp1)
  MISTRA_STAMP_ADD(2, acc[0]) MISTRA_STAMP_ADD(3, acc[1]) MISTRA_STAMP_ADD(4, acc[2]) MISTRA_STAMP_ADD(5, acc[3]) MISTRA_STAMP_ADD(8, acc[4]) MISTRA_STAMP_ADD(12, tp1 - tp0)
#else
#pragma unroll MISTRA_DENSE_UNROLL
  for (int P = 0; P < MISTRA_DIAG_DENSE_PANELS; P++) dense_panel<MT, NT>(T0, T1, P, wave, lane);
#endif
  if (dense_is_storer(wave)) dense_store_panel<MT, NT>(15, wave, lane);      // (behind the last panel's second barrier)
#ifdef MISTRA_DIAG_STAMPS
  lds_barrier();
  MISTRA_STAMP(tz)
  MISTRA_STAMP_ADD(11, tz - t0)
#endif
}

}  // namespace

// VARIANT: 0 = the product kernel; 1 = phase timing (MISTRA_CHEM_PROFILE, capi.cpp); 2 = first-step dump: the intermediate
// results of the FIRST attempt of the first step of every cell go to a.dump (layout: kernel_args.hpp), for the phase-level
// parity tests (tests/test_gpu_parity.py); the integration itself is untouched.
template <class MT, int NT, int VARIANT>
__global__ __launch_bounds__(NT, MT::WAVES_PER_SIMD) void ros3_integrate_kernel(const KernelArgs a) {
  constexpr bool PROF = VARIANT == 1, DUMP = VARIANT == 2;
  constexpr int NVAR = MT::NVAR, NFIX = MT::NFIX, NREACT = MT::NREACT, NNZ = MT::NNZ, NCONST = MT::NCONST;
  constexpr int NW = NT / 64;
  constexpr int SPT = (NVAR + NT - 1) / NT, RPT = (NREACT + NT - 1) / NT;
  constexpr int JPT = (MT::NJNZ + NT - 1) / NT, ZPT = (NNZ - MT::NJNZ + NT - 1) / NT;
  using L = LdsLayout<MT, NT>;

  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* const M = lds + L::M;
  double* const XS = M + NNZ;
  double* const X = lds + L::X;
  double* const AB = lds + L::AB;
  double* const JBp = lds + L::JB;      // Jac_SP's B products: inside the Ghimj area where it is large enough, else the A array (ros3_kernel.hpp)
  double* const red = lds + L::RED;
  int* const flags = reinterpret_cast<int*>(lds + L::FLAGS);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cell = blockIdx.x;
  if (cell >= a.ncell) return;
  if (lds_addr(lds) != 0u) {      // the VM and the tail chain address M by absolute LDS offsets
    if (t == 0) GM_(a.ierr)[cell] = -99;
    return;
  }

  // ---- where this wave's stream starts in each table program (and how long the gather-sum ones are): the same for the whole
  //      call, kept in scalar registers
  struct { uint32_t lu_row0, fwd_row0, bwd_row0, vdot_row0, jvs_row0; int vdot_rows, jvs_rows; } hdr;
  {
    const auto u = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    hdr.lu_row0 = u(G_(a.lu.wave_base)[wave]);
    hdr.fwd_row0 = u(G_(a.solve_head_fwd.wave_base)[wave]);
    hdr.bwd_row0 = u(G_(a.solve_head_bwd.wave_base)[wave]);
    hdr.vdot_row0 = u(G_(a.vdot.wave_base)[wave]);
    hdr.jvs_row0 = u(G_(a.jvs.wave_base)[wave]);
    hdr.vdot_rows = (int)u(G_(a.vdot.rows)[wave]);
    hdr.jvs_rows = (int)u(G_(a.jvs.rows)[wave]);
  }

  // ---- per-cell inputs: coalesced cell-major reads, once
  double y[SPT], rct[RPT];
#pragma unroll
  for (int q = 0; q < SPT; q++) {
    const int s = q * NT + t;
    y[q] = s < NVAR ? G_(a.var_in)[(size_t)cell * NVAR + s] : 0.0;
  }
  // (the register-starved kernels fetch them per use, like their factor words; MISTRA_RCT_RESIDENT_SLOTS of the RPT values per thread stay in
  //  registers all the same: every value kept is one global re-read per Fun / Jac_SP less — the re-reads that miss the L2 are what is left
  //  of aer's HBM-side excess)
#ifndef MISTRA_RCT_RESIDENT_SLOTS
#define MISTRA_RCT_RESIDENT_SLOTS 0
#endif
  constexpr bool RCT_PER_USE = MT::WAVES_PER_SIMD > MISTRA_RESIDENT_MAX_WPS || L::RCT_IN_LDS;
  constexpr int RCT_KEEP = !RCT_PER_USE ? RPT : (MISTRA_RCT_RESIDENT_SLOTS < RPT ? MISTRA_RCT_RESIDENT_SLOTS : RPT);
  auto load_rct = [&](bool opaque) {      // this thread's rate constants; opaque: the per-use fetch of the values that are not kept
    const double* rc = a.rconst;
    if (opaque) asm volatile("" : "+s"(rc));
    int tt = t;
    if (opaque) asm volatile("" : "+v"(tt));      // (addresses derived where they are used: hoisted out of the step loop they are registers that spill)
#pragma unroll
    for (int q = opaque ? RCT_KEEP : 0; q < RPT; q++) {
      const int r = q * NT + tt;
      if constexpr (L::RCT_IN_LDS) {      // (per use: out of the cell's copy in LDS, put there once, below)
        if (opaque) { rct[q] = r < NREACT ? lds_ld(8u * (uint32_t)(L::RCT + r)) : 0.0; continue; }
      }
      rct[q] = r < NREACT ? G_(rc)[(size_t)cell * NREACT + r] : 0.0;
    }
  };
  if constexpr (L::RCT_IN_LDS) {      // global -> LDS, once; read back by the same thread that wrote them (r = q*NT + t): no barrier involved
#pragma unroll
    for (int q = 0; q < RPT; q++) {
      const int r = q * NT + t;
      if (r < NREACT) lds[L::RCT + r] = G_(a.rconst)[(size_t)cell * NREACT + r];
      rct[q] = 0.0;
    }
  } else {
    load_rct(false);
  }
  // Ghimj slots this thread fills in ros_PrepareMatrix (static per mechanism): fetched once, not once per attempt
  // (two 16-bit positions per register)
  // RESIDENT: the kernel with 256 registers (tot) keeps these words for the whole call; the 128-register ones (aer, gas: four
  // waves per SIMD) fetch them where they are used — there they were spilled to scratch and came back from it one at a time.
  constexpr bool RESIDENT = MT::WAVES_PER_SIMD <= MISTRA_RESIDENT_MAX_WPS;
  uint32_t jpos[(JPT + 1) / 2], zpos[(ZPT + 1) / 2];
  auto load_pos = [&]() {
    const uint16_t* jp = a.jvs_pos;
    const uint16_t* zp = a.zero_pos;
    int t = threadIdx.x;
    if constexpr (!RESIDENT) asm volatile("" : "+s"(jp), "+s"(zp), "+v"(t));      // opaque: neither the loads nor the per-thread addresses are hoisted out of the step loop
#pragma unroll
    for (int q = 0; q < (JPT + 1) / 2; q++)
      jpos[q] = (uint32_t)G_(jp)[(2 * q) * NT + t] | ((2 * q + 1 < JPT ? (uint32_t)G_(jp)[(2 * q + 1) * NT + t] : (uint32_t)kPosNone) << 16);
#pragma unroll
    for (int q = 0; q < (ZPT + 1) / 2; q++)
      zpos[q] = (uint32_t)G_(zp)[(2 * q) * NT + t] | ((2 * q + 1 < ZPT ? (uint32_t)G_(zp)[(2 * q + 1) * NT + t] : (uint32_t)kPosNone) << 16);
  };
  if constexpr (RESIDENT) load_pos();
  // factor words of the products this thread forms in Fun (one per owned reaction): static per mechanism, kept in registers
  // for the whole integration.  The (up to three per reaction) words of Jac_SP's products are NOT: values that live across
  // the calls of the step loop need callee-saved registers, there are not enough of those, and the compiler's answer was to
  // spill nine of the twelve and reload them one at a time, each load's latency exposed.  jac() fetches them in one batch of
  // coalesced loads instead (48 KB table, L2-resident; measured: the batch lands in ~380 cycles, Jac_SP's products went from
  // 11 000 to 7 400 cycles per call, the kernel's scratch from 104 to 8 bytes per lane).
  // (tot, round 4: with the four words resident the kernel was 3 registers over what it can keep across its calls — one of these words and an
  //  LDS address went through scratch, reloaded right behind a barrier in every Fun, a memory round trip each; fetched per use they ride in
  //  the batch of Jac_SP's words that fun_jac issues anyway, and the kernel has no scratch)
#ifndef MISTRA_FFAC_RESIDENT
#define MISTRA_FFAC_RESIDENT 0
#endif
  constexpr bool FFAC_RESIDENT = RESIDENT && MISTRA_FFAC_RESIDENT;
  uint64_t ffac[RPT];
  auto load_ffac = [&]() {
    const uint64_t* ff = a.fun_fac;
    if constexpr (!FFAC_RESIDENT) asm volatile("" : "+s"(ff));
#pragma unroll
    for (int q = 0; q < RPT; q++) ffac[q] = G_(ff)[q * NT + t];
  };
  if constexpr (FFAC_RESIDENT) load_ffac();
  if (t < NFIX) X[NVAR + t] = G_(a.fix)[(size_t)cell * NFIX + t];
  if (t < NCONST) X[NVAR + NFIX + t] = G_(a.consts)[t];
  if constexpr (MT::DENSE_ND > 0) {      // the dense tail block's row table stays in LDS for the whole call
    if (t < 64) *(lds_u32x4*)(uintptr_t)(8u * (uint32_t)L::DINFO + 16u * (uint32_t)t) = G_(reinterpret_cast<const u32x4*>(a.dense.row_info))[t];
    for (int i = t; i < L::SCHUR_WORDS; i += NT)      // the Schur steps' operand cells (uint16 pairs)
      *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)(8u * (uint32_t)L::SCHUR + 4u * (uint32_t)i) = G_(reinterpret_cast<const uint32_t*>(a.dense.schur_cells))[i];
  }
  if (t == 0) {   // the VM's constant cells: 0.0 (padding update slots), 1.0 (neutral factor), -1.0 (partial-sum combine)
    M[NNZ + NVAR] = 0.0;
    M[NNZ + NVAR + 1] = 1.0;
    M[NNZ + NVAR + 3] = -1.0;
  }

  // optional phase timing (diagnostics only): cycles of wave 0 between phase boundaries, summed per cell
  constexpr bool profiling = PROF;       // a compile-time variant: sixteen 64-bit counters are not carried by the product kernel
  unsigned long long pc[kProfSlots] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = profiling ? clock64() : 0ull;
  const unsigned long long t_begin = t_last;
  auto lap = [&](int slot) {
    if constexpr (profiling) {
      const unsigned long long now = clock64();
      pc[slot] += now - t_last;
      t_last = now;
    }
  };

  // first-step dump (VARIANT 2): one block of doubles per cell, see DumpLayout in kernel_args.hpp
  bool dumping = DUMP;
  gptr_mut<double> dump = nullptr;
  if constexpr (DUMP) dump = GM_(a.dump) + (size_t)cell * (size_t)(5 * NVAR + 2 * NNZ + 2);
  auto dump_vec = [&](int at, const double (&v)[SPT]) {
    if constexpr (DUMP) {
      if (dumping) {
#pragma unroll
        for (int q = 0; q < SPT; q++) {
          const int s = q * NT + t;
          if (s < NVAR) dump[at + s] = v[q];
        }
      }
    }
  };
  auto dump_matrix = [&](int at) {      // Ghimj as it stands in LDS (every lane's stores are behind a barrier at the call sites)
    if constexpr (DUMP) {
      if (dumping)
        for (int i = t; i < NNZ; i += NT) dump[at + i] = M[i];
    }
  };

  // ---- Fun_x (gas.f:2043): X <- v; A(r) = RCT(r)*X*X*X; Vdot = signed sums of A
  static_assert(SPT <= 2 && JPT * NT <= NNZ, "output cells of the gather-sum machine: at most two species per thread, JVS sums inside Ghimj");
  // Fun_x's sums land in the thread's own cells of XS: species t and, where a thread owns two (one wavefront per cell: NT = 64 < NVAR), t + NT.
  // A thread whose second species does not exist parks that (empty) sum in the trash cell: the stride between a lane's outputs is per lane.
  auto vdot_out = [&](int tt, uint32_t& addr, uint32_t& stride) {
    constexpr uint32_t trash = 8u * (uint32_t)(NNZ + NVAR + 2);
    addr = tt < NVAR ? 8u * (uint32_t)(NNZ + tt) : trash;
    if constexpr (SPT == 1) stride = tt < NVAR ? 8u * (uint32_t)NT : 0u;
    else stride = tt + NT < NVAR ? 8u * (uint32_t)NT : trash - addr;
  };
  auto fun = [&](const double (&v)[SPT], double (&out)[SPT]) {
    if constexpr (!FFAC_RESIDENT) load_ffac();
    if constexpr (RCT_PER_USE) load_rct(true);
#pragma unroll
    for (int q = 0; q < SPT; q++) {
      const int s = q * NT + t;
      if (s < NVAR) X[s] = v[q];
    }
    lds_barrier();
    {
      // every factor read is issued before the first product is stored: X and the product array are one LDS object to the
      // compiler, so a read behind a store stays behind it, and product by product each one waited out its own LDS round trip
      double f0[RPT], f1[RPT], f2[RPT];
      uint32_t slot[RPT];
#pragma unroll
      for (int q = 0; q < RPT; q++) {
        uint64_t w = ffac[q];
        asm volatile("" : "+v"(w));   // decode here: hoisted out of the step loop, the derived addresses only spill
        f0[q] = X[w & 0xFFFFu];
        f1[q] = X[(w >> 16) & 0xFFFFu];
        f2[q] = X[(w >> 32) & 0xFFFFu];
        slot[q] = (uint32_t)(w >> 48);
      }
#pragma unroll
      for (int q = 0; q < RPT; q++) {
        double p = rct[q] * f0[q];
        p = p * f1[q];
        p = p * f2[q];
        AB[slot[q]] = p;            // a slot without a reaction (rct = 0) writes a spare cell: no branch
      }
    }
    lds_barrier();
    lap(15);
    // sums land in this thread's own cells of XS (free here: the solves copy their result out before Fun runs again);
    // a thread without a species parks its (empty) sum in the trash cell
    {
      int tt = t;
      if constexpr (!RESIDENT) asm volatile("" : "+v"(tt));      // (derived here: kept across the step loop, the stride was one more register stored to scratch per step)
      uint32_t oa, os;
      vdot_out(tt, oa, os);
      gsum_run<NT, MT::RING_LOW>(a.vdot, hdr.vdot_row0, hdr.vdot_rows, lane, oa, os);
    }
#pragma unroll
    for (int q = 0; q < SPT; q++) {
      const int s = q * NT + t;
      out[q] = s < NVAR ? XS[s] : 0.0;
    }
  };

  // ---- Jac_SP_x (gas.f:2656) on the V already in X: B products under their reaction, JVS sums into registers
  double jac0[JPT];
  auto jac = [&]() {
    uint64_t jfac[3 * RPT];
    {
      const uint64_t* jf = a.jac_fac;
      asm volatile("" : "+s"(jf));      // opaque: loads through it are not hoisted out of the step loop (and spilled there)
#pragma unroll
      for (int q = 0; q < 3 * RPT; q++) jfac[q] = G_(jf)[q * NT + t];
      if constexpr (RCT_PER_USE) load_rct(true);
    }
    lds_barrier();   // every lane is done reading AB as A
#pragma unroll
    for (int q = 0; q < RPT; q++) {      // the three products under one reaction: nine factor reads in flight, then the stores (see fun)
      double f0[3], f1[3], f2[3];
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const uint64_t w = jfac[q * 3 + b];
        f0[b] = X[w & 0xFFFFu];
        f1[b] = X[(w >> 16) & 0xFFFFu];
        f2[b] = X[(w >> 32) & 0xFFFFu];
      }
#pragma unroll
      for (int b = 0; b < 3; b++) {
        double p = rct[q] * f0[b];
        p = p * f1[b];
        p = p * f2[b];
        JBp[(uint32_t)(jfac[q * 3 + b] >> 48)] = p;         // unused product slots write a spare cell: no branch
      }
    }
    lds_barrier();
    lap(13);
    // sums land in this thread's own cells of the Ghimj area (free here: ros_PrepareMatrix rebuilds it from jac0)
    gsum_run<NT, MT::RING_LOW>(a.jvs, hdr.jvs_row0, hdr.jvs_rows, lane, 8u * (uint32_t)t, 8u * (uint32_t)NT);
    lap(14);
#pragma unroll
    for (int q = 0; q < JPT; q++) jac0[q] = M[q * NT + t];
  };

  // ---- the step's first Fun_x and its Jac_SP_x in one go (both read the same V): the A and the B products are formed in ONE phase — the
  //      B products have an array of their own inside the Ghimj area — and the two gather-sum programs run back to back.  Two
  //      barrier phases and one table-stream start-up fewer per step than fun() followed by jac(); operation order inside every
  //      product and sum unchanged.
  auto fun_jac = [&](const double (&v)[SPT], double (&out)[SPT]) {
    uint64_t jfac[3 * RPT];
    {
      const uint64_t* jf = a.jac_fac;
      asm volatile("" : "+s"(jf));      // opaque: loads through it are not hoisted out of the step loop (and spilled there)
#pragma unroll
      for (int q = 0; q < 3 * RPT; q++) jfac[q] = G_(jf)[q * NT + t];
    }
    if constexpr (!FFAC_RESIDENT) load_ffac();
    if constexpr (RCT_PER_USE) load_rct(true);
#pragma unroll
    for (int q = 0; q < SPT; q++) {
      const int s = q * NT + t;
      if (s < NVAR) X[s] = v[q];
    }
    lds_barrier();   // X is complete; nobody reads the Ghimj area any more (the last solve's sweeps are behind the error norm's barriers)
    {
      double f0[RPT], f1[RPT], f2[RPT];
      uint32_t slot[RPT];
#pragma unroll
      for (int q = 0; q < RPT; q++) {
        uint64_t w = ffac[q];
        asm volatile("" : "+v"(w));
        f0[q] = X[w & 0xFFFFu];
        f1[q] = X[(w >> 16) & 0xFFFFu];
        f2[q] = X[(w >> 32) & 0xFFFFu];
        slot[q] = (uint32_t)(w >> 48);
      }
#pragma unroll
      for (int q = 0; q < RPT; q++) {
        double p = rct[q] * f0[q];
        p = p * f1[q];
        p = p * f2[q];
        AB[slot[q]] = p;
      }
    }
#pragma unroll
    for (int q = 0; q < RPT; q++) {
      double f0[3], f1[3], f2[3];
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const uint64_t w = jfac[q * 3 + b];
        f0[b] = X[w & 0xFFFFu];
        f1[b] = X[(w >> 16) & 0xFFFFu];
        f2[b] = X[(w >> 32) & 0xFFFFu];
      }
#pragma unroll
      for (int b = 0; b < 3; b++) {
        double p = rct[q] * f0[b];
        p = p * f1[b];
        p = p * f2[b];
        JBp[(uint32_t)(jfac[q * 3 + b] >> 48)] = p;
      }
    }
    lds_barrier();
    lap(13);
    {
      int tt = t;
      if constexpr (!RESIDENT) asm volatile("" : "+v"(tt));
      uint32_t oa, os;
      vdot_out(tt, oa, os);
      gsum_run<NT, MT::RING_LOW>(a.vdot, hdr.vdot_row0, hdr.vdot_rows, lane, oa, os);
    }
    gsum_run<NT, MT::RING_LOW>(a.jvs, hdr.jvs_row0, hdr.jvs_rows, lane, 8u * (uint32_t)t, 8u * (uint32_t)NT);      // (other cells, another source array: no barrier between)
    lap(14);
#pragma unroll
    for (int q = 0; q < SPT; q++) {
      const int s = q * NT + t;
      out[q] = s < NVAR ? XS[s] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < JPT; q++) jac0[q] = M[q * NT + t];
  };

  // ---- ros_PrepareMatrix_x (gas.f:1404), first half: Ghimj = -Jac0, diagonal += 1/(H*gamma).
  //      Returns (workgroup-uniform) whether a diagonal is exactly zero, the condition KppDecomp_x tests (gas.f:6157).
  auto prepare = [&](double ghinv, const double (&rhs)[SPT]) -> bool {
    if constexpr (!RESIDENT) load_pos();
    if (t == 0) { flags[0] = 0; flags[1] = 0x7fffffff; }
    lds_barrier();   // also: all readers of M from the previous attempt are done
#pragma unroll
    for (int q = 0; q < SPT; q++) {      // stage-1 right-hand side: the LU program forward-sweeps it while it factorises
      const int s = q * NT + t;
      if (s < NVAR) XS[s] = rhs[q];
    }
    // branch-free: a slot the thread does not have stores to the trash cell, the diagonal term is a select
    bool zero_diag = false;
    auto put = [&](uint32_t p, double minus_jac) {
      const bool none = p == kPosNone, dg = !none && (p & kPosDiag);
      const double vd = minus_jac + ghinv;
      const double v = dg ? vd : minus_jac;
      zero_diag |= dg && (v == 0.0);
      M[none ? (uint32_t)(NNZ + NVAR + 2) : (p & 0x7FFFu)] = v;
    };
#pragma unroll
    for (int q = 0; q < JPT; q++) {
      uint32_t pw = jpos[q / 2];
      asm volatile("" : "+v"(pw));        // unpack here, not hoisted into registers that live across the step loop
      put((q & 1) ? (pw >> 16) : (pw & 0xFFFFu), -jac0[q]);
    }
#pragma unroll
    for (int q = 0; q < ZPT; q++) {      // fill-in slots: -0.0.  A diagonal among them (a species whose rate does not depend on itself) sits in
      uint32_t pw = zpos[q / 2];          // slot 0 of its thread (schedule.cpp sorts them there): the other slots have nothing to select
      asm volatile("" : "+v"(pw));
      const uint32_t p = (q & 1) ? (pw >> 16) : (pw & 0xFFFFu);
      if (q == 0) put(p, scalar_const(-0.0));      // (formed in scalar registers here: as a hoisted vector-register pair the zero was the 128-register kernel's one spilled value)
      else M[p == kPosNone ? (uint32_t)(NNZ + NVAR + 2) : (p & 0x7FFFu)] = -0.0;
    }
    if (zero_diag) flags[0] = 1;
    lds_barrier();
    return flags[0] != 0;
  };

  // ---- KppSolve_x (gas.f:6206) on a register vector.  swept = true: XS already holds the forward-swept vector (the LU
  //      program carried the stage-1 right-hand side through the elimination), only the backward half is left.
  // the tail chain (one wave).  SWEPT: the forward half is done already (stage 1: inside the LU program), up to the dense block's own
  // columns where the mechanism has one
  auto tail = [&](auto swept_tag) {
    constexpr bool SWEPT = decltype(swept_tag)::value;
    constexpr uint32_t xs_tail = 8u * (NNZ + NVAR - 64 * MT::TAIL_REGS), r_tail = 8u * (NNZ + NVAR + 4 + NVAR - 64 * MT::TAIL_REGS);
#ifndef MISTRA_LOW_BLOCK_TAIL      // 1: the kernels with the low ring placement whose tail is ONE register (gas) run the block form too
#define MISTRA_LOW_BLOCK_TAIL 1
#endif
    if constexpr (MT::RING_LOW && !(MISTRA_LOW_BLOCK_TAIL && MT::TAIL_REGS <= MISTRA_LOW_BLOCK_TAIL)) {
      static_assert(MT::DENSE_ND == 0, "the column form has no dense-block variant");
      tail_solve_columns<MT::TAIL_REGS, SWEPT ? MT::TAIL_REGS : 0, true>(a.tail, xs_tail, r_tail, lane);
    } else if constexpr (MT::RING_LOW) {
      tail_solve<MT::TAIL_REGS, SWEPT ? 4 * MT::TAIL_REGS : 0, true>(a.tail, xs_tail, r_tail, lane);
    } else {
      tail_solve<MT::TAIL_REGS, !SWEPT ? 0 : MT::DENSE_ND ? 4 * (MT::TAIL_REGS - 1) : 4 * MT::TAIL_REGS, false>(a.tail, xs_tail, r_tail, lane);
    }
  };
  auto solve = [&](double (&k)[SPT], bool swept) {
    for (int i = t; i < a.n_temps; i += NT) M[NNZ + 2 * NVAR + 4 + i] = 0.0;         // partial-sum cells of the head sweeps
    if (!swept) {
#pragma unroll
      for (int q = 0; q < SPT; q++) {
        const int s = q * NT + t;
        if (s < NVAR) XS[s] = k[q];
      }
      lds_barrier();
      lap(6);
      vm_run<NT, 4, kVmSweepUpdPerRec>(a.solve_head_fwd, hdr.fwd_row0, lane);                                          // head rows, all waves
      lap(8);
      if (wave == 0)                                                                         // tail chain, one wave
        tail(std::false_type{});
    } else {
      lap(6);
      if (wave == 0)      // with the dense tail block the forward chain still has the block's own columns to do
        tail(std::true_type{});
    }
    lds_barrier();
    lap(9);
    vm_run<NT, 4, kVmSweepUpdPerRec>(a.solve_head_bwd, hdr.bwd_row0, lane);
    lap(10);
#pragma unroll
    for (int q = 0; q < SPT; q++) {
      const int s = q * NT + t;
      if (s < NVAR) k[q] = XS[s];
    }
  };

  // ---- ros_ErrorNorm_x (gas.f:1341): wave shuffle reduction, then the NW partial sums in a fixed order
  auto error_norm = [&](const double (&y0)[SPT], const double (&y1)[SPT], const double (&ye)[SPT]) -> double {
    double part = 0.0;
#pragma unroll
    for (int q = 0; q < SPT; q++) {
      const int s = q * NT + t;
      if (s < NVAR) {
        const double ymax = fmax_f(fabs(y0[q]), fabs(y1[q]));
        const double scale = 1.0e-25 + 1.0e-3 * ymax;
        const double e = ye[q] / scale;
        part = part + e * e;
      }
    }
    part = wave_sum(part);
    lds_barrier();   // red[] may still be read from the previous step
    {
      int wv = wave;
      asm volatile("" : "+v"(wv));      // (the cell's address is formed here: kept across the step loop it was one more spilled register)
      if (lane == 0) red[wv] = part;
    }
    lds_barrier();
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) sum += red[w];
    return sqrt(sum / (double)NVAR);
  };

  // ---- RosenbrockIntegrator_x (gas.f:1180-1336); option values are the ones INTEGRATE_x/Rosenbrock_x fix
  const double Tstart = a.tin, Tend = a.tout;
  const double Roundoff = 2.220446049250313e-16, Hmin = 0.0, Hmax = wave_uniform(fabs(Tend - Tstart));
  const double FacMin = 0.2, FacMax = 6.0, FacRej = 0.1, FacSafe = 0.9;
  const double Direction = (Tend >= Tstart) ? 1.0 : -1.0;
  double T = Tstart;
  // Hstart: INTEGRATE_x fixes RPAR(3) = 1e-3 (gas.f:743).  Opt-in departure from the reference: a caller that keeps
  // each cell's last step size from one chemistry timestep to the next passes it in a.hstart; <= 0 means the reference's value.
  const double hstart0 = (a.hstart && G_(a.hstart)[cell] > 0.0) ? G_(a.hstart)[cell] : 1.0e-3;
  // The step size H lives across the whole step loop, and the compiler carried a vector-register copy of it (and of Hexit) around
  // the loop through scratch: 8-16 bytes per lane and step, stored through to HBM — most of aer's memory traffic in rounds 2 and 3.
  // So it lives in LDS: one cell per wave behind the error norm's partial sums, written by the wave's lane 0 and read back by the
  // same wave (LDS operations of a wave complete in order: no barrier is involved); every use below fetches it afresh.  Hexit, which
  // only thread 0 needs when the integrator returns, sits in the last cell of that block.
  auto Hget = [&]() -> double {
    int wv = wave;
    asm volatile("" : "+v"(wv));
    return wave_uniform(*(volatile lds_f64*)(uintptr_t)(8u * (uint32_t)(L::RED + L::RED_H + wv)));
  };
  auto Hset = [&](double v) {
    int wv = wave;
    asm volatile("" : "+v"(wv));
    if (lane == 0) *(volatile lds_f64*)(uintptr_t)(8u * (uint32_t)(L::RED + L::RED_H + wv)) = v;
  };
  static_assert(NW <= L::RED_H && L::RED_H + NW <= L::RED_HEXIT, "red[]: partial sums | H per wave | Hexit");
  if (t == 0) *(volatile lds_f64*)(uintptr_t)(8u * (uint32_t)(L::RED + L::RED_HEXIT)) = 0.0;
  {
    double H0 = fmin_f(fmin_f(hstart0, fabs(Tend - Tstart)), Hmax);
    if (fabs(H0) <= 10.0 * Roundoff) H0 = 1.0e-5;
    Hset(H0);
  }
  bool RejectLastH = false, RejectMoreH = false;
  int nfun = 0, njac = 0, nstp = 0, nacc = 0, nrej = 0, ndec = 0, nsol = 0, nsng = 0;
  int ierr = 1;

  double ynew[SPT], fcn0[SPT], fcn[SPT], k1[SPT], k2[SPT], k3[SPT], yerr[SPT];

  while (fabs(Tend - T) >= Roundoff) {
    if (nstp > a.max_steps) { ierr = -6; break; }      // Max_no_steps: 100000 (gas.f:1042, 1199)
    {
      const double H = Hget();
      if (((T + scalar_const(0.1) * H) == T) || (H <= Roundoff)) { ierr = -7; break; }
      if (t == 0) *(volatile lds_f64*)(uintptr_t)(8u * (uint32_t)(L::RED + L::RED_HEXIT)) = H;
      Hset(fmin_f(H, fabs(Tend - T)));
    }

    lap(6);
    if constexpr (!RESIDENT) {      // (Jac_SP below defines every entry.  Said here with an empty statement that "writes" them, so that no value — not even an
#pragma unroll                     //  undefined one — is carried around the step loop: the 128-register kernels carried one through scratch)
      for (int q = 0; q < JPT; q++) asm volatile("" : "=v"(jac0[q]));
    }
    if constexpr (L::MERGE_FUN_JAC) {
      fun_jac(y, fcn0);
      dump_vec(0, fcn0);
    } else {
      fun(y, fcn0);
      dump_vec(0, fcn0);
      lap(0);
      jac();
    }
    // ros_FunTimeDerivative_x (gas.f:1375): Fun_x does not depend on T and RCONST is frozen, so
    // dFdT = (1/Delta)*(Fun - Fcn0) is an exact +0.0; the evaluation is skipped, its count and its "+ HG*0.0" are kept.
    nfun += 2;
    njac += 1;
    lap(1);

    bool accepted = false;
    while (!accepted) {
      {
        // K1's right-hand side, Fcn0 + HG*dFdT with dFdT = +0.0 (see above); independent of H
#pragma unroll
        for (int q = 0; q < SPT; q++) k1[q] = fcn0[q] + 0.0;
        int nconsecutive = 0;
        bool singular = true;
        while (singular) {
          const double ghinv = wave_uniform(1.0 / (Direction * Hget() * kRosGamma1));
          singular = prepare(ghinv, k1);
          dump_matrix(NVAR);      // Ghimj = 1/(H*gamma) - Jac0
          ndec += 1;
          lap(2);
          if (singular) {
            // which row: KppDecomp_x returns the FIRST row whose diagonal is exactly zero as it reaches it (IER = k, gas.f:6157 — rows are
            // untouched until their own turn, so the test sees the prepared value); ros_PrepareMatrix_x prints it (gas.f:1456).  Rare
            // path: Ghimj is still as prepared (no LU ran), every species' thread looks at its diagonal slot.
            if (a.sing_rows && nsng < 8) {
#pragma unroll
              for (int q = 0; q < SPT; q++) {      // every species' thread looks at its diagonal slot(s)
                const int sp = q * NT + t;
                const uint32_t dp = sp < NVAR ? (uint32_t)G_(a.diag_pos)[sp] : (uint32_t)kPosNone;
                if (dp != kPosNone && M[dp] == 0.0) atomicMin(&flags[1], sp + 1);
              }
              lds_barrier();
              if (t == 0) GM_(a.sing_rows)[(size_t)cell * 8 + nsng] = flags[1];
            }
            lds_barrier();   // everyone has read flags[0] (and flags[1]) before the retry clears them
            nsng += 1;
            nconsecutive += 1;
            if (nconsecutive <= 5) Hset(Hget() * 0.5);
            else { ierr = -8; break; }
          } else {
            vm_run<NT, MT::VM_SLOTS>(a.lu, hdr.lu_row0, lane);
            if constexpr (MT::SCALE_PASS) {      // else the scaling is the LU program's last round
              lap(3);
              scale_run<NT, MT::RING_LOW>(a.lu_scale, wave, lane);      // L(k,j) *= R(j); tail block: U(i,c) *= R(i)
              lds_barrier();
              lap(11);
            }
            if constexpr (MT::DENSE_ND > 0) {
              static_assert(MT::SCALE_PASS, "the Schur steps read what the scaling pass leaves");
              lap(3);
              dense_lu<MT, NT>(lane);
              lds_barrier();
              lap(12);
            }
            lap(3);
            if constexpr (DUMP) {
              if (MT::DENSE_ND == 0) lds_barrier();      // (the dense tail's caller has just passed one)
              dump_matrix(NVAR + NNZ);      // the factors as the kernel keeps them; R(k) = 1/U(k,k) behind them
              if (dumping)
                for (int i = t; i < NVAR; i += NT) dump[NVAR + 2 * NNZ + i] = M[NNZ + NVAR + 4 + i];
            }
          }
        }
        if (ierr == -8) break;
      }
      const double dh = wave_uniform(Direction * Hget());
      // stage 1: its right-hand side went through the LU program above, only the backward half of the solve is left
      lap(6);
      solve(k1, true);
      dump_vec(2 * NVAR + 2 * NNZ, k1);
      lap(4);
      // stage 2: new function value at Y + A21*K1
#pragma unroll
      for (int q = 0; q < SPT; q++) ynew[q] = y[q] + kRosA1 * k1[q];
      lap(6);
      fun(ynew, fcn);
      lap(0);
      nfun += 1;
      {
        const double hc = wave_uniform(kRosC1 / dh);
#pragma unroll
        for (int q = 0; q < SPT; q++) k2[q] = (fcn[q] + hc * k1[q]) + dh * 0.0;      // + HG*dFdT with dFdT = +0.0: (dh*gamma2)*0.0, a zero of dh's sign (gamma2 > 0)
      }
      lap(6);
      solve(k2, false);
      dump_vec(3 * NVAR + 2 * NNZ, k2);
      lap(4);
      // stage 3 reuses the stage-2 function value
      {
        const double hc1 = wave_uniform(kRosC2 / dh), hc2 = wave_uniform(kRosC3 / dh);
#pragma unroll
        for (int q = 0; q < SPT; q++) k3[q] = ((fcn[q] + hc1 * k1[q]) + hc2 * k2[q]) + dh * 0.0;      // (dh*gamma3)*0.0 likewise (gamma3 > 0)
      }
      lap(6);
      solve(k3, false);
      dump_vec(4 * NVAR + 2 * NNZ, k3);
      lap(4);
      nsol += 3;
#pragma unroll
      for (int q = 0; q < SPT; q++) {
        ynew[q] = ((y[q] + kRosM1 * k1[q]) + kRosM2 * k2[q]) + kRosM3 * k3[q];
        yerr[q] = ((0.0 + kRosE1 * k1[q]) + kRosE2 * k2[q]) + kRosE3 * k3[q];
      }
      lap(6);
      const double Err = error_norm(y, ynew, yerr);
      if constexpr (DUMP) {
        if (dumping && t == 0) { dump[5 * NVAR + 2 * NNZ] = Err; dump[5 * NVAR + 2 * NNZ + 1] = Hget(); }
        dumping = false;      // the first attempt only
      }
      lap(5);
      double root;
      if constexpr (MT::WAVES_PER_SIMD > 2) root = err_root(Err);      // register-starved kernels: see err_root
      else root = pow(Err, 1.0 / kRosElo);
      const double Fac = fmin_f(FacMax, fmax_f(FacMin, FacSafe / root));
      const double H = Hget();
      double Hnew = H * Fac;
      nstp += 1;
      if ((Err <= 1.0) || (H <= Hmin)) {
        nacc += 1;
#pragma unroll
        for (int q = 0; q < SPT; q++) y[q] = ynew[q];
        T = wave_uniform(T + dh);
        Hnew = fmax_f(Hmin, fmin_f(Hnew, Hmax));
        if (RejectLastH) Hnew = fmin_f(Hnew, H);
        RejectLastH = false;
        RejectMoreH = false;
        Hset(Hnew);
        accepted = true;
      } else {
        if (RejectMoreH) Hnew = H * scalar_const(FacRej);
        RejectMoreH = RejectLastH;
        RejectLastH = true;
        Hset(Hnew);
        if (nacc >= 1) nrej += 1;
      }
    }
    if (ierr < 0) break;
  }

  // ---- results
  {
    int tt = t;
    asm volatile("" : "+v"(tt));      // (the output offset is formed here: shared with the input load's it was a 64-bit value kept across the whole step loop — in scratch, in the 128-register kernels)
#pragma unroll
    for (int q = 0; q < SPT; q++) {
      const int s = q * NT + tt;
      if (s < NVAR) GM_(a.var_out)[(size_t)cell * NVAR + s] = y[q];
    }
  }
  if (t == 0) {
    GM_(a.ierr)[cell] = ierr;
    gptr_mut<int32_t> st = GM_(a.stats) + (size_t)cell * 8;
    st[0] = nfun; st[1] = njac; st[2] = nstp; st[3] = nacc; st[4] = nrej; st[5] = ndec; st[6] = nsol; st[7] = nsng;
    if constexpr (profiling) {
      lap(6);
      pc[7] = clock64() - t_begin;
      for (int k = 0; k < kProfSlots; k++) GM_(a.prof)[(size_t)cell * kProfSlots + k] = pc[k];
    }
    if (a.texit_hexit) {
      GM_(a.texit_hexit)[(size_t)cell * 2] = T;
      GM_(a.texit_hexit)[(size_t)cell * 2 + 1] = *(volatile lds_f64*)(uintptr_t)(8u * (uint32_t)(L::RED + L::RED_HEXIT));
    }
    if (a.h_last) GM_(a.h_last)[cell] = Hget();
  }
}

// ---- launchers (one explicit instantiation per supported <mechanism, workgroup size>)
template <class MT, int NT>
hipError_t launch_ros3(const KernelArgs& a, hipStream_t stream, bool* lds_configured) {
  constexpr size_t lds_bytes = LdsLayout<MT, NT>::TOTAL * sizeof(double);
  bool& configured = *lds_configured;      // per device (capi.cpp: DeviceState): the attribute is a property of the device's code object
  auto kern = ros3_integrate_kernel<MT, NT, 0>;
  auto kern_prof = ros3_integrate_kernel<MT, NT, 1>;      // MISTRA_CHEM_PROFILE diagnostics (capi.cpp)
  auto kern_dump = ros3_integrate_kernel<MT, NT, 2>;      // first-step dump (mistra_chem_debug_first_step)
  if (!configured) {
    for (const void* k : {reinterpret_cast<const void*>(kern), reinterpret_cast<const void*>(kern_prof), reinterpret_cast<const void*>(kern_dump)}) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
    configured = true;
  }
  if (a.ncell <= 0) return hipSuccess;
  if (a.dump) hipLaunchKernelGGL(kern_dump, dim3((unsigned)a.ncell), dim3(NT), lds_bytes, stream, a);
  else if (a.prof) hipLaunchKernelGGL(kern_prof, dim3((unsigned)a.ncell), dim3(NT), lds_bytes, stream, a);
  else hipLaunchKernelGGL(kern, dim3((unsigned)a.ncell), dim3(NT), lds_bytes, stream, a);
  return hipGetLastError();
}

#ifdef MISTRA_DIAG_STAMPS
extern "C" int mistra_diag_dense_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dense_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_dense_stamps), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif

template hipError_t launch_ros3<GasTraits, kGasNT>(const KernelArgs&, hipStream_t, bool*);
template hipError_t launch_ros3<AerTraits, kAerNT>(const KernelArgs&, hipStream_t, bool*);
template hipError_t launch_ros3<TotTraits, kTotNT>(const KernelArgs&, hipStream_t, bool*);
// (a 1024-thread tot variant was measured slower, and instantiating it caps the register budget of the shared
//  non-inlined device functions at that of a 16-wave workgroup)

}  // namespace mistra
