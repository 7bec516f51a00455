// Compile-time mechanism sizes and the LDS carve-up shared by the kernel and its launcher.
// Sizes: gas_Parameters.h:28-49 | aer_Parameters.h:28-49 | tot_Parameters.h:28-49 (NVAR NFIX NREACT LU_NONZERO);
// NB / NJNZ are counted from Jac_SP_x (number of B products / of JVS slots that are not `= 0`), NCONST from the
// factor literals (1.0 padding + the 2 of squared reactants); TAIL_REGS*64 = rows of the solve's tail chain
// (schedule.cpp: build_tail_solve); MAX_TEMPS = partial-sum cells reserved for the head sweeps of the solves
// (schedule.cpp: split_long_entries); DENSE_ND / DENSE_KB = rows of the dense tail block held in MFMA accumulator tiles and
// its rank-4 Schur steps (schedule.hpp: DenseTail, dense_config).  The host checks the loaded table against these.
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_args.hpp"

namespace mistra {

// Workgroup sizes and register budgets (waves per SIMD the compiler must leave room for) of the three instantiations; the
// macros exist for same-box A/B builds (tools/build_variant.sh), the product library is built without them.
#ifndef MISTRA_GAS_NT
#define MISTRA_GAS_NT 64             // ONE WAVEFRONT PER CELL (round 4): no workgroup barrier in the whole step loop, two species per lane; same box, 1e5 cells:
#endif                               // 128 threads / 8 cells per CU 2.96 M, 64 threads 9 cells per CU 3.26 M (128 registers, 64 B scratch) / 3.45 M (168 registers, none)
#ifndef MISTRA_AER_NT
#define MISTRA_AER_NT 256            // round 4: FOUR waves per cell with 256 registers (two species per lane, everything resident, the block-form tail chain), still two
#endif                               // cells per CU: 54.0 k -> 60.4 k timesteps/s on the same box (eight waves per cell were held to 128 registers)
#ifndef MISTRA_TOT_NT
#define MISTRA_TOT_NT 512
#endif
#ifndef MISTRA_GAS_WPS
#define MISTRA_GAS_WPS 3             // 168 registers: LDS (16.0 KB per cell) admits 10 one-wave cells per CU, i.e. at most three waves per SIMD anyway
#endif
#ifndef MISTRA_AER_WPS
#define MISTRA_AER_WPS 2
#endif
#ifndef MISTRA_AER_RING_LOW          // 1: look-ahead ring in v64.. (kernels held to 128 registers), 0: in v192.. (256 registers: the block-form tail chain)
#define MISTRA_AER_RING_LOW 0
#endif
#ifndef MISTRA_AER_SCALE_PASS        // whether the schedule compiler gives the factorisation's scaling its own pass at this workgroup size (schedule.cpp: >= 16 cells per thread)
#define MISTRA_AER_SCALE_PASS 1
#endif
#ifndef MISTRA_TOT_DENSE             // 0: A/B builds without the dense tail block (the whole LU as an LDS-VM program; schedule.hpp: dense_config follows)
#define MISTRA_TOT_DENSE 1
#endif
#ifndef MISTRA_TOT_WPS
#define MISTRA_TOT_WPS 2
#endif
#ifndef MISTRA_TOT_VM_SLOTS          // ring depth of the LDS VM executor (table rows in flight per lane): 4, 6 or 8 slots of 8 registers from v48 up
#define MISTRA_TOT_VM_SLOTS 4        // (the kernels held to 128 registers have room for 4)
#endif
#ifndef MISTRA_RESIDENT_MAX_WPS      // kernels with more waves per SIMD than this fetch their static per-thread words where they are used (ros3_kernel.hip)
#define MISTRA_RESIDENT_MAX_WPS 2
#endif
constexpr int kGasNT = MISTRA_GAS_NT, kAerNT = MISTRA_AER_NT, kTotNT = MISTRA_TOT_NT;

struct GasTraits { static constexpr int NVAR = 102, NFIX = 3, NREACT = 331, NNZ = 1110, NB = 568, NCONST = 2, NJNZ = 945, TAIL_REGS = 1, MAX_TEMPS = 10, WAVES_PER_SIMD = MISTRA_GAS_WPS, DENSE_ND = 0, DENSE_KB = 0; static constexpr bool RING_LOW = true, SCALE_PASS = false, RCT_LDS = false; static constexpr int VM_SLOTS = 4; };
struct AerTraits { static constexpr int NVAR = 257, NFIX = 5, NREACT = 979, NNZ = 6579, NB = 1598, NCONST = 2, NJNZ = 2831, TAIL_REGS = 2, MAX_TEMPS = 192, WAVES_PER_SIMD = MISTRA_AER_WPS, DENSE_ND = 0, DENSE_KB = 0; static constexpr bool RING_LOW = MISTRA_AER_RING_LOW, SCALE_PASS = MISTRA_AER_SCALE_PASS, RCT_LDS = true; static constexpr int VM_SLOTS = 4; };
struct TotTraits { static constexpr int NVAR = 417, NFIX = 7, NREACT = 1627, NNZ = 13503, NB = 2628, NCONST = 2, NJNZ = 4709, TAIL_REGS = 2, MAX_TEMPS = 768, WAVES_PER_SIMD = MISTRA_TOT_WPS, DENSE_ND = MISTRA_TOT_DENSE ? 64 : 0, DENSE_KB = MISTRA_TOT_DENSE ? 14 : 0; static constexpr bool RING_LOW = false, SCALE_PASS = true, RCT_LDS = false; static constexpr int VM_SLOTS = MISTRA_TOT_VM_SLOTS; };

constexpr int round_up2(int x) { return (x + 1) & ~1; }
// spare cells behind a product array (slots no reaction owns write there, one cell per lane of a wave — two lanes per cell in the one-wave
// workgroups, whose LDS budget decides how many cells share a CU): schedule.cpp builds the tables with the same count
constexpr int spare_cells(int nt) { return nt >= 128 ? 64 : 32; }
constexpr int max_i(int a, int b) { return a > b ? a : b; }

// offsets in doubles into the dynamic LDS block
template <class MT, int NT>
struct LdsLayout {
  static constexpr int M = 0;                                         // Ghimj | XS | 0.0 | 1.0 | trash | -1.0 | R | temps
  static constexpr int X = M + round_up2(MT::NNZ + 2 * MT::NVAR + 4 + MT::MAX_TEMPS);  // V | F | consts
  static constexpr int AB = X + round_up2(MT::NVAR + MT::NFIX + MT::NCONST);           // A or B products
  static constexpr int AB_TRASH = max_i(MT::NREACT, MT::NB);                           // spare cells, one per lane: products no reaction owns land here
  // Jac_SP's B products may live INSIDE the Ghimj area instead (dead between a step's start and ros_PrepareMatrix), behind the cells
  // the JVS sums land in: the step's first Fun and its Jac_SP then form their products in ONE phase and sum them back to back
  // (ros3_kernel.hip: fun_jac).  Where the area is too small for that (gas) B shares the array of A as before.
  static constexpr int JVS_CELLS = (MT::NJNZ + NT - 1) / NT * NT;
  static constexpr int SPARE = spare_cells(NT);
  static constexpr bool MERGE_FUN_JAC = JVS_CELLS + round_up2(AB_TRASH + SPARE) <= MT::NNZ;
  static constexpr int JB = MERGE_FUN_JAC ? M + JVS_CELLS : AB;                             // base of the B products
  // the A array: where B has an array of its own it holds the NREACT rate products and its spare cells (schedule.cpp: a_trash) — tot 8 KB,
  // aer 4.9 KB smaller than the shared array was (round 4)
  static constexpr int A_CELLS = MERGE_FUN_JAC ? MT::NREACT : AB_TRASH;
  // ... and aer (two cells per CU) keeps the cell's RATE CONSTANTS in the room that frees (traits: RCT_LDS) instead of fetching them from
  // global memory in front of every Fun / Jac_SP (~300 times per cell, round 3) or giving them eight registers
  static constexpr bool RCT_IN_LDS = MERGE_FUN_JAC && MT::RCT_LDS;
  static constexpr int RCT = AB + round_up2(A_CELLS + SPARE);
  static constexpr int RED = RCT + (RCT_IN_LDS ? round_up2(MT::NREACT) : 0);                            // (one shared cell: every such store of a wave hit the same address, and LDS serialises those)                             // per-wave partial sums
  // RED: per-wave partial sums of the error norm [RED_H) | the step size H, one cell per wave [RED_H, RED_HEXIT) | Hexit
  static constexpr int RED_H = NT == 64 ? 1 : 16, RED_HEXIT = NT == 64 ? 2 : 31, RED_CELLS = NT == 64 ? 4 : 32;
  static constexpr int FLAGS = RED + RED_CELLS;
  static constexpr int DINFO = FLAGS + 2;                                               // dense tail block: row table of the block's 64 rows, 16 bytes each (schedule.hpp: DenseTail)
  static constexpr int SCHUR = DINFO + (MT::DENSE_ND > 0 ? 64 * 2 : 0);                 // ... and the Schur steps' operand cells, uint16 x 8 x DENSE_KB x 64
  static constexpr int SCHUR_WORDS = MT::DENSE_ND > 0 ? 8 * MT::DENSE_KB * 64 / 2 : 0;   // in 32-bit words
  static constexpr int TOTAL = SCHUR + SCHUR_WORDS / 2;
  // dense_lu's panel buffers live in the A/B product array, which nothing reads between Jac_SP and the next Fun:
  // two buffers of [64][4] panel columns + [64][4] panel rows, two 64-entry broadcast rows of the eliminating wave
  static constexpr int PANEL = AB;
  static constexpr int PANEL_CELLS = 2 * 2 * 64 * 4 + 2 * 64 + 4;      // (+ the panel's four pivot reciprocals)
  static_assert(MT::DENSE_ND == 0 || PANEL_CELLS <= A_CELLS + SPARE, "panel buffers must fit the product array");
  static_assert(TOTAL * 8 <= 160 * 1024, "cell state does not fit the 160 KiB LDS of a gfx950 CU");
  static_assert(NT % 64 == 0 && NT <= 1024 && NT / 64 <= 32, "workgroup size");
};

template <class MT, int NT>
hipError_t launch_ros3(const KernelArgs& a, hipStream_t stream, bool* lds_configured);

}  // namespace mistra
