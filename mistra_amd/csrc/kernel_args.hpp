// Plain-old-data argument blocks shared by the host API (capi.cpp) and the device code (ros3_kernel.hip).
#pragma once
#include <cstdint>

namespace mistra {

constexpr int kProfSlots = 16;   // phase counters of the profiling kernel variant
constexpr int kVmSweepUpdPerRec = 3;   // updates per record of the sweep programs (schedule.hpp: VM_SWEEP_UPD_PER_REC; ros3_kernel.hip: vm_run)

struct VmDev {                 // LDS VM program in device memory (see schedule.hpp)
  const uint32_t* wave_base;   // [NW]         first record row of each wave's stream
  const uint16_t* blk_n;       // [nrounds*NW] record rows per (round, wave)
  const uint32_t* recs;        // uint4 per lane and row, 16-byte aligned
  int nrounds;
};

struct GsDev {                 // gather-sum program in device memory (see schedule.hpp)
  const uint32_t* wave_base;   // [NW]     first row of each wave's stream
  const uint16_t* rows;        // [NW]     rows of each wave's stream, multiples of 4
  const uint32_t* recs;        // two uint4 per lane and row: 4 LDS byte addresses, 4 float coefficients
};

struct ScaleDev {              // final scaling of the factorisation (schedule.hpp: ScaleProgram)
  const uint32_t* recs;        // u32x4 per lane and slot: two (tgt, aux) LDS byte-address pairs
  int nslots;                  // per lane, multiple of 8
  int wave_slots;              // wave w starts at slot w*wave_slots (nslots + the look-ahead slack)
};

struct TailDev {               // tail chain of the triangular solves (schedule.hpp: TailSolve)
  const uint32_t* fwd;         // u32x4 per lane and group of 4 columns, columns ascending
  const uint32_t* bwd;         // same, columns descending
};

struct DenseDev {              // dense tail block (schedule.hpp: DenseTail); null where the mechanism has none
  const uint32_t* row_info;    // [192][4]: first M cell of a row's column range, absent-column mask lo / hi, 0 (the kernel keeps rows 0..63)
  const uint16_t* schur_cells; // [8 * DENSE_KB * 64]: operand cells of the Schur steps in MFMA lane order (schedule.hpp: DenseTail)
};

struct KernelArgs {
  // per-cell data, cell-major (one cell's VAR / FIX / RCONST contiguous, as COMMON /GDATA_x/ holds them)
  const double* var_in;        // [ncell][NVAR]
  const double* fix;           // [ncell][NFIX]
  const double* rconst;        // [ncell][NREACT]
  double* var_out;             // [ncell][NVAR]   may alias var_in
  int32_t* ierr;               // [ncell]         1 = success, <0 = ros_ErrorMsg code (gas.f:1474)
  int32_t* stats;              // [ncell][8]      Nfun,Njac,Nstp,Nacc,Nrej,Ndec,Nsol,Nsng  (COMMON /Statistics/)
  double* texit_hexit;         // [ncell][2] or null: what INTEGRATE_x leaves in TIN and STEPMIN
  const double* hstart;        // [ncell] or null: first step size per cell instead of INTEGRATE_x's 1e-3 (opt-in, not the reference's behaviour)
  double* h_last;              // [ncell] or null: the step size H when the integrator returned (ros_ErrorMsg_x prints it, gas.f:1506)
  int32_t* sing_rows;          // [ncell][8] or null: rows (1-based) of the zero pivots KppDecomp_x met in this call, in order of occurrence — the
                               //   IER it returns and ros_PrepareMatrix_x prints (gas.f:6157, 1456); entries past min(Nsng, 8) are not written
  double* dump;                // [ncell][5*NVAR + 2*LU_NONZERO + 2] or null: first-step dump (kernel VARIANT 2): Fcn0 | Ghimj prepared |
                               //   Ghimj factorised (kernel form) | R | K1 | K2 | K3 | Err, H  of the first attempt of the first step
  unsigned long long* prof;    // [ncell][kProfSlots] or null: shader-clock cycles per phase (diagnostics, see capi.cpp)
  double tin, tout;
  int32_t ncell;
  int32_t n_temps;             // partial-sum cells the solve programs use (zeroed per solve)
  int32_t max_steps;           // Max_no_steps of RosenbrockIntegrator_x (gas.f:1199): 100000, the default INTEGRATE_x leaves (capi.cpp: make_args)
  // mechanism schedule
  const double* consts;        // [NCONST]
  const uint64_t* fun_fac;
  const uint64_t* jac_fac;
  const uint16_t* jvs_pos;
  const uint16_t* zero_pos;
  const uint16_t* diag_pos;
  GsDev vdot, jvs;
  VmDev lu, solve_head_fwd, solve_head_bwd;
  ScaleDev lu_scale;
  TailDev tail;
  DenseDev dense;
};

}  // namespace mistra
