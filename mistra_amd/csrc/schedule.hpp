// Host-side schedule compiler: mechanism tables -> static per-thread programs for the one-workgroup-per-cell
// Rosenbrock kernel (ros3_kernel.hip).  Everything here is integer bookkeeping done once per mechanism at init.
//
// Two tiny "machines" run inside the kernel, both over LDS-resident data:
//
//  * the LDS VM ("vm"): memory M = [ Ghimj (nnz) | XS (nvar) | 0.0 ].  A program is a list of rounds separated
//    by workgroup barriers; inside a round every lane walks its own list of fixed 16-byte RECORDS:
//        w0 = tgt | dv<<14 | FIRST | LAST | DIV | ACTIVE        w1..w3 = i1 | i2<<14   (three update slots)
//        FIRST : acc = M[tgt]                 (else the lane's acc carries over from its previous record)
//        each update slot: acc = acc - M[i1]*M[i2]   (one multiply, one subtract, no contraction; an unused slot
//                                                     points both indices at the 0.0 cell: acc - 0*0 = acc exactly)
//        LAST  : M[tgt] = DIV ? acc / M[dv] : acc     (only if ACTIVE; idle padding records are not)
//    A wave's records form one linear stream over all rounds (the last row of a round carries an end-of-round mark, a
//    wave without work in a round gets one null row), so table loads run ahead of use whatever the round structure
//    and nothing but the records themselves is read from memory.
//    It expresses the sparse LU (KppDecomp_x, gas.f:6142: entry (k,c) receives  -L(k,j)*U(j,c)  for ascending j,
//    L entries are divided by the pivot) and both triangular sweeps of KppSolve_x (gas.f:6206) with the reference's
//    per-entry operation ORDER preserved where that is free: an entry's updates are cut into chunks, a chunk is
//    issued no earlier than the first round in which its operands are final, chunks of one entry stay in ascending-j
//    order.  That cuts the LU's dependency depth from ~9600 serial updates (tot) to ~165 rounds.  Chunks that are not
//    on the critical path are merged forward (issued later, together with the entry's next chunk) to save record
//    headers.  The backward sweep is the exception to order-keeping: the reference subtracts U(i,c)*X(c) for ASCENDING
//    c while the X(c) become known in DESCENDING c, which would serialise whole dot products; there the updates are
//    applied in readiness order (keep_order = false) — same terms, different summation order, round-off level.
//
//  * the gather-sum machine ("gsum"): out = c0*src[i0] + c1*src[i1] + ... left to right, coefficient as float
//    (every stoichiometric coefficient in the reference is a default-REAL literal or a small integer),
//    stored as groups of four terms (one 16-byte index load + one 16-byte coefficient load per lane and group).
//    It expresses the Vdot aggregation of Fun_x (gas.f:2395) and the JVS construction of Jac_SP_x (gas.f:3812).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mech_tables.hpp"

namespace mistra {

constexpr uint32_t VM_IDX_BITS = 14;
constexpr uint32_t VM_IDX_MASK = (1u << VM_IDX_BITS) - 1;
constexpr uint32_t VM_FIRST = 1u << 28;
constexpr uint32_t VM_LAST = 1u << 29;
constexpr uint32_t VM_DIV = 1u << 30;
constexpr uint32_t VM_ACTIVE = 1u << 31;
constexpr uint32_t VM_W1_EOR = 1u << 31;    // on w1 of every lane of a row: last row of this round for the wave -> barrier
constexpr uint32_t VM_W1_NULL = 1u << 30;   // on w1: the row carries no work (a wave with nothing to do in a round)
constexpr int VM_UPD_PER_REC = 3;
constexpr int VM_LOOKAHEAD_ROWS = 16;    // >= 2x the kernel's table look-ahead depth (ros3_kernel.hip: kVmDepth = 8, kGsDepth = 4)

constexpr uint32_t GS_FIRST = 1u << 16;
constexpr uint32_t GS_NOP = 1u << 17;

constexpr uint16_t POS_DIAG = 0x8000;   // flag on a Ghimj slot number: the slot is a diagonal
constexpr uint16_t POS_NONE = 0xFFFF;

struct VmEntry {
  int tgt = 0;                                  // M index updated
  int dv = -1;                                  // M index of the divisor applied after the last update, -1 = none
  int phase = 0;                                // phases run strictly one after the other
  bool keep_order = true;                       // false: updates may be applied in the order their operands get ready
  std::vector<std::pair<int, int>> upd;         // (i1, i2) pairs, applied in this order (if keep_order)
};

struct VmProgram {
  int nt = 0, nw = 0, nrounds = 0, zero_slot = 0;
  std::vector<uint32_t> wave_base;              // [nw]  first record row of each wave's linear stream
  std::vector<uint16_t> blk_n;                  // [nrounds*nw] record rows of (round, wave), null rows included (census / emulator)
  std::vector<uint32_t> recs;                   // [((wave_base[w] + row)*64 + lane)*4 + k]   one uint4 per lane and row
  // census
  int64_t n_updates = 0, n_items = 0, n_records = 0, wave_rows = 0, crit_rows = 0;
};

struct GsumProgram {
  int nt = 0, nw = 0, nq = 0;
  std::vector<uint32_t> blk_base;               // [nq*nw]  first group row of (q, wave)
  std::vector<uint16_t> blk_n;                  // [nq*nw]  group rows (4 terms per lane and row)
  std::vector<uint32_t> idx;                    // [((blk_base + row)*64 + lane)*4 + k]  src index | GS_FIRST | GS_NOP
  std::vector<float> coef;                      // same indexing
  int64_t n_terms = 0, wave_rows = 0;
};

// Triangular solves, tail part.  The last m rows of the LU pattern (the gas-phase block every other species couples
// to) form a nearly dense triangle whose substitution is an inherently serial chain; it is run by ONE wave with the
// tail of the solution vector in registers (lane l holds rows h+l, h+64+l), the pivot value passed lane-to-lane by
// v_readlane, and only the matrix entries gathered from LDS through these per-column index tables:
//   word (16 bit) for column q, lane l, register r  =  Ghimj slot of entry (row h+64r+l, column h+q), or the 0.0 cell
struct TailSolve {
  int m = 0, h = 0, regs = 0;                   // tail rows [h, h+m), m = 64*regs, regs in {1,2}
  std::vector<uint32_t> fwd;                    // [((q/4)*64 + lane)*4 + q%4]  lo16: r=0, hi16: r=1; columns ascending
  std::vector<uint32_t> bwd;                    // same, columns DEscending: group g, word c  <->  q = m-1-(4g+c)
  std::vector<uint16_t> diag;                   // [regs*64] Ghimj slot of the diagonal of tail row r*64+lane
};

struct KernelSchedule {
  int nt = 0, nw = 0;
  int spt = 0;   // species per thread          s = q*nt + t
  int rpt = 0;   // reactions per thread        r = q*nt + t
  int jpt = 0;   // structurally non-zero Jacobian entries per thread
  int zpt = 0;   // structurally zero (fill-in) entries per thread
  int n_jnz = 0, n_jzero = 0;
  // Fun_x products: A(r) = RCT(r)*X[f1]*X[f2]*X[f3], padded with the constant 1.0
  std::vector<uint64_t> fun_fac;                // [rpt*nt]  f1 | f2<<16 | f3<<32 | valid<<48
  GsumProgram vdot;                             // src = A (LDS), output (q,t) = species q*nt+t
  // Jac_SP_x products, grouped under the reaction that owns the rate constant: up to 3 B's per reaction
  std::vector<uint64_t> jac_fac;                // [(q*3 + b)*nt + t]  f1 | f2<<16 | f3<<32 | out<<48 (0xFFFF = none)
  GsumProgram jvs;                              // src = B (LDS), output (q,t) = jac0 register q of thread t
  std::vector<uint16_t> jvs_pos;                // [jpt*nt] Ghimj slot (| POS_DIAG) of that output, POS_NONE = idle
  std::vector<uint16_t> zero_pos;               // [zpt*nt] Ghimj slots that Jac_SP_x sets to 0 (| POS_DIAG)
  std::vector<uint16_t> diag_pos;               // [spt*nt] Ghimj slot of (s,s), POS_NONE past nvar
  VmProgram lu, solve;                          // solve = the whole of KppSolve_x as one VM program (kept for tests)
  VmProgram solve_head_fwd, solve_head_bwd;     // head rows (+ head-column part of tail rows) around the tail chain
  TailSolve tail;
};

VmProgram build_vm_program(std::vector<VmEntry> entries, int msize, int zero_slot, int nt, int merge_budget = 6);
std::vector<VmEntry> lu_entries(const MechTables& m);
std::vector<VmEntry> solve_entries(const MechTables& m);
std::vector<VmEntry> solve_head_fwd_entries(const MechTables& m, int h);
std::vector<VmEntry> solve_head_bwd_entries(const MechTables& m, int h);
TailSolve build_tail_solve(const MechTables& m, int zero_slot);
GsumProgram build_gsum_program(const std::vector<std::vector<std::pair<int, double>>>& outputs,
                               const std::vector<int>& slot_of_output, int nq, int nt);
KernelSchedule build_kernel_schedule(const MechTables& m, int nt);
std::string describe(const KernelSchedule& s);

}  // namespace mistra
