// Host-side schedule compiler: mechanism tables -> static per-thread programs for the one-workgroup-per-cell
// Rosenbrock kernel (ros3_kernel.hip).  Everything here is integer bookkeeping done once per mechanism at init.
//
// Two tiny "machines" run inside the kernel, both over LDS-resident data:
//
//  * the LDS VM ("vm"): memory M = [ Ghimj (nnz) | XS (nvar) | 0.0 | 1.0 | trash | -1.0 | R (nvar) | temps ],
//    R(k) = 1/U(k,k).
//    A program is a list of rounds separated by workgroup barriers; inside a round every lane walks its own list of
//    fixed 32-byte RECORDS whose fields are LDS byte offsets (a 16-byte packing of 14-bit indices was tried: the extra
//    decode arithmetic cost more than the halved table stream saved).  Records are self-contained (no state carried from one
//    to the next, no per-lane flags to decode):
//        d0 = tgt     d1 = aux | marks     d2..d4 = a1,r1,u1     d5..d7 = a2,r2,u2
//        acc = M[tgt];         (a continuation record finds what the lane's previous record stored)
//        acc -= (M[a1]*M[r1])*M[u1];   acc -= (M[a2]*M[r2])*M[u2];      (three roundings per update, no
//                                                                                        contraction)
//        M[tgt] = acc * M[aux]            aux = the 1.0 cell unless the entry is scaled by a pivot reciprocal
//        RCP:  M[tgt] = acc;  M[aux] = 1/acc           (a pivot publishes its reciprocal; rows with such lanes are marked)
//    An unused update slot points all three operands at the 0.0 cell (acc - (0*0)*0 = acc exactly); an idle lane
//    targets the trash cell.  An entry with more than two updates in a round takes consecutive records of one lane
//    (store, reload: LDS is in-order within a wave).  A wave's records form one linear stream over all rounds (the
//    last row of a round carries an end-of-round mark, a wave without work gets one null row), so table loads run
//    ahead of use whatever the round structure and nothing but the records themselves is read from memory.
//
//    Sparse LU (KppDecomp_x, gas.f:6142): entry (k,c) receives  -L(k,j)*U(j,c)  for ascending j.  The reference forms
//    the multiplier L(k,j) = W(j)/U(j,j) first; here an update reads the UNSCALED W(j), the pivot's reciprocal R(j) and
//    U(j,c) and forms (W(j)*R(j))*U(j,c): one round per pivot instead of two (no separate multiplier round on the
//    dependency chain), at the price of a multiplier that can differ from the quotient in the last bit.  Pivots store
//    their reciprocal when they become final (RCP, one IEEE division per pivot); the L entries are scaled in place by
//    one last round.  Per-entry update ORDER is the reference's (ascending j): an entry's updates are cut into
//    chunks, a chunk is issued no earlier than the first round in which its operands are final, chunks of one entry
//    stay in order; chunks off the critical path are merged forward to save record headers.  That cuts the LU's
//    dependency depth from ~9600 serial updates (tot) to ~100 rounds.
//    The LU program also carries the vector in XS through the elimination (X(i) -= (W(i,j)*R(j))*X(j), the forward
//    sweep of KppSolve_x for the right-hand side known before the factorisation, i.e. stage 1 of the Rosenbrock step):
//    those records ride in the rounds the factorisation needs anyway.
//    Triangular sweeps of KppSolve_x (gas.f:6206): updates are (L(i,j), 1.0, X(j)); the backward sweep multiplies by
//    R(i) instead of dividing and applies its terms in readiness order (keep_order = false: the reference subtracts
//    U(i,c)*X(c) for ASCENDING c while the X(c) become known in DESCENDING c).  Long dot products of the head sweeps
//    (a tail row has up to 42 head-column terms) are cut into partial sums over the temp cells and combined
//    (split_long_entries) so that no lane walks them serially.  Same terms, round-off level changes;
//    the CPU test-suite measures what such re-associations do to the reference algorithm itself.
//
//  * the gather-sum machine ("gsum"): out = c0*src[i0] + c1*src[i1] + ... left to right, coefficient as float
//    (every stoichiometric coefficient in the reference is a default-REAL literal or a small integer).
//    A wave's table is one linear stream of 32-byte records (four absolute LDS byte addresses + four floats per lane
//    and row), fetched through the same look-ahead ring as the tail chain.  Every lane of a wave produces its q-th output
//    in the same rows; the last of them carries a flush mark: the accumulator goes to the lane's q-th output cell in LDS
//    and restarts at -0.0 (so neither the first term nor the padding terms, (-0.0f * M[0.0 cell]), need a flag).  An
//    output with few terms costs one row, not a turn of the ring: the Jacobian's 4 709 outputs have two terms on
//    average.  It expresses the Vdot aggregation of Fun_x (gas.f:2395) and the JVS construction of Jac_SP_x (gas.f:3812)
//    in the reference's term order.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "kernel_args.hpp"
#include "mech_tables.hpp"

namespace mistra {

// record marks: every mark rides on d1 (the aux operand, which only marked rows read), so that d0 and d2..d7 are LDS
// byte addresses the executor hands to ds_read/ds_write as they stand
constexpr uint32_t VM_D1_RCP = 1u;           // lane: publish 1/result to aux instead of scaling by M[aux]
constexpr uint32_t VM_D1_CONT = 2u;          // lane: continuation record of the lane's previous record (same target).  The
                                             // shipped executor reloads the target (LDS is in-order within a wave); an
                                             // executor that prefetched operands would have to carry the accumulator
constexpr uint32_t VM_ROW_EOR = 1u << 24;    // every lane of a row: last row of this round for the wave -> barrier
constexpr uint32_t VM_ROW_NULL = 1u << 25;   // every lane of a row: the row carries no work (a wave with nothing to do in a round)
constexpr uint32_t VM_ROW_AUX = 1u << 26;    // every lane of a row: some lane publishes a reciprocal OR scales by M[aux] != 1.0
                                             // cell; rows without the mark skip the aux read and the final multiply
constexpr uint32_t VM_ROW_LOCAL = 1u << 27;  // with VM_ROW_EOR, wave 0 only: the NEXT round is this wave's own too (a small round the other waves have no
                                             // rows in): no barrier between them — the LDS operations of a wave complete in order
constexpr int VM_ROW_EOR_BIT = 24, VM_ROW_NULL_BIT = 25, VM_ROW_AUX_BIT = 26, VM_ROW_LOCAL_BIT = 27;
constexpr uint32_t VM_AUX_MASK = 0x00FFFFF8u;   // byte address part of d1
constexpr int VM_REC_WORDS = 8;
inline size_t vm_rec_index(size_t row, int lane, int k) { return row * 512 + (k < 4 ? 0 : 256) + (size_t)lane * 4 + (size_t)(k & 3); }
constexpr int VM_UPD_PER_REC = 2;          // updates (a, r, u) per record of the LU program
constexpr int VM_SWEEP_UPD_PER_REC = kVmSweepUpdPerRec;    // updates (a, u) per record of the triangular sweeps, whose middle operand is always the 1.0 cell
constexpr int VM_LOOKAHEAD_ROWS = 16;    // >= 2x the kernel's table look-ahead depth (ros3_kernel.hip)

// VM memory map for a mechanism with nnz LU slots and nvar species
struct VmLayout {
  int nnz = 0, nvar = 0, max_temps = 0;         // max_temps: partial-sum cells reserved behind R (ros3_kernel.hpp: MAX_TEMPS)
  int xs(int i = 0) const { return nnz + i; }
  int zero() const { return nnz + nvar; }
  int one() const { return nnz + nvar + 1; }
  int trash() const { return nnz + nvar + 2; }
  int minus_one() const { return nnz + nvar + 3; }
  int rdiag(int k = 0) const { return nnz + nvar + 4 + k; }
  int temp(int t = 0) const { return nnz + 2 * nvar + 4 + t; }      // partial-sum cells, zeroed before each solve
  int size() const { return nnz + 2 * nvar + 4 + max_temps; }
};

constexpr int GS_ROW_ALIGN = 4;         // a wave's row count is padded to a multiple of this (= ring depth in rows)
constexpr uint32_t GS_ROW_FLUSH = 1u;   // on the first address of every lane of a row: the lane's output is complete after this row

constexpr uint16_t POS_DIAG = 0x8000;   // flag on a Ghimj slot number: the slot is a diagonal
constexpr uint16_t POS_NONE = 0xFFFF;

struct VmUpd { int a, r, u; };                  // acc -= (M[a]*M[r])*M[u]

struct VmEntry {
  int tgt = 0;                                  // M index updated
  int mulr = -1;                                // M index of a factor applied after the last update, -1 = none
  int rcp = -1;                                 // M index that receives 1/result when the entry is final, -1 = none
  int phase = 0;                                // phases run strictly one after the other
  bool keep_order = true;                       // false: updates may be applied in the order their operands get ready
  std::vector<VmUpd> upd;                       // applied in this order (if keep_order)
};

struct VmProgram {
  int nt = 0, nw = 0, nrounds = 0, zero_slot = 0;
  int nbarriers = 0;                            // rounds that end in a workgroup barrier: what the executor counts (nrounds minus the local ones)
  int upd_per_rec = VM_UPD_PER_REC;             // 2: d2..d7 = (a1,r1,u1),(a2,r2,u2);  3: d2..d7 = (a1,u1),(a2,u2),(a3,u3), acc -= M[a]*M[u]
  std::vector<uint32_t> wave_base;              // [nw]  first record row of each wave's linear stream
  std::vector<uint16_t> blk_n;                  // [nrounds*nw] record rows of (round, wave), null rows included (census / emulator); 0 = the wave skips
                                                // the round (a local round of wave 0)
  std::vector<uint32_t> recs;                   // per row 512 words, PLANAR: words 0-3 of all 64 lanes, then words 4-7 of all
                                                // lanes (vm_rec_index): each of the executor's two 16-byte loads per record then
                                                // covers 16 whole cache lines instead of half of 32
  // census
  int64_t n_updates = 0, n_items = 0, n_records = 0, wave_rows = 0, crit_rows = 0;
  int64_t lds_cycles = 0;                       // modelled LDS-array cycles of all operand gathers and stores (bank conflicts included)
};

struct GsumProgram {
  int nt = 0, nw = 0, nq = 0;
  std::vector<uint32_t> wave_base;              // [nw]     first row of each wave's linear stream
  std::vector<uint16_t> rows;                   // [nw]     rows of the wave's stream (multiple of GS_ROW_ALIGN); exactly nq of them carry GS_ROW_FLUSH
  std::vector<uint32_t> recs;                   // [((wave_base[w] + row)*64 + lane)*8 + k]  k<4: LDS byte address (| mark on k=0), k>=4: float bits
  int64_t n_terms = 0, wave_rows = 0;
};

// Triangular solves, tail part.  The last m rows of the LU pattern (the gas-phase block every other species couples
// to) form a nearly dense triangle whose substitution is an inherently serial chain; it is run by ONE wave with the
// tail of the solution vector in registers (lane l holds rows h+l, h+64+l), the pivot value passed lane-to-lane by
// v_readlane, and only the matrix entries gathered from LDS through these per-column index tables:
//   word (16 bit) for column q, lane l, register r  =  Ghimj slot of entry (row h+64r+l, column h+q), or the 0.0 cell
// Forward: x(i) -= L(i,q)*x(q), columns ascending, as the reference's sweep orders the terms of a row.  Backward: the
// LU program leaves the tail block's upper triangle ROW-SCALED, U'(i,c) = U(i,c)*R(i), so that with x := R .* x the
// chain is  x(i) -= U'(i,q)*x(q)  (columns descending) with no quotient between consecutive columns; the reference
// computes (x(i) - sum U(i,c) x(c)) / U(i,i) — same terms, each carrying one more rounding.
struct TailSolve {
  int m = 0, h = 0, regs = 0;                   // tail rows [h, h+m), m = 64*regs, regs in {1,2}
  std::vector<uint32_t> fwd;                    // [((q/4)*64 + lane)*4 + q%4]  lo16: r=0, hi16: r=1; columns ascending
  std::vector<uint32_t> bwd;                    // same, columns DEscending: group g, word c  <->  q = m-1-(4g+c)
};

// The factorisation's last act, M[tgt] *= M[aux] for ~11 000 independent cells of the tot mechanism (L(k,j) *= R(j), and
// U(i,c) *= R(i) inside the tail block): far cheaper as its own tight pass than as VM rows, which would drag six unused
// operand gathers and the mark decoding along for every pair.
struct ScaleProgram {
  int nt = 0, nw = 0, nslots = 0;               // nslots: 16-byte slots per lane (multiple of 8), two (tgt, aux) byte-address pairs each
  std::vector<uint32_t> recs;                   // [((w*nslots + slot)*64 + lane)*4 + k]  (+ VM_LOOKAHEAD_ROWS slots of slack per wave)
  int64_t n_pairs = 0;
};

// Dense tail block.  The last nd rows and columns of the LU pattern (the gas-phase species every aqueous bin couples to)
// are a completely filled square once the fill-in is there (tot: 4089 of 4096 slots): ~40 % of KppDecomp_x's updates land
// in it from the pivots just before it, another ~40 % are its own factorisation.  The kernel holds that block D in
// REGISTERS, as 16x16 accumulator tiles of v_mfma_f64_16x16x4_f64 (lane l, register r of a tile: row (l>>4)+4r, column
// l&15), two tiles per wave on eight waves (wave w: block row w>>1, block columns 2(w&1), 2(w&1)+1), and works on it in
// rank-4 steps (ros3_kernel.hip: dense_lu):
//   * pivots [jm, h), four at a time ("Schur steps"): D -= W(:,j..j+3) * U'(j..j+3,:), operands gathered from their Ghimj
//     slots by the cell tables below: W the still unscaled L slots (the step also leaves L = W*R(j) in them), U' the
//     row-scaled U slots the scaling pass has left in the rows of the solves' tail chain (TailSolve).  Pivots below jm
//     reach D through the LU program as before (they touch few slots).
//   * D's own factorisation in panels of four pivots: the tile owners publish the panel's four columns and rows in LDS,
//     ONE wave eliminates inside the panel (lane = row for the L part, lane = column for the U part), everybody applies
//     the rank-4 update with one MFMA per tile.
// Every slot of D still receives its updates in ascending pivot order, as KppDecomp_x applies them (gas.f:6160-6171);
// inside an MFMA the four products are accumulated with fused multiply-adds.  Slots that are not in the sparsity pattern
// read as zero and stay zero.  The finished factors go back to their Ghimj slots in the form the solves expect:
// L(i,j) multipliers, U'(i,c) = U(i,c)*R(i) row-scaled, R(i) = 1/U(i,i) published (schedule.hpp: TailSolve).
struct DenseTail {
  int nd = 0, h = 0, jm = 0, kb = 0;            // D = rows/columns [h, h+nd), nd = 64; kb = (h - jm)/4 Schur steps
  // Where the block's entries and the Schur steps' operands live in Ghimj.  A CSR row keeps its columns ascending, so
  // the slots of one row inside a column range are contiguous: the slot of (row, range column c) is
  //     first + c - popcount(absent & ((1 << c) - 1)),   absent bit c set = not in the pattern (reads as the 0.0 cell).
  // row_info[g][4] = {first, absent lo, absent hi, 0}; the kernel keeps the table in LDS (2.9 KB) for the whole call:
  //   g = i         (i < 64)       row h+i, columns [h, n)      the block itself
  //   g = 64 + i    (i < 64)       row h+i, columns [jm, h)     L operands of the Schur steps
  //   g = 128 + r   (r < 4 kb)     row jm+r, columns [h, n)     U operands of the Schur steps
  std::vector<uint32_t> row_info;
  // The Schur steps' operand cells, resolved on the host (the row-table arithmetic above cost the kernel ~3 000 cycles of address
  // work per decomposition): uint16 M cells, absent = the 0.0 cell, in the order the MFMA lanes take them (lane l of a wave:
  // lrow = l>>4, lcol = l&15; wave w: block row I = w>>1, block columns 2(w&1), 2(w&1)+1):
  //   W part  [I (4)][k (kb)][lane]          cell of (row h + 16I + lcol, column jm + 4k + lrow)          A operand of step k
  //   U part  [w&1 (2)][k (kb)][t (2)][lane]  cell of (row jm + 4k + lrow, column h + 16(2(w&1) + t) + lcol)   B operand, tile t
  // The kernel keeps the table in LDS for the whole call (14 336 bytes for kb = 14).
  std::vector<uint16_t> schur_cells;
  size_t schur_w(int I, int k, int lane) const { return ((size_t)I * kb + k) * 64 + lane; }
  size_t schur_u(int half, int k, int t, int lane) const { return (size_t)4 * kb * 64 + (((size_t)half * kb + k) * 2 + t) * 64 + lane; }
  static constexpr int kInfoRowsMax = 192;
  int info_rows() const { return 128 + 4 * kb; }
  int cell(int g, int c) const {                // (host mirror of the kernel's arithmetic) M cell, -1 if absent
    const uint64_t mask = (uint64_t)row_info[(size_t)g * 4 + 1] | ((uint64_t)row_info[(size_t)g * 4 + 2] << 32);
    if ((mask >> c) & 1) return -1;
    return (int)row_info[(size_t)g * 4] + c - __builtin_popcountll(mask & ((1ull << c) - 1));
  }
};

// which mechanisms the product runs with the dense tail block (the kernel's traits, ros3_kernel.hpp, must agree)
struct DenseConfig { int nd, kb; };
#ifndef MISTRA_TOT_DENSE
#define MISTRA_TOT_DENSE 1
#endif
inline DenseConfig dense_config(const MechTables& m) { return m.nvar == 417 && MISTRA_TOT_DENSE ? DenseConfig{64, 14} : DenseConfig{0, 0}; }

struct KernelSchedule {
  int nt = 0, nw = 0;
  int spt = 0;   // species per thread          s = q*nt + t
  int rpt = 0;   // reactions per thread        r = q*nt + t
  int jpt = 0;   // structurally non-zero Jacobian entries per thread
  int zpt = 0;   // structurally zero (fill-in) entries per thread
  int n_jnz = 0, n_jzero = 0;
  uint32_t ab_base_bytes = 0, jb_base_bytes = 0;      // LDS byte addresses of the A products and of the B products (ros3_kernel.hpp: LdsLayout::AB, JB)
  // Fun_x products: A(r) = RCT(r)*X[f1]*X[f2]*X[f3], padded with the constant 1.0
  int a_trash = 0;                              // first spare cell of the A array: nreact where B has an array of its own, else max(nreact, nb)
  std::vector<uint64_t> fun_fac;                // [rpt*nt]  f1 | f2<<16 | f3<<32 | out<<48   (out = reaction, or a spare cell a_trash + lane for a thread without one)
  GsumProgram vdot;                             // src = A (LDS), output (q,t) = species q*nt+t
  // Jac_SP_x products, grouped under the reaction that owns the rate constant: up to 3 B's per reaction
  std::vector<uint64_t> jac_fac;                // [(q*3 + b)*nt + t]  f1 | f2<<16 | f3<<32 | out<<48 (the spare cell = none)
  GsumProgram jvs;                              // src = B (LDS), output (q,t) = jac0 register q of thread t
  std::vector<uint16_t> jvs_pos;                // [jpt*nt] Ghimj slot (| POS_DIAG) of that output, POS_NONE = idle
  std::vector<uint16_t> zero_pos;               // [zpt*nt] Ghimj slots that Jac_SP_x sets to 0 (| POS_DIAG)
  std::vector<uint16_t> diag_pos;               // [spt*nt] Ghimj slot of (s,s), POS_NONE past nvar
  VmProgram lu, solve;                          // solve = the whole of KppSolve_x as one VM program (kept for tests)
  ScaleProgram lu_scale;                        // runs right after lu
  VmProgram solve_head_fwd, solve_head_bwd;     // head rows (+ head-column part of tail rows) around the tail chain
  int n_temps = 0;                              // temp cells the two head programs use (zeroed by the kernel per solve)
  TailSolve tail;
  DenseTail dense;                              // nd = 0: the mechanism runs without the dense tail block
};

// local_max: rounds of at most this many records are walked by wave 0 alone, runs of them without a barrier in between (0: every round by all waves)
VmProgram build_vm_program(std::vector<VmEntry> entries, const VmLayout& lay, int nt, int merge_budget = 2, int upd_per_rec = VM_UPD_PER_REC,
                           int local_max = 0);
// with_rhs: also forward-sweep the vector held in XS while factorising (rows of an appended right-hand-side column)
// scale_pairs: where to put the (tgt, aux) pairs of the final scaling; nullptr = keep them as a last phase of VM entries
// dense_h >= 0: rows/columns [dense_h, n) are the dense tail block (DenseTail): the program leaves out the updates of
// its slots by pivots >= dense_jm, its pivots' reciprocals and every scaling that involves them.
std::vector<VmEntry> lu_entries(const MechTables& m, const VmLayout& lay, bool with_rhs, int tail_h = -1,
                                std::vector<std::pair<int, int>>* scale_pairs = nullptr, int dense_h = -1, int dense_jm = -1);
DenseTail build_dense_tail(const MechTables& m, const VmLayout& lay, int nd, int kb);
ScaleProgram build_scale_program(const std::vector<std::pair<int, int>>& pairs, const VmLayout& lay, int nt);
int split_long_entries(std::vector<VmEntry>& entries, const VmLayout& lay, int threshold, int first_temp);
std::vector<VmEntry> solve_entries(const MechTables& m, const VmLayout& lay);
std::vector<VmEntry> solve_head_fwd_entries(const MechTables& m, const VmLayout& lay, int h);
std::vector<VmEntry> solve_head_bwd_entries(const MechTables& m, const VmLayout& lay, int h, int split_over = 0, int first_temp = 0,
                                            int* temps_used = nullptr);
TailSolve build_tail_solve(const MechTables& m, const VmLayout& lay, int regs = 0);      // regs = 0: by mechanism size
GsumProgram build_gsum_program(const std::vector<std::vector<std::pair<int, double>>>& outputs,
                               const std::vector<int>& slot_of_output, int nq, int nt, uint32_t src_base_bytes,
                               uint32_t zero_cell_bytes);
// ab_base_bytes: LDS byte address of the A/B product array the gather-sum tables point into (ros3_kernel.hpp: LdsLayout)
// dense_nd = 64 with dense_kb Schur steps: the mechanism's last 64 rows are factorised as a dense block (DenseTail)
// jb_base_bytes: ... of Jac_SP's B products (0: they share the A array)
KernelSchedule build_kernel_schedule(const MechTables& m, int nt, uint32_t ab_base_bytes, int max_temps, int dense_nd = 0,
                                     int dense_kb = 0, uint32_t jb_base_bytes = 0);
std::string describe(const KernelSchedule& s);

}  // namespace mistra
