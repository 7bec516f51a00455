// TEST-ONLY host emulator of the kernel's static programs (LDS VM + gather-sum machine, see
// mistra_amd/csrc/schedule.hpp).  It lets the CPU test-suite check the schedule compiler against the oracle
// without a GPU: same words, same per-lane order, lanes executed one after the other, with a hazard check that no
// lane reads an M slot another lane writes in the same round.  It is NOT part of the product library and is never
// a fallback: mistra_amd/ does not link it.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../mistra_amd/csrc/mech_tables.hpp"
#include "../../mistra_amd/csrc/schedule.hpp"

using namespace mistra;

struct Emu {
  MechTables m;
  KernelSchedule s;
  std::string text;
  static constexpr int kMaxTemps = 768;
  VmLayout lay() const { return VmLayout{m.nnz, m.nvar, kMaxTemps}; }
  std::vector<double> fresh_m() const {
    std::vector<double> M((size_t)lay().size(), 0.0);
    M[(size_t)lay().one()] = 1.0;
    M[(size_t)lay().minus_one()] = -1.0;
    return M;
  }
};

static int run_vm(const VmProgram& P, std::vector<double>& M, int trash) {
  const int one_cell = trash - 1;      // VmLayout: ... | 0.0 | 1.0 | trash | -1.0 | R | temps
  const int nt = P.nt;
  std::vector<int> writer(M.size());
  std::vector<uint32_t> row_of_wave((size_t)P.nw, 0);
  for (int r = 0; r < P.nrounds; r++) {
    std::fill(writer.begin(), writer.end(), -1);
    std::vector<std::pair<int, int>> reads;   // (slot, lane)
    std::vector<double> snapshot = M;         // every lane of a round sees the pre-round memory of OTHER lanes
    // a local round (wave 0 ends it with VM_ROW_LOCAL: no barrier) has no rows in the other waves
    bool local_round = false;
    if (P.blk_n[(size_t)r * P.nw] > 0) {
      const size_t last = (size_t)P.wave_base[0] + row_of_wave[0] + P.blk_n[(size_t)r * P.nw] - 1;
      local_round = (P.recs[vm_rec_index(last, 0, 1)] & VM_ROW_LOCAL) != 0;
    }
    for (int w = 0; w < P.nw; w++) {
      const int n = P.blk_n[(size_t)r * P.nw + w];
      if (local_round ? (w > 0 && n != 0) : n == 0) return -12;      // every wave meets every barrier, and only those
      const size_t base = (size_t)P.wave_base[(size_t)w] + row_of_wave[(size_t)w];
      row_of_wave[(size_t)w] += (uint32_t)n;
      for (int l = 0; l < 64; l++) {
        const int lane = w * 64 + l;
        if (lane >= nt) break;
        double carry = 0.0;
        bool own_raw = false;
        for (int ridx = 0; ridx < n; ridx++) {
          uint32_t rec[VM_REC_WORDS];
          for (int k = 0; k < VM_REC_WORDS; k++) rec[k] = P.recs[vm_rec_index(base + (size_t)ridx, l, k)];
          if ((rec[1] & VM_ROW_EOR) ? (ridx != n - 1) : (ridx == n - 1)) return -5;   // round mark on the wrong row
          if ((rec[1] & VM_ROW_LOCAL) && (w != 0 || !(rec[1] & VM_ROW_EOR))) return -11;    // a local round is wave 0's
          if (rec[1] & VM_ROW_NULL) continue;
          auto rd = [&](uint32_t off) {
            const int i = (int)((off & VM_AUX_MASK) >> 3);
            if (i != trash) {
              reads.emplace_back(i, lane);
              if (writer[i] == lane) own_raw = true;   // the kernel prefetches operands: a lane may not read back its own store
            }
            return (writer[i] == lane) ? M[(size_t)i] : snapshot[(size_t)i];
          };
          auto wr = [&](int i, double v) {
            M[(size_t)i] = v;
            if (i == trash) return 0;
            if (writer[i] >= 0 && writer[i] != lane) return -2;   // two lanes write one slot in a round
            writer[i] = lane;
            return 0;
          };
          if ((rec[0] & 7u) || (rec[0] >> 24)) return -10;   // d0 must be a bare address
          const int tgt = (int)(rec[0] >> 3), aux = (int)((rec[1] & VM_AUX_MASK) >> 3);
          double acc = (rec[1] & VM_D1_CONT) ? carry : rd(rec[0]);
          if ((rec[1] & VM_D1_CONT) && ridx == 0) return -7;   // a continuation cannot open a round
          if (P.upd_per_rec == 2) {
            for (int u = 0; u < 2; u++) {
              const double av = rd(rec[2 + 3 * u]), rv = rd(rec[3 + 3 * u]), uv = rd(rec[4 + 3 * u]);
              const double mlt = av * rv;
              const double p = mlt * uv;
              acc = acc - p;
            }
          } else {      // records of the sweeps: three updates of two operands
            for (int u = 0; u < 3; u++) {
              const double av = rd(rec[2 + 2 * u]), uv = rd(rec[3 + 2 * u]);
              const double p = av * uv;
              acc = acc - p;
            }
          }
          if (rec[1] & VM_D1_RCP) {
            if (!(rec[1] & VM_ROW_AUX)) return -6;   // row mark missing
            if (wr(tgt, acc)) return -2;
            if (wr(aux, 1.0 / acc)) return -2;
            carry = acc;
          } else {
            double res = acc;
            if (rec[1] & VM_ROW_AUX) res = acc * rd(rec[1]);        // rows without the mark skip the (then 1.0) factor
            else if ((int)((rec[1] & VM_AUX_MASK) >> 3) != one_cell) return -9;   // unmarked row with a real scale factor
            if (wr(tgt, res)) return -2;
            carry = res;
          }
        }
        if (own_raw) return -8;
      }
    }
    for (auto& rd : reads)
      if (writer[rd.first] >= 0 && writer[rd.first] != rd.second) return -3;   // read of a slot another lane writes this round   // read of a slot that is written in this round (the kernel prefetches operands,
                                              // so not even the writing lane itself may read it back before the barrier)
  }
  return 0;
}

static void run_gsum(const GsumProgram& P, const std::vector<double>& src, uint32_t src_base, uint32_t zero_cell,
                     std::vector<double>& out) {
  out.assign((size_t)P.nq * P.nt, 0.0);
  for (int w = 0; w < P.nw; w++)
    for (int l = 0; l < 64; l++) {
      double acc = -0.0;
      int q = 0;
      for (size_t row = P.wave_base[(size_t)w]; row < (size_t)P.wave_base[(size_t)w] + P.rows[(size_t)w]; row++) {
        const size_t at = (row * 64 + (size_t)l) * 8;
        for (int k = 0; k < 4; k++) {
          const uint32_t addr = P.recs[at + k] & ~7u;
          float cf;
          std::memcpy(&cf, &P.recs[at + 4 + k], 4);
          const double x = addr == zero_cell ? 0.0 : src[(size_t)((addr - src_base) / 8)];
          acc = acc + (double)cf * x;
        }
        if (P.recs[at] & GS_ROW_FLUSH) {
          if (q >= P.nq) std::abort();
          out[(size_t)q++ * P.nt + (size_t)w * 64 + (size_t)l] = acc;
          acc = -0.0;
        }
      }
      if (q != P.nq) std::abort();        // every lane flushes exactly nq times
    }
}

// The kernel's solve: head forward (VM) -> tail chain forward/backward (one wave, registers) -> head backward (VM).
// The tail loops below mirror tail_solve of ros3_kernel.hip statement by statement.
// head_forward = false: the vector went through the LU program (stage 1).  With the dense tail block only its head-column
// terms are in by then, so the tail chain still runs its forward half (ros3_kernel.hip: solve, swept).
static int run_solve_split(const KernelSchedule& s, std::vector<double>& M, const VmLayout& lay, bool head_forward = true) {
  const int nnz = lay.nnz;
  int rc = head_forward ? run_vm(s.solve_head_fwd, M, lay.trash()) : 0;
  if (rc) return rc;
  // columns the tail chain's forward half starts from: all of them; none; or, after the LU program of a mechanism with the
  // dense tail block, the block's own columns (its rows lack exactly those terms)
  const int q_first = head_forward ? 0 : s.dense.nd > 0 ? s.dense.h - s.tail.h : s.tail.m;
  const TailSolve& T = s.tail;
  const int R = T.regs, m = T.m;
  std::vector<double> x((size_t)R * 64), rd((size_t)R * 64);
  for (int i = 0; i < m; i++) {
    x[(size_t)i] = M[(size_t)nnz + T.h + i];
    rd[(size_t)i] = M[(size_t)lay.rdiag(T.h + i)];        // R(k) = 1/U(k,k), published by the LU program
  }
  auto idx = [&](const std::vector<uint32_t>& tab, int pos, int lane, int r) {
    uint32_t w = tab[((size_t)(pos / 4) * 64 + lane) * 4 + pos % 4];
    return (int)(r == 0 ? (w & 0xFFFFu) : (w >> 16));
  };
  for (int q = q_first; q < m; q++) {
    const int rq = q / 64, lq = q % 64;
    const double xq = x[(size_t)rq * 64 + lq];
    for (int r = rq; r < R; r++)
      for (int lane = 0; lane < 64; lane++) {
        const double l = M[(size_t)idx(T.fwd, q, lane, r)];
        x[(size_t)r * 64 + lane] = std::fma(-l, xq, x[(size_t)r * 64 + lane]);      // the kernel's tail chain uses the fused form (ros3_kernel.hip: tail_solve)
      }
  }
  // backward on the row-scaled triangle the LU program leaves in the tail block: x = R .* x; x(i) -= U'(i,q) * x(q)
  for (int i = 0; i < m; i++) x[(size_t)i] = x[(size_t)i] * rd[(size_t)i];
  for (int q = m - 1; q >= 0; q--) {
    const int rq = q / 64, lq = q % 64;
    const double xq = x[(size_t)rq * 64 + lq];
    for (int r = 0; r <= rq; r++)
      for (int lane = 0; lane < 64; lane++) {
        const double u = M[(size_t)idx(T.bwd, m - 1 - q, lane, r)];
        x[(size_t)r * 64 + lane] = std::fma(-u, xq, x[(size_t)r * 64 + lane]);
      }
  }
  for (int i = 0; i < m; i++) M[(size_t)nnz + T.h + i] = x[(size_t)i];
  return run_vm(s.solve_head_bwd, M, lay.trash());
}

// The dense tail block (schedule.hpp: DenseTail) as dense_lu of ros3_kernel.hip works on it: same tables, same tile and
// operand layouts, same order of operations (an MFMA = four fused multiply-adds in k order per element).
static void run_dense(const DenseTail& D, std::vector<double>& M, const VmLayout& lay) {
  static double T[8][2][4][64];
  auto at = [&](int g, int c) { const int p = D.cell(g, c); return (size_t)(p < 0 ? lay.zero() : p); };
  for (int w = 0; w < 8; w++)
    for (int q = 0; q < 2; q++)
      for (int r = 0; r < 4; r++)
        for (int l = 0; l < 64; l++) T[w][q][r][l] = M[at(16 * (w >> 1) + (l >> 4) + 4 * r, 16 * (2 * (w & 1) + q) + (l & 15))];
  for (int k = 0; k < D.kb; k++) {
    for (int w = 0; w < 8; w++)
      for (int q = 0; q < 2; q++)
        for (int r = 0; r < 4; r++)
          for (int l = 0; l < 64; l++) {
            const int row = 16 * (w >> 1) + (l >> 4) + 4 * r, col = 16 * (2 * (w & 1) + q) + (l & 15);
            double acc = T[w][q][r][l];
            for (int kk = 0; kk < 4; kk++) acc = std::fma(-M[at(64 + row, 4 * k + kk)], M[at(128 + 4 * k + kk, col)], acc);
            T[w][q][r][l] = acc;
          }
    for (int i = 0; i < 64; i++)            // the step leaves the multipliers L = W*R in the slots
      for (int kk = 0; kk < 4; kk++)
        if (D.cell(64 + i, 4 * k + kk) >= 0) M[at(64 + i, 4 * k + kk)] = M[at(64 + i, 4 * k + kk)] * M[(size_t)lay.rdiag(D.jm + 4 * k + kk)];
  }
  static double PL[64][4], PU[64][4];
  for (int p = 0; p < D.nd / 4; p++) {
    const int K = p >> 2, s = p & 3, j0 = 4 * p;
    for (int I = K; I < 4; I++)             // the panel's four columns, from the tiles of block column K
      for (int r = 0; r < 4; r++)
        for (int l = 0; l < 64; l++)
          if (((l & 15) >> 2) == s) PL[16 * I + 4 * r + (l >> 4)][l & 3] = T[2 * I + (K >> 1)][K & 1][r][l];
    for (int J = K; J < 4; J++)             // its four rows, from the tiles of block row K
      for (int l = 0; l < 64; l++) PU[16 * J + (l & 15)][l >> 4] = T[2 * K + (J >> 1)][J & 1][s][l];
    double a[64][4], b[64][4], R[4], dg[4];
    for (int l = 0; l < 64; l++)
      for (int k = 0; k < 4; k++) { a[l][k] = PL[l][k]; b[l][k] = PU[l][k]; }
    {   // the kernel factorises the 4x4 diagonal block redundantly in every lane and applies it to the lanes' own entries:
        // the same operations on the same operands as eliminating lane by lane
      double d[4][4];
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) d[r][c] = PU[j0 + c][r];
      for (int k = 0; k < 4; k++) {
        dg[k] = d[k][k];
        R[k] = 1.0 / d[k][k];
        for (int r = k + 1; r < 4; r++) {
          d[r][k] = d[r][k] * R[k];
          for (int c = k + 1; c < 4; c++) d[r][c] = std::fma(-d[r][k], d[k][c], d[r][c]);
        }
      }
      for (int l = 0; l < 64; l++) {
        for (int k = 0; k < 4; k++) {
          for (int k1 = 0; k1 < k; k1++) a[l][k] = std::fma(-a[l][k1], d[k1][k], a[l][k]);
          a[l][k] = a[l][k] * R[k];
        }
        for (int k = 1; k < 4; k++)
          for (int k1 = 0; k1 < k; k1++) b[l][k] = std::fma(-d[k][k1], b[l][k1], b[l][k]);
      }
    }
    for (int l = 0; l < 64; l++)
      for (int k = 0; k < 4; k++) {
        const int jp = j0 + k;
        if (l > jp && D.cell(l, jp) >= 0) M[(size_t)D.cell(l, jp)] = a[l][k];                     // L(h+l, h+jp)
        if (l >= jp && D.cell(jp, l) >= 0) M[(size_t)D.cell(jp, l)] = l == jp ? dg[k] : b[l][k] * R[k];   // U'(h+jp, h+l)
        PL[l][k] = l >= j0 + 4 ? -a[l][k] : 0.0;
        PU[l][k] = l >= j0 + 4 ? b[l][k] : 0.0;
      }
    for (int k = 0; k < 4; k++) M[(size_t)lay.rdiag(D.h + j0 + k)] = R[k];
    const int K2 = (j0 + 4) >> 4;
    for (int w = 0; w < 8; w++)
      for (int q = 0; q < 2; q++) {
        const int I = w >> 1, J = 2 * (w & 1) + q;
        if (I < K2 || J < K2) continue;
        for (int r = 0; r < 4; r++)
          for (int l = 0; l < 64; l++) {
            double acc = T[w][q][r][l];
            for (int kk = 0; kk < 4; kk++) acc = std::fma(PL[16 * I + (l >> 4) + 4 * r][kk], PU[16 * J + (l & 15)][kk], acc);
            T[w][q][r][l] = acc;
          }
      }
  }
}

extern "C" {

void* emu_create(const char* mech_path, int nt) {
  Emu* e = new Emu;
  std::string err;
  if (!e->m.load(mech_path, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); delete e; return nullptr; }
  try {
    const DenseConfig dc = nt >= 512 ? dense_config(e->m) : DenseConfig{0, 0};      // as the product configures the mechanism
    e->s = build_kernel_schedule(e->m, nt, 8u * (uint32_t)(e->lay().size() + 1000), Emu::kMaxTemps, dc.nd, dc.kb);   // any base will do here
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "schedule: %s\n", ex.what());
    delete e;
    return nullptr;
  }
  e->text = describe(e->s);
  return e;
}
void emu_destroy(void* h) { delete (Emu*)h; }
int emu_tail_h(void* h) { return ((Emu*)h)->s.tail.h; }
int emu_dense_nd(void* h) { return ((Emu*)h)->s.dense.nd; }
const char* emu_describe(void* h) { return ((Emu*)h)->text.c_str(); }

// KppDecomp on G (nnz doubles, in place) through the LU program; R (nvar) receives the pivot reciprocals the program
// publishes.  Returns 0 or a negative hazard code.
// LU: factors as the kernel's LU program leaves them (tail block's upper triangle row-scaled), or, with
// reference_form, as KppDecomp leaves them: the tail rows are then scaled here.
static std::vector<double> solve_memory(const Emu* e, const double* LU, const double* X, bool reference_form) {
  std::vector<double> M = e->fresh_m();
  std::memcpy(M.data(), LU, sizeof(double) * e->m.nnz);
  std::memcpy(M.data() + e->m.nnz, X, sizeof(double) * e->m.nvar);
  for (int k = 0; k < e->m.nvar; k++) M[(size_t)e->lay().rdiag(k)] = 1.0 / LU[e->m.diag[(size_t)k]];
  if (reference_form)
    for (int k = e->s.tail.h; k < e->m.nvar; k++)
      for (int p = e->m.diag[(size_t)k] + 1; p < e->m.crow[(size_t)k + 1]; p++) M[(size_t)p] = M[(size_t)p] * M[(size_t)e->lay().rdiag(k)];
  return M;
}

// X (nvar, optional): right-hand side in, its forward sweep L^-1 b out (the LU program carries it along).
int emu_lu(void* h, double* G, double* R, double* X) {
  Emu* e = (Emu*)h;
  std::vector<double> M = e->fresh_m();
  std::memcpy(M.data(), G, sizeof(double) * e->m.nnz);
  if (X) std::memcpy(M.data() + e->m.nnz, X, sizeof(double) * e->m.nvar);
  int rc = run_vm(e->s.lu, M, e->lay().trash());
  {   // the scaling pass that follows the LU program (ros3_kernel.hip: scale_run)
    const ScaleProgram& sp = e->s.lu_scale;
    const size_t wave_slots = (size_t)sp.nslots + VM_LOOKAHEAD_ROWS;
    for (int w = 0; w < sp.nw; w++)
      for (int sl = 0; sl < sp.nslots; sl++)
        for (int l = 0; l < 64; l++)
          for (int k = 0; k < 2; k++) {
            const size_t at = (((size_t)w * wave_slots + (size_t)sl) * 64 + (size_t)l) * 4 + 2 * (size_t)k;
            const size_t tgt = sp.recs[at] / 8, aux = sp.recs[at + 1] / 8;
            M[tgt] = M[tgt] * M[aux];
          }
  }
  if (e->s.dense.nd > 0) run_dense(e->s.dense, M, e->lay());
  std::memcpy(G, M.data(), sizeof(double) * e->m.nnz);
  if (R) std::memcpy(R, M.data() + e->lay().rdiag(), sizeof(double) * e->m.nvar);
  if (X) std::memcpy(X, M.data() + e->m.nnz, sizeof(double) * e->m.nvar);
  return rc;
}

// backward half of the kernel's solve (tail chain backward + head backward) on a vector already forward-swept
int emu_solve_backward(void* h, const double* LU, double* X) {
  Emu* e = (Emu*)h;
  std::vector<double> M = solve_memory(e, LU, X, false);     // LU comes from emu_lu
  int rc = run_solve_split(e->s, M, e->lay(), false);
  std::memcpy(X, M.data() + e->m.nnz, sizeof(double) * e->m.nvar);
  return rc;
}

// the kernel's whole solve (head forward, tail chain, head backward) on factors in the kernel's own form (from emu_lu, or the
// GPU's first-step dump)
int emu_solve_kernel_form(void* h, const double* LU, double* X) {
  Emu* e = (Emu*)h;
  std::vector<double> M = solve_memory(e, LU, X, false);
  int rc = run_solve_split(e->s, M, e->lay());
  std::memcpy(X, M.data() + e->m.nnz, sizeof(double) * e->m.nvar);
  return rc;
}

int emu_solve(void* h, const double* LU, double* X) {
  Emu* e = (Emu*)h;
  std::vector<double> M = solve_memory(e, LU, X, false);     // whole-solve VM program: reference-form factors, unscaled
  int rc = run_vm(e->s.solve, M, e->lay().trash());
  std::memcpy(X, M.data() + e->m.nnz, sizeof(double) * e->m.nvar);
  return rc;
}

int emu_solve_split(void* h, const double* LU, double* X) {
  Emu* e = (Emu*)h;
  std::vector<double> M = solve_memory(e, LU, X, true);      // reference-form factors in
  int rc = run_solve_split(e->s, M, e->lay());
  std::memcpy(X, M.data() + e->m.nnz, sizeof(double) * e->m.nvar);
  return rc;
}

static void make_x(const Emu* e, const double* V, const double* F, std::vector<double>& X) {
  X.resize((size_t)e->m.nx());
  std::memcpy(X.data(), V, sizeof(double) * e->m.nvar);
  std::memcpy(X.data() + e->m.nvar, F, sizeof(double) * e->m.nfix);
  std::memcpy(X.data() + e->m.nspec(), e->m.consts.data(), sizeof(double) * e->m.nconst);
}

void emu_fun(void* h, const double* V, const double* F, const double* RCT, double* Vdot) {
  Emu* e = (Emu*)h;
  const KernelSchedule& s = e->s;
  std::vector<double> X, A((size_t)s.rpt * s.nt + 1, 0.0), out;
  make_x(e, V, F, X);
  for (int q = 0; q < s.rpt; q++)
    for (int t = 0; t < s.nt; t++) {
      uint64_t w = s.fun_fac[(size_t)q * s.nt + t];
      const int out = (int)((w >> 48) & 0xFFFF), r = q * s.nt + t;
      if (out >= s.a_trash) continue;       // a spare cell (one per lane): this thread owns no reaction in slot q
      if (out != r) std::abort();
      double a = RCT[r];
      a = a * X[w & 0xFFFF];
      a = a * X[(w >> 16) & 0xFFFF];
      a = a * X[(w >> 32) & 0xFFFF];
      A[(size_t)r] = a;
    }
  run_gsum(s.vdot, A, s.ab_base_bytes, 8u * (uint32_t)e->lay().zero(), out);
  for (int i = 0; i < e->m.nvar; i++) Vdot[i] = out[(size_t)i];
}

// Jac_SP followed by the kernel's matrix preparation: G = -Jac0 (+ghinv on the diagonal), fill-in slots -0.0.
// With ghinv = 0 and negate = 0 the plain JVS array comes back (fill-in slots +0.0).
void emu_jac_prepare(void* h, const double* V, const double* F, const double* RCT, double ghinv, int negate, double* G) {
  Emu* e = (Emu*)h;
  const KernelSchedule& s = e->s;
  std::vector<double> X, B((size_t)e->m.nb + 1, 0.0), jac0;
  make_x(e, V, F, X);
  for (int q = 0; q < s.rpt; q++)
    for (int b = 0; b < 3; b++)
      for (int t = 0; t < s.nt; t++) {
        uint64_t w = s.jac_fac[((size_t)q * 3 + b) * s.nt + t];
        int out = (int)((w >> 48) & 0xFFFF);
        if (out >= std::max(e->m.nreact, e->m.nb)) continue;
        double v = RCT[q * s.nt + t];
        v = v * X[w & 0xFFFF];
        v = v * X[(w >> 16) & 0xFFFF];
        v = v * X[(w >> 32) & 0xFFFF];
        B[(size_t)out] = v;
      }
  run_gsum(s.jvs, B, s.jb_base_bytes, 8u * (uint32_t)e->lay().zero(), jac0);
  for (int i = 0; i < e->m.nnz; i++) G[i] = std::nan("");
  for (size_t i = 0; i < s.jvs_pos.size(); i++) {
    uint16_t p = s.jvs_pos[i];
    if (p == POS_NONE) continue;
    double v = negate ? -jac0[i] : jac0[i];
    if (p & POS_DIAG) v = v + ghinv;
    G[p & 0x7FFF] = v;
  }
  for (size_t i = 0; i < s.zero_pos.size(); i++) {
    uint16_t p = s.zero_pos[i];
    if (p == POS_NONE) continue;
    double v = negate ? -0.0 : 0.0;
    if (p & POS_DIAG) v = v + ghinv;
    G[p & 0x7FFF] = v;
  }
}

}  // extern "C"

// per-round critical slot counts of a VM program (which: 0 = LU, 1 = solve); returns the number of rounds
extern "C" int emu_round_profile(void* h, int which, int* crit, int* total, int cap) {
  Emu* e = (Emu*)h;
  const VmProgram& P = which ? e->s.solve : e->s.lu;
  for (int r = 0; r < P.nrounds && r < cap; r++) {
    int c = 0, t = 0;
    for (int w = 0; w < P.nw; w++) {
      int n = P.blk_n[(size_t)r * P.nw + w];
      c = n > c ? n : c;
      t += n;
    }
    crit[r] = c;
    total[r] = t;
  }
  return P.nrounds;
}

// per-round profile of the head-sweep programs (which: 0 = head forward, 1 = head backward)
extern "C" int emu_head_profile(void* h, int which, int* crit, int* total, int cap) {
  Emu* e = (Emu*)h;
  const VmProgram& P = which ? e->s.solve_head_bwd : e->s.solve_head_fwd;
  for (int r = 0; r < P.nrounds && r < cap; r++) {
    int c = 0, t = 0;
    for (int w = 0; w < P.nw; w++) {
      int n = P.blk_n[(size_t)r * P.nw + w];
      c = n > c ? n : c;
      t += n;
    }
    crit[r] = c;
    total[r] = t;
  }
  return P.nrounds;
}

// rows of each wave's gather-sum stream (which: 0 = Fun_x's sums, 1 = Jac_SP_x's); returns the number of waves
extern "C" int emu_gsum_rows(void* h, int which, int* rows, int cap) {
  Emu* e = (Emu*)h;
  const GsumProgram& P = which ? e->s.jvs : e->s.vdot;
  for (int w = 0; w < P.nw && w < cap; w++) rows[w] = P.rows[(size_t)w];
  return P.nw;
}
