// Mechanism tables (.mech, written by tools/extract_mech.py) — host-side loader used by the schedule compiler.
// The tables are the data form of the reference's generated Fun_x / Jac_SP_x term lists and LU sparsity
// (gas.f:2043,2656,6718 | aer.f:2741,4368,23480 | tot.f:4145,6845,44435).  0-based indices.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mistra {

struct MechTables {
  int nvar = 0, nfix = 0, nreact = 0, nnz = 0, nb = 0, nconst = 0;
  std::vector<int32_t> crow, icol, diag;        // CSR of the LU pattern (incl. fill-in); diag[k] = slot of (k,k)
  std::vector<int32_t> a_ptr, a_fac;            // A(i) = RCT(i) * prod X[a_fac[..]],  X = [V | F | consts]
  std::vector<int32_t> b_rct, b_ptr, b_fac;     // B(m) = RCT(b_rct[m]) * prod X[b_fac[..]]
  std::vector<int32_t> vd_ptr, vd_idx;          // Vdot(j) = sum vd_coef * A(vd_idx)
  std::vector<int32_t> jv_ptr, jv_idx;          // JVS(k)  = sum jv_coef * B(jv_idx); empty list = structural zero
  std::vector<double> vd_coef, jv_coef, consts;

  int nspec() const { return nvar + nfix; }
  int nx() const { return nvar + nfix + nconst; }          // length of the extended vector X
  bool load(const std::string& path, std::string* err);
};

}  // namespace mistra
