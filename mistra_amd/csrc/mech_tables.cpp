#include "mech_tables.hpp"

#include <cstdio>
#include <cstring>

namespace mistra {

namespace {
constexpr int32_t kMagic = 0x48434D4B;  // 'KMCH'
constexpr int32_t kVersion = 2;

template <class T>
bool take(const std::vector<char>& raw, size_t& off, size_t n, std::vector<T>& out) {
  if (off + n * sizeof(T) > raw.size()) return false;
  out.resize(n);
  std::memcpy(out.data(), raw.data() + off, n * sizeof(T));
  off += n * sizeof(T);
  return true;
}
}  // namespace

bool MechTables::load(const std::string& path, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) {
    if (err) *err = "cannot open mechanism table " + path;
    return false;
  }
  std::fseek(f, 0, SEEK_END);
  long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<char> raw((size_t)sz);
  size_t got = std::fread(raw.data(), 1, raw.size(), f);
  std::fclose(f);
  if (got != raw.size() || raw.size() < 48) {
    if (err) *err = "short read on " + path;
    return false;
  }
  int32_t h[12];
  std::memcpy(h, raw.data(), sizeof h);
  if (h[0] != kMagic || h[1] != kVersion) {
    if (err) *err = path + ": not a KMCH v2 mechanism table";
    return false;
  }
  nvar = h[2]; nfix = h[3]; nreact = h[4]; nnz = h[5];
  const int n_afac = h[6];
  nb = h[7];
  const int n_bfac = h[8], n_vd = h[9], n_jv = h[10];
  nconst = h[11];
  size_t off = 48;
  bool ok = take(raw, off, (size_t)nvar + 1, crow) && take(raw, off, (size_t)nnz, icol) &&
            take(raw, off, (size_t)nvar, diag) && take(raw, off, (size_t)nreact + 1, a_ptr) &&
            take(raw, off, (size_t)n_afac, a_fac) && take(raw, off, (size_t)nb, b_rct) &&
            take(raw, off, (size_t)nb + 1, b_ptr) && take(raw, off, (size_t)n_bfac, b_fac) &&
            take(raw, off, (size_t)nvar + 1, vd_ptr) && take(raw, off, (size_t)n_vd, vd_idx) &&
            take(raw, off, (size_t)nnz + 1, jv_ptr) && take(raw, off, (size_t)n_jv, jv_idx);
  off += (8 - off % 8) % 8;
  ok = ok && take(raw, off, (size_t)n_vd, vd_coef) && take(raw, off, (size_t)n_jv, jv_coef) &&
       take(raw, off, (size_t)nconst, consts);
  if (!ok || off != raw.size()) {
    if (err) *err = path + ": truncated or oversized mechanism table";
    return false;
  }
  // structural sanity: the kernels index LDS with these numbers
  for (int k = 0; k < nvar; k++) {
    if (crow[k] > diag[k] || diag[k] >= crow[k + 1] || icol[diag[k]] != k) {
      if (err) *err = path + ": inconsistent CSR pattern";
      return false;
    }
    for (int p = crow[k] + 1; p < crow[k + 1]; p++)
      if (icol[p] <= icol[p - 1]) {
        if (err) *err = path + ": CSR columns not ascending";
        return false;
      }
  }
  if (consts.empty() || consts[0] != 1.0) {
    if (err) *err = path + ": constant slot 0 must be 1.0 (padding factor)";
    return false;
  }
  return true;
}

}  // namespace mistra
