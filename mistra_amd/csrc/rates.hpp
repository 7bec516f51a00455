// Rate-constant tables (mistra_amd/mech/<mech>.rates, written by tools/extract_rates.py) and the device evaluator's entry.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace mistra {

struct RatesDev {              // device copy of one mechanism's table
  const double* consts;        // literal pool
  const int32_t* offs;         // [nreact + 1] first word of each reaction's postfix program
  const int32_t* words;        // opcode | operand << 8   (0 const, 1 env slot, 2 + 3 - 4 * 5 / 6 neg 7 call function id)
  const int32_t* fslot;        // env slots of what the rate-law functions read from COMMON themselves, -1 = not in this mechanism
  int nreact, nenv;            // reactions; doubles per cell in the input vector ("env", layout: tools/extract_rates.py ENV)
};

struct RatesTable {
  int nreact = 0, nenv = 0;
  std::vector<double> consts;
  std::vector<int32_t> offs, words, fslot;
  bool load(const std::string& path, std::string* err);
};

// st_coeff_a / st_coeff_t (kpp.f90:857-1038 | 664-851) as four tables of the same format, one per setting of the namelist switches
// lpJoyce14bc + 2*lpBuxmann15alph (mistra_amd/mech/<mech>.stcoeff, tools/extract_stcoeff.py); outputs = NSPEC, env = 5 doubles per layer
struct StcoeffTable {
  RatesTable v[4];
  bool load(const std::string& path, std::string* err);
};

hipError_t launch_update_rconst(const RatesDev& R, const double* d_env, double* d_rconst, int ncell, hipStream_t stream);

}  // namespace mistra
