#include "schedule.hpp"

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <numeric>
#include <stdexcept>

namespace mistra {

namespace {

// Schedule-compiler switches for same-box A/B measurements.  They change the programs (and with them round-off), so the
// product library does not read the environment: only a diagnostic build (-DMISTRA_DIAG_ENV, tools/diag_dense.sh env) does.
inline const char* diag_env(const char* name) {
#ifdef MISTRA_DIAG_ENV
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

struct Item {            // one chunk of one entry inside one round
  int entry;
  int first, count;      // range of entry.upd
  bool final;            // completes the entry: apply MULR / RCP
};

uint32_t vm_off(int idx, uint32_t flags) {
  if (idx < 0 || idx >= (1 << 21)) throw std::logic_error("VM index out of range");
  return ((uint32_t)idx << 3) | flags;
}

// LDS bank class of an M cell for 8-byte accesses (M starts at LDS address 0): ds_read_b64 / ds_write_b64 serve 32 lanes
// per pass and two lanes of a pass collide when their DIFFERENT addresses fall on the same pair of the 64 banks
// (MI355X_MICROARCH.md §LDS); equal addresses are broadcast.
inline int bank_class(int idx) { return idx & 31; }

// LDS cycles of one 8-byte wave access with these per-lane M cells (-1 = none): per half-wave, the largest number of
// distinct addresses on one bank pair.
int lds_pass_cycles(const int* cell64) {
  int total = 0;
  for (int half = 0; half < 2; half++) {
    int seen[32][8], nseen[32] = {0}, worst = 0;
    for (int l = 0; l < 32; l++) {
      int c = cell64[half * 32 + l];
      if (c < 0) continue;
      int b = bank_class(c), k = 0;
      while (k < nseen[b] && seen[b][k] != c) k++;
      if (k == nseen[b] && nseen[b] < 8) seen[b][nseen[b]++] = c;
      worst = std::max(worst, nseen[b]);
    }
    total += std::max(worst, 1);
  }
  return total;
}

// The deal of rounds 1-3, kept for the eight-wave workgroups (tot), where it measured 0.4 % faster than the longest-first deal below although
// its rounds are a quarter longer in rows (same box, round 4; for four waves and one the longest-first deal wins: aer +0.3 %, gas +3.7 %).
// Deal the items of one round (sorted by decreasing record count) over lanes.  Lane LOADS follow the snake deal (pass p
// gives every lane its p-th item, alternate passes run backwards); small sets are packed into few waves.  WHICH item of a
// run of equally long items goes to which lane of the pass is chosen against LDS bank conflicts: the dense LU rounds are
// bound by the LDS pipe, whose gathers cost ~4 array cycles per wave instruction when the 32 lanes of a pass are
// thrown at the banks at random and 1 when they are spread.  Records of one pass sit in (nearly) the same rows of
// their lanes, i.e. are issued together; an item goes to the half-wave in which its operand cells meet the fewest
// distinct cells of their bank classes already placed there.
std::vector<int> deal_snake(const std::vector<Item>& items, const std::vector<VmEntry>& entries, int nt, int upr, std::vector<int>& row0) {
  const int n = (int)items.size();
  std::vector<int> lane((size_t)n);
  static const bool plain_deal = diag_env("MISTRA_DIAG_PLAIN_DEAL") != nullptr;   // A/B diagnostic: positions as sorted
  const int waves_used = std::max(1, std::min(nt / 64, (n + 63) / 64));
  const int lanes = waves_used * 64, groups = lanes / 32;
  auto nrec = [upr](const Item& it) { return std::max(1, (it.count + upr - 1) / upr); };
  struct Slot { std::vector<uint16_t> cells; int cnt[32]; };
  for (int p0 = 0; p0 < n; p0 += lanes) {                 // one pass
    const int pn = std::min(lanes, n - p0), pass = p0 / lanes;
    std::vector<int> pos_lane((size_t)pn);
    for (int j = 0; j < pn; j++) pos_lane[(size_t)j] = (pass & 1) ? (lanes - 1 - j) : j;
    const int rmax = nrec(items[(size_t)p0]);
    std::vector<Slot> slot((size_t)groups * rmax * 7);      // [group][record][operand]: cells present, per bank class
    for (Slot& sl : slot) std::fill(sl.cnt, sl.cnt + 32, 0);
    auto operands = [&](const Item& it, int r, int* cell /*7*/) {
      const VmEntry& E = entries[(size_t)it.entry];
      cell[0] = E.tgt;
      const int per = upr == 2 ? 3 : 2;      // operand cells per update: (a, r, u), or (a, u) in records of three updates
      for (int u = 0; u < upr; u++) {
        const int i = r * upr + u;
        if (i < it.count) {
          const VmUpd& up = E.upd[(size_t)(it.first + i)];
          cell[1 + per * u] = up.a;
          if (upr == 2) { cell[2 + 3 * u] = up.r; cell[3 + 3 * u] = up.u; }
          else cell[2 + 2 * u] = up.u;
        } else {
          for (int o = 0; o < per; o++) cell[1 + per * u + o] = -1;      // the 0.0 cell in every lane: broadcast
        }
      }
    };
    for (int j0 = 0; j0 < pn;) {                          // one run of equally long items
      int j1 = j0;
      while (j1 < pn && nrec(items[(size_t)(p0 + j1)]) == nrec(items[(size_t)(p0 + j0)])) j1++;
      const int nr = nrec(items[(size_t)(p0 + j0)]);
      std::vector<std::vector<int>> free_lanes((size_t)groups);      // lanes of this run, by half-wave
      for (int j = j0; j < j1; j++) free_lanes[(size_t)(pos_lane[(size_t)j] / 32)].push_back(pos_lane[(size_t)j]);
      for (int j = j0; j < j1; j++) {
        const Item& it = items[(size_t)(p0 + j)];
        int best = -1, best_cost = 0;
        for (int g = 0; g < groups; g++) {
          if (free_lanes[(size_t)g].empty()) continue;
          int cost = 0;
          for (int r = 0; r < nr; r++) {
            int cell[7];
            operands(it, r, cell);
            for (int o = 0; o < 7; o++) {
              if (cell[o] < 0) continue;
              const Slot& sl = slot[((size_t)g * rmax + r) * 7 + o];
              if (std::find(sl.cells.begin(), sl.cells.end(), (uint16_t)cell[o]) != sl.cells.end()) continue;   // broadcast
              cost += (o == 0 ? 2 : 1) * sl.cnt[bank_class(cell[o])];      // the target is read and written
            }
          }
          if (best < 0 || cost < best_cost) { best = g; best_cost = cost; }
        }
        if (plain_deal) best = pos_lane[(size_t)j] / 32;
        if (plain_deal) {
          std::vector<int>& fl = free_lanes[(size_t)best];
          fl.erase(std::find(fl.begin(), fl.end(), pos_lane[(size_t)j]));
          lane[(size_t)(p0 + j)] = pos_lane[(size_t)j];
        } else {
          lane[(size_t)(p0 + j)] = free_lanes[(size_t)best].back();
          free_lanes[(size_t)best].pop_back();
        }
        for (int r = 0; r < nr; r++) {
          int cell[7];
          operands(it, r, cell);
          for (int o = 0; o < 7; o++) {
            if (cell[o] < 0) continue;
            Slot& sl = slot[((size_t)best * rmax + r) * 7 + o];
            if (std::find(sl.cells.begin(), sl.cells.end(), (uint16_t)cell[o]) != sl.cells.end()) continue;
            sl.cells.push_back((uint16_t)cell[o]);
            sl.cnt[bank_class(cell[o])]++;
          }
        }
      }
      j0 = j1;
    }
  }
  // first rows: a lane walks its items in the order dealt
  row0.assign((size_t)n, 0);
  {
    std::vector<int> ld((size_t)lanes, 0);
    for (int k = 0; k < n; k++) { row0[(size_t)k] = ld[(size_t)lane[(size_t)k]]; ld[(size_t)lane[(size_t)k]] += nrec(items[(size_t)k]); }
  }
  return lane;
}

// Deal the items of one round (sorted by decreasing record count) over lanes.  row0: the items' first record rows within their lanes
// (a lane walks its items in the order of their rows).
//   * Rows: longest-first into lanes that still have room under the bound R = max(longest item, records / lanes) keeps every lane at
//     or below R (until round 4 a snake deal — pass p gives every lane its p-th item — left the lanes that drew the long items of pass 0
//     up to three rows above the rest: tot's LU program had 102 critical rows where 79 suffice).
//   * Reciprocals: the record that publishes a pivot's reciprocal carries an IEEE division (~150 cycles of dependent work, paid by its
//     whole wave) behind its row.  Those items go first, side by side into the LAST waves, whose lanes then take one row less than the
//     others: the division runs while the other waves walk their last row.  (The snake deal hid it by accident — a post-pass moved
//     the record into a wave that happened to be emptier; with level waves and the division on top of a full one the 23 rows saved
//     bought nothing: measured.)
//   * Banks: an 8-byte gather costs one LDS array cycle per half-wave when its 32 lanes are spread over the bank pairs and as many as the
//     fullest bank pair holds DIFFERENT cells otherwise (equal cells are broadcast).  Lanes of one half-wave with the same load are
//     interchangeable, so the choice is (half-wave, first row): the one where the item's operand cells raise those maxima least, then
//     where they meet the fewest cells of their bank classes.
// Small sets are packed into few waves.  Which lane walks an item changes no arithmetic: every entry keeps its update order.
// (Tried on top, dropped: sweeps of pairwise swaps between half-waves of one-record items that share a row, accepted where the modelled
// cycles of both half-waves' seven gathers drop — 1.7 % fewer modelled cycles for 1.6 s of start-up time per sweep.)
std::vector<int> deal(const std::vector<Item>& items, const std::vector<VmEntry>& entries, int nt, int upr, std::vector<int>& row0, int max_waves = 0) {
  const int n = (int)items.size();
  std::vector<int> lane((size_t)n);
  row0.assign((size_t)n, 0);
  static const bool plain_deal = diag_env("MISTRA_DIAG_PLAIN_DEAL") != nullptr;   // A/B diagnostic: the snake deal, positions as sorted
  auto nrec = [upr](const Item& it) { return std::max(1, (it.count + upr - 1) / upr); };
  auto publishes = [&](const Item& it) { return it.final && entries[(size_t)it.entry].rcp >= 0; };
  long total = 0;
  int n_rcp = 0, longest = 1;
  for (const Item& it : items) { total += nrec(it); n_rcp += publishes(it); longest = std::max(longest, nrec(it)); }
  // How many waves take part: W waves of R = max(longest item, records / 64 W) rows; the W with the smaller  max(c_lds W R, c_row R),
  // then the fewer rows, then the shorter round.  c_lds = 0 (as shipped): the fewest waves that reach the smallest R.
  static const int c_lds = diag_env("MISTRA_DIAG_DEAL_CLDS") ? std::atoi(diag_env("MISTRA_DIAG_DEAL_CLDS")) : 0;
  static const int c_row = diag_env("MISTRA_DIAG_DEAL_CROW") ? std::atoi(diag_env("MISTRA_DIAG_DEAL_CROW")) : 200;
  static const bool rcp_slack = diag_env("MISTRA_DIAG_NO_RCP_SLACK") == nullptr;
  const int rcp_waves = rcp_slack ? (n_rcp + 63) / 64 : 0;
  int waves_used = 1;
  {
    long best_t = -1, best_rows = 0;
    for (int w = 1; w <= (max_waves > 0 ? max_waves : nt / 64) && !plain_deal; w++) {
      const long r = std::max<long>(longest, (total + 64L * std::min(rcp_waves, w) + 64L * w - 1) / (64L * w));
      const long t = std::max<long>((long)c_lds * w * r, (long)c_row * r);
      if (best_t < 0 || t < best_t || (t == best_t && w * r <= best_rows)) { best_t = t; best_rows = w * r; waves_used = w; }
    }
    if (plain_deal) waves_used = std::max(1, std::min(max_waves > 0 ? max_waves : nt / 64, (n + 63) / 64));
  }
  const int lanes = waves_used * 64, groups = lanes / 32;
  if (plain_deal) {
    for (int k = 0; k < n; k++) {
      const int pass = k / lanes, j = k % lanes;
      lane[(size_t)k] = (pass & 1) ? (lanes - 1 - j) : j;
    }
    std::vector<int> ld((size_t)lanes, 0);
    for (int k = 0; k < n; k++) { row0[(size_t)k] = ld[(size_t)lane[(size_t)k]]; ld[(size_t)lane[(size_t)k]] += nrec(items[(size_t)k]); }
    return lane;
  }
  const int rw = std::min(rcp_waves, waves_used);      // the last rw waves take the publishing items and one row less
  int R = std::max<int>(longest, (int)((total + 64L * rw + lanes - 1) / lanes));
  auto cap = [&](int l) { return (l / 64 >= waves_used - rw && R > 1) ? R - 1 : R; };
  struct Slot { uint16_t cells[32]; uint8_t cnt[32]; uint8_t n, worst; };      // one gather of one half-wave: distinct cells, per bank class
  std::vector<Slot> slot;                                                     // [group][row][operand]
  int rows_cap = 0;
  auto grow = [&](int rows) {
    if (rows <= rows_cap) return;
    std::vector<Slot> s2((size_t)groups * rows * 7);
    for (Slot& sl : s2) { std::fill(sl.cnt, sl.cnt + 32, 0); sl.n = 0; sl.worst = 0; }
    for (int g = 0; g < groups; g++)
      for (int r = 0; r < rows_cap; r++)
        for (int o = 0; o < 7; o++) s2[((size_t)g * rows + r) * 7 + o] = slot[((size_t)g * rows_cap + r) * 7 + o];
    slot.swap(s2);
    rows_cap = rows;
  };
  grow(R + 2);
  auto operands = [&](const Item& it, int r, int* cell /*7*/) {
    const VmEntry& E = entries[(size_t)it.entry];
    cell[0] = E.tgt;
    const int per = upr == 2 ? 3 : 2;      // operand cells per update: (a, r, u), or (a, u) in records of three updates
    for (int u = 0; u < upr; u++) {
      const int i = r * upr + u;
      if (i < it.count) {
        const VmUpd& up = E.upd[(size_t)(it.first + i)];
        cell[1 + per * u] = up.a;
        if (upr == 2) { cell[2 + 3 * u] = up.r; cell[3 + 3 * u] = up.u; }
        else cell[2 + 2 * u] = up.u;
      } else {
        for (int o = 0; o < per; o++) cell[1 + per * u + o] = -1;      // the 0.0 cell in every lane: broadcast
      }
    }
  };
  auto present = [](const Slot& sl, int c) {
    for (int k = 0; k < sl.n; k++)
      if (sl.cells[k] == (uint16_t)c) return true;
    return false;
  };
  std::vector<int> load((size_t)lanes, 0);
  std::vector<int> cells((size_t)16 * 7);
  // place item k into one of the half-waves [g_lo, g_hi)
  auto place = [&](int k, int g_lo, int g_hi) {
    const Item& it = items[(size_t)k];
    const int nr = nrec(it);
    if ((int)cells.size() < nr * 7) cells.resize((size_t)nr * 7);
    for (int r = 0; r < nr; r++) operands(it, r, &cells[(size_t)r * 7]);
    int best_lane = -1;
    long best_hard = 0, best_soft = 0;
    for (int pass = 0; pass < 2 && best_lane < 0; pass++) {      // second pass (nothing fits under the bound): the bound gives way
      for (int g = g_lo; g < g_hi; g++) {
        // candidates: one lane per distinct load of the half-wave (a one-record item fits wherever a row is free; longer ones go to
        // the emptiest lane only, so that the lanes do not fragment under the bound)
        uint64_t tried = 0;
        int lmin = g * 32;
        for (int q = 1; q < 32; q++)
          if (load[(size_t)(g * 32 + q)] < load[(size_t)lmin]) lmin = g * 32 + q;
        for (int q = 0; q < 32; q++) {
          const int l = (nr == 1 && pass == 0) ? g * 32 + q : lmin;
          const int r0 = load[(size_t)l];
          if (r0 < 64 && ((tried >> r0) & 1)) continue;
          if (r0 < 64) tried |= 1ull << r0;
          if (pass == 0 && r0 + nr > cap(l)) continue;
          grow(r0 + nr);
          long hard = 0, soft = 0;
          for (int r = 0; r < nr; r++)
            for (int o = 0; o < 7; o++) {
              const int c = cells[(size_t)r * 7 + o];
              if (c < 0) continue;
              const Slot& sl = slot[((size_t)g * rows_cap + r0 + r) * 7 + o];
              if (present(sl, c)) continue;                                   // broadcast
              const int w = o == 0 ? 2 : 1;                                   // the target is read and written
              const int cnt = sl.cnt[bank_class(c)];
              if (cnt + 1 > std::max<int>(sl.worst, 1)) hard += w;
              soft += w * cnt;
            }
          // (a later first row is a fuller lane: among equals the emptier one, which keeps the waves' row counts level)
          if (best_lane < 0 || hard < best_hard || (hard == best_hard && (soft < best_soft || (soft == best_soft && r0 < load[(size_t)best_lane])))) {
            best_lane = l; best_hard = hard; best_soft = soft;
          }
          if (!(nr == 1 && pass == 0)) break;
        }
      }
    }
    const int g = best_lane / 32, r0 = load[(size_t)best_lane];
    lane[(size_t)k] = best_lane;
    row0[(size_t)k] = r0;
    load[(size_t)best_lane] += nr;
    if (r0 + nr > cap(best_lane)) R = std::max(R + (cap(best_lane) < R ? 1 : 0), r0 + nr);      // (the bound gave way)
    for (int r = 0; r < nr; r++)
      for (int o = 0; o < 7; o++) {
        const int c = cells[(size_t)r * 7 + o];
        if (c < 0) continue;
        Slot& sl = slot[((size_t)g * rows_cap + r0 + r) * 7 + o];
        if (present(sl, c)) continue;
        if (sl.n < 32) sl.cells[sl.n++] = (uint16_t)c;
        const int b = bank_class(c);
        sl.cnt[b]++;
        sl.worst = std::max(sl.worst, sl.cnt[b]);
      }
  };
  if (rw > 0)
    for (int k = 0; k < n; k++)
      if (publishes(items[(size_t)k])) place(k, groups - 2 * rw, groups);
  for (int k = 0; k < n; k++)
    if (!(rw > 0 && publishes(items[(size_t)k]))) place(k, 0, groups);
  return lane;
}

int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
VmProgram build_vm_program(std::vector<VmEntry> entries, const VmLayout& lay, int nt, int merge_budget, int upr, int local_max) {
  if (upr != 2 && upr != 3) throw std::invalid_argument("records hold two (a, r, u) or three (a, u) updates");
  if (nt % 64 != 0 || nt <= 0) throw std::invalid_argument("nt must be a positive multiple of 64");
  const int msize = lay.size(), zero_slot = lay.zero();
  VmProgram P;
  P.nt = nt;
  P.nw = nt / 64;
  P.zero_slot = zero_slot;
  P.upd_per_rec = upr;

  std::vector<int> fin((size_t)msize, 0);   // round at whose end M[x] holds its final value (0 = input)
  std::map<int, std::vector<Item>> rounds;
  int base = 0, phase_max = 0;
  int phase = entries.empty() ? 0 : entries[0].phase;

  for (size_t e = 0; e < entries.size(); e++) {
    VmEntry& E = entries[e];
    if (E.phase != phase) {
      if (E.phase < phase) throw std::invalid_argument("entries must be ordered by phase");
      phase = E.phase;
      base = std::max(base, phase_max);     // the next phase starts after every round of the previous one
    }
    const int n = (int)E.upd.size();
    const bool has_final_op = E.mulr >= 0 || E.rcp >= 0;
    if (n == 0 && !has_final_op) continue;  // nothing to do, value stays as it is
    std::vector<int> ru((size_t)n);
    for (int i = 0; i < n; i++) {
      const VmUpd& u = E.upd[(size_t)i];
      if (u.a == E.tgt || u.r == E.tgt || u.u == E.tgt) throw std::invalid_argument("VM entry reads its own target");
      ru[i] = std::max({base + 1, fin[u.a] + 1, fin[u.r] + 1, fin[u.u] + 1});      // operands must be final
    }
    if (E.keep_order) {
      for (int i = 1; i < n; i++) ru[i] = std::max(ru[i], ru[i - 1]);   // chunks keep the given order
    } else {
      std::vector<int> perm((size_t)n);
      std::iota(perm.begin(), perm.end(), 0);
      std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return ru[x] < ru[y]; });
      std::vector<VmUpd> u2((size_t)n);
      std::vector<int> r2((size_t)n);
      for (int i = 0; i < n; i++) { u2[i] = E.upd[(size_t)perm[i]]; r2[i] = ru[(size_t)perm[i]]; }
      E.upd.swap(u2);
      ru.swap(r2);
    }
    const int last = n ? ru[n - 1] : base;
    int rfinal = last;                      // round in which the entry becomes final
    if (E.mulr >= 0) rfinal = std::max({last, fin[E.mulr] + 1, base + 1});
    if (n == 0) rfinal = std::max(rfinal, base + 1);
    // eager chunks: maximal runs of equal readiness round
    struct Chunk { int round, first, count; };
    std::vector<Chunk> chunks;
    for (int i = 0; i < n;) {
      int j = i;
      while (j < n && ru[j] == ru[i]) j++;
      chunks.push_back({ru[i], i, j - i});
      i = j;
    }
    // smooth: an entry that will not be final for many rounds (a tail-block cell of the LU collects ~100 updates from
    // the head pivots) need not take each burst of updates in the round it becomes possible — one lane would walk ten
    // records while the other waves wait at the barrier with one or two.  Bursts are paid out at a steady rate over the
    // rounds up to the entry's last burst, the backlog carried forward: same updates, same order, same arithmetic, and
    // nothing is final later than before (the entry's last chunk stays where it was).
    {
      static const int smooth = diag_env("MISTRA_DIAG_SMOOTH") ? std::atoi(diag_env("MISTRA_DIAG_SMOOTH")) : 1;
      int last_burst = -1;
      for (size_t k = 0; k + 1 < chunks.size(); k++)
        if (chunks[k].count >= 2 * upr) last_burst = (int)k;
      if (smooth && E.keep_order && last_burst >= 1) {
        int total = 0;
        for (int k = 0; k <= last_burst; k++) total += chunks[(size_t)k].count;
        const int r0 = chunks[0].round, r1 = chunks[(size_t)last_burst].round;
        int per = (total + (r1 - r0)) / (r1 - r0 + 1);               // updates per round over [r0, r1]
        per = std::max(2 * upr, (per + upr - 1) / upr * upr);
        std::vector<Chunk> out;
        int next_first = 0, k = 0, avail = 0;                        // avail: updates ready but not yet paid out
        for (int r = r0; r <= r1; r++) {
          while (k <= last_burst && chunks[(size_t)k].round <= r) avail += chunks[(size_t)k++].count;
          const int pay = r == r1 ? avail : std::min(avail, per);
          if (pay > 0) {
            out.push_back({r, next_first, pay});
            next_first += pay;
            avail -= pay;
          }
        }
        for (size_t kk = (size_t)last_burst + 1; kk < chunks.size(); kk++) out.push_back(chunks[kk]);
        chunks.swap(out);
      }
    }
    // merge forward: a chunk may run later, together with the entry's next chunk (same order, fewer headers).
    // The entry's last chunk decides when the entry is final, so it only grows up to one record.
    std::vector<Chunk> merged;
    for (int k = (int)chunks.size() - 1; k >= 0; k--) {
      if (!merged.empty()) {
        Chunk& cur = merged.back();
        bool cur_is_final = merged.size() == 1;
        int limit = cur_is_final ? upr : std::max(merge_budget, upr);
        if (cur.count + chunks[(size_t)k].count <= limit) {
          cur.first = chunks[(size_t)k].first;
          cur.count += chunks[(size_t)k].count;
          continue;
        }
      }
      merged.push_back(chunks[(size_t)k]);
    }
    bool final_placed = false;
    for (int k = (int)merged.size() - 1; k >= 0; k--) {
      const Chunk& c = merged[(size_t)k];
      bool fin_here = (k == 0) && (rfinal == c.round);
      rounds[c.round].push_back({(int)e, c.first, c.count, fin_here});
      final_placed |= fin_here;
    }
    if (!final_placed && has_final_op) rounds[rfinal].push_back({(int)e, n, 0, true});
    const int f = has_final_op ? rfinal : last;
    fin[E.tgt] = f;
    if (E.rcp >= 0) fin[E.rcp] = f;
    phase_max = std::max(phase_max, f);
  }

  P.nrounds = (int)rounds.size();
  P.blk_n.assign((size_t)P.nrounds * P.nw, 0);
  // Local rounds.  A round of a few dozen records costs a workgroup what every round costs it — all waves through a barrier and its LDS
  // round trips, ~650 cycles — for one or two rows of work.  Rounds of at most local_max records are walked by wave 0 ALONE, and a run of
  // consecutive ones without any barrier in between: the LDS operations of one wave complete in order, so a round sees what the round
  // before it stored.  Only the run's last round ends in the barrier, which the other waves meet with a null row (they have no rows at
  // all in the rounds before it).  Same records, same order per entry: the arithmetic does not know who walks it.
  std::vector<char> local((size_t)P.nrounds, 0), local_open((size_t)P.nrounds, 0);      // local_open: local AND followed by another local round
  if (local_max > 0) {      // (a workgroup of ONE wave — gas — has nobody to wait for: all of its rounds are local, whatever they hold)
    int r = 0;
    for (auto& kv : rounds) {
      long recs = 0;
      for (const Item& it : kv.second) recs += std::max(1, (it.count + upr - 1) / upr);
      local[(size_t)r++] = P.nw == 1 || recs <= local_max;
    }
    for (int q = 0; q + 1 < P.nrounds; q++) local_open[(size_t)q] = local[(size_t)q] && local[(size_t)q + 1];
  }
  // per-wave linear record streams: stream[w] = rows of 64 records of VM_REC_WORDS words
  std::vector<std::vector<uint32_t>> stream((size_t)P.nw);
  const uint32_t zoff = vm_off(zero_slot, 0);
  const uint32_t one_off = vm_off(lay.one(), 0), trash_off = vm_off(lay.trash(), 0);
  // (Measured in round 4: giving every idle lane a store target of its own instead of the one trash cell changes nothing — 25 49x
  //  timesteps/s either way on one box; stores of many lanes to one address do not cost the VM what a bank conflict would.)
  auto idle_record = [&](std::vector<uint32_t>& out, uint32_t row_flags) {
    out.push_back(trash_off);
    out.push_back(one_off | row_flags);
    for (int q = 2; q < VM_REC_WORDS; q++) out.push_back(zoff);
  };
  int ridx = 0;
  for (auto& kv : rounds) {
    std::vector<Item>& items = kv.second;
    auto nrec = [upr](const Item& it) { return std::max(1, (it.count + upr - 1) / upr); };
    std::stable_sort(items.begin(), items.end(), [&](const Item& a, const Item& b) { return nrec(a) > nrec(b); });
    std::vector<int> row0;      // first record row of every item within its lane
    static const bool lpt_everywhere = diag_env("MISTRA_DIAG_LPT_DEAL") != nullptr;      // A/B diagnostic
    const bool snake = nt / 64 >= 8 && !local[(size_t)ridx] && !lpt_everywhere;
    std::vector<int> lane = snake ? deal_snake(items, entries, nt, upr, row0) : deal(items, entries, nt, upr, row0, local[(size_t)ridx] ? 1 : 0);
    // The record that publishes a pivot's reciprocal carries an IEEE division (~150 cycles of dependent work) on top of its
    // row; every other wave then waits for it at the barrier.  Where the waves of a round do not all have the same number
    // of rows, that record goes to a wave with fewer: it swaps lanes with an equally long item there (no lane's row count
    // changes), and the division runs while the fuller waves walk their extra row.
    {
      const bool no_rcp_move = !(snake || diag_env("MISTRA_DIAG_PLAIN_DEAL") != nullptr) || diag_env("MISTRA_DIAG_NO_RCP_MOVE") != nullptr;     // (the longest-first deal places these items itself; the move belongs to the snake deal)
      std::vector<int> lane_rows((size_t)nt, 0);
      for (size_t k = 0; k < items.size(); k++) lane_rows[(size_t)lane[k]] += nrec(items[k]);
      const int nw = nt / 64;
      std::vector<int> wave_rows((size_t)nw, 0);
      for (int t = 0; t < nt; t++) wave_rows[(size_t)(t / 64)] = std::max(wave_rows[(size_t)(t / 64)], lane_rows[(size_t)t]);
      for (size_t k = 0; k < items.size() && !no_rcp_move; k++) {
        const Item& it = items[k];
        if (!(it.final && entries[(size_t)it.entry].rcp >= 0)) continue;
        const int w_from = lane[k] / 64;
        int w_to = w_from;
        for (int w = 0; w < nw; w++)
          if (wave_rows[(size_t)w] > 0 && wave_rows[(size_t)w] < wave_rows[(size_t)w_to]) w_to = w;
        if (w_to == w_from) continue;
        for (size_t j = 0; j < items.size(); j++)            // an equally long item in the emptier wave that publishes nothing
          if (lane[j] / 64 == w_to && nrec(items[j]) == nrec(it) && !(items[j].final && entries[(size_t)items[j].entry].rcp >= 0)) {
            std::swap(lane[k], lane[j]);
            std::swap(row0[k], row0[j]);
            break;
          }
      }
    }
    std::vector<std::vector<uint32_t>> prog((size_t)nt);     // VM_REC_WORDS words per record
    std::vector<size_t> by_row(items.size());                // a lane walks its items in the order of their first rows
    std::iota(by_row.begin(), by_row.end(), (size_t)0);
    std::stable_sort(by_row.begin(), by_row.end(), [&](size_t x, size_t y) { return row0[x] < row0[y]; });
    for (size_t kk = 0; kk < items.size(); kk++) {
      const size_t k = by_row[kk];
      const Item& it = items[k];
      const VmEntry& E = entries[(size_t)it.entry];
      std::vector<uint32_t>& w = prog[(size_t)lane[k]];
      const int nr = nrec(it);
      for (int r = 0; r < nr; r++) {
        const bool last_rec = r == nr - 1;
        uint32_t f1 = 0;
        int aux = lay.one();
        if (last_rec && it.final) {
          if (E.mulr >= 0) aux = E.mulr;
          if (E.rcp >= 0) {
            if (E.mulr >= 0) throw std::logic_error("an entry cannot both scale and publish a reciprocal");
            f1 = VM_D1_RCP;
            aux = E.rcp;
          }
        }
        w.push_back(vm_off(E.tgt, 0));
        w.push_back(vm_off(aux, f1 | (r > 0 ? VM_D1_CONT : 0u)));
        for (int u = 0; u < upr; u++) {
          int i = r * upr + u;
          if (i < it.count) {
            const VmUpd& up = E.upd[(size_t)(it.first + i)];
            w.push_back(vm_off(up.a, 0));
            if (upr == 2) w.push_back(vm_off(up.r, 0));
            else if (up.r != lay.one()) throw std::logic_error("a record of three updates has no middle operand: it must be the 1.0 cell");
            w.push_back(vm_off(up.u, 0));
          } else {
            for (int o = 0; o < (upr == 2 ? 3 : 2); o++) w.push_back(zoff);
          }
        }
      }
      P.n_updates += it.count;
      P.n_items++;
      P.n_records += nr;
    }
    int crit = 0;
    for (int wv = 0; wv < P.nw; wv++) {
      size_t n = 0;
      for (int l = 0; l < 64; l++) n = std::max(n, prog[(size_t)wv * 64 + l].size() / VM_REC_WORDS);
      if (n > 0xFFFF) throw std::logic_error("VM block too long");
      if (local_open[(size_t)ridx] && wv > 0) {          // wave 0's own round, and its next one too: the others are not here
        if (n != 0) throw std::logic_error("a local round has records outside wave 0");
        continue;
      }
      const size_t rows = std::max<size_t>(n, 1);       // a wave with no work still gets a null row carrying the round mark
      P.blk_n[(size_t)ridx * P.nw + wv] = (uint16_t)rows;
      for (size_t r = 0; r < rows; r++) {
        uint32_t row_flags = (r == rows - 1 ? VM_ROW_EOR | (local_open[(size_t)ridx] ? VM_ROW_LOCAL : 0u) : 0u) | (n == 0 ? VM_ROW_NULL : 0u);
        for (int l = 0; l < 64; l++) {
          const auto& w = prog[(size_t)wv * 64 + l];
          if (r * VM_REC_WORDS < w.size() &&
              ((w[r * VM_REC_WORDS + 1] & VM_D1_RCP) || (w[r * VM_REC_WORDS + 1] & VM_AUX_MASK) != one_off))
            row_flags |= VM_ROW_AUX;      // the row needs its aux operand (reciprocal to publish, or a scale factor != 1.0)
        }
        for (int l = 0; l < 64; l++) {
          const auto& w = prog[(size_t)wv * 64 + l];
          if (r * VM_REC_WORDS < w.size()) {
            for (int q = 0; q < VM_REC_WORDS; q++) stream[(size_t)wv].push_back(w[r * VM_REC_WORDS + q] | (q == 1 ? row_flags : 0u));
          } else {
            idle_record(stream[(size_t)wv], row_flags);      // loads the 0.0 cell, stores nothing
          }
        }
      }
      P.wave_rows += (int64_t)rows;
      crit = std::max(crit, (int)rows);
    }
    P.crit_rows += crit;
    P.nbarriers += local_open[(size_t)ridx] ? 0 : 1;
    ridx++;
  }
  // every wave must meet exactly nbarriers barriers (a mismatch would hang the workgroup): counted on the streams as built
  for (int wv = 0; wv < P.nw; wv++) {
    int eor = 0;
    const std::vector<uint32_t>& st = stream[(size_t)wv];
    for (size_t row = 0; row < st.size() / (64 * VM_REC_WORDS); row++) {
      const uint32_t d1 = st[row * 64 * VM_REC_WORDS + 1];
      if ((d1 & VM_ROW_LOCAL) && (!(d1 & VM_ROW_EOR) || wv != 0)) throw std::logic_error("local mark outside the end of a round of wave 0");
      eor += (d1 & VM_ROW_EOR) && !(d1 & VM_ROW_LOCAL);
    }
    if (eor != P.nbarriers) throw std::logic_error("the waves of a VM program do not meet the same number of barriers");
  }
  // census: LDS-array cycles of the gathers as placed (per wave row: target read + write, six operand reads)
  for (int wv = 0; wv < P.nw; wv++) {
    const std::vector<uint32_t>& st = stream[(size_t)wv];
    for (size_t row = 0; row < st.size() / (64 * VM_REC_WORDS); row++) {
      if (st[row * 64 * VM_REC_WORDS + 1] & VM_ROW_NULL) continue;
      for (int q = 0; q < VM_REC_WORDS; q++) {
        if (q == 1) continue;
        int cell[64];
        for (int l = 0; l < 64; l++) cell[l] = (int)(st[(row * 64 + (size_t)l) * VM_REC_WORDS + (size_t)q] >> 3);
        P.lds_cycles += (q == 0 ? 2 : 1) * lds_pass_cycles(cell);
      }
    }
  }
  P.wave_base.assign((size_t)P.nw, 0);
  for (int wv = 0; wv < P.nw; wv++) {
    P.wave_base[(size_t)wv] = (uint32_t)(P.recs.size() / (64 * VM_REC_WORDS));
    // idle rows of slack so that the executor's look-ahead loads past the last record stay in bounds
    std::vector<uint32_t>& st = stream[(size_t)wv];
    for (int rr = 0; rr < VM_LOOKAHEAD_ROWS; rr++)
      for (int l = 0; l < 64; l++) idle_record(st, 0);
    // device layout: planar rows (vm_rec_index)
    const size_t rows = st.size() / (64 * VM_REC_WORDS), row0 = P.wave_base[(size_t)wv];
    P.recs.resize((row0 + rows) * 64 * VM_REC_WORDS);
    for (size_t r = 0; r < rows; r++)
      for (int l = 0; l < 64; l++)
        for (int k = 0; k < VM_REC_WORDS; k++) P.recs[vm_rec_index(row0 + r, l, k)] = st[(r * 64 + (size_t)l) * VM_REC_WORDS + (size_t)k];
  }
  return P;
}

// ---------------------------------------------------------------------------------------------------------------
// KppDecomp_x (gas.f:6142-6176), entry view: slot p=(k,c) of row k gets  - L(k,j)*U(j,c)  for every j < min(k,c) with
// both factors present, ascending j (the order in which the reference's kk/jj loops touch W(c)).  The multiplier is
// taken as W(k,j)*R(j) from the unscaled slot (see schedule.hpp); pivots publish R(k) = 1/U(k,k) when final; phase 1
// scales the L slots in place so that the solves find L(k,j) where the reference leaves it.  tail_h >= 0: phase 1 also
// scales the strictly-upper entries of the tail block by their ROW's pivot reciprocal, U'(i,c) = U(i,c)*R(i) for
// i >= tail_h: the tail chain's backward sweep then runs on a unit-diagonal triangle (schedule.hpp: TailSolve).
std::vector<VmEntry> lu_entries(const MechTables& m, const VmLayout& lay, bool with_rhs, int tail_h,
                                std::vector<std::pair<int, int>>* scale_pairs, int dense_h, int dense_jm) {
  const int n = m.nvar;
  const int dh = dense_h >= 0 ? dense_h : n, djm = dense_h >= 0 ? dense_jm : n;     // dense block [dh, n), Schur pivots [djm, dh)
  std::vector<VmEntry> out((size_t)m.nnz);
  std::vector<int> where((size_t)n, -1);    // column -> slot in the current row
  for (int k = 0; k < n; k++) {
    for (int p = m.crow[k]; p < m.crow[k + 1]; p++) where[(size_t)m.icol[p]] = p;
    for (int p = m.crow[k]; p < m.crow[k + 1]; p++) {
      VmEntry& E = out[(size_t)p];
      E.tgt = p;
      E.phase = 0;
      if (m.icol[p] == k && k < dh) E.rcp = lay.rdiag(k);      // the dense block's pivots are published by dense_lu
    }
    for (int pl = m.crow[k]; pl < m.diag[k]; pl++) {
      int j = m.icol[pl];
      for (int pu = m.diag[j] + 1; pu < m.crow[j + 1]; pu++) {
        int t = where[(size_t)m.icol[pu]];
        if (t < 0) throw std::logic_error("LU pattern is not closed under fill-in");
        if (k >= dh && m.icol[pu] >= dh && j >= djm) continue;   // a slot of the dense block, a pivot dense_lu applies
        out[(size_t)t].upd.push_back({pl, lay.rdiag(j), pu});
      }
    }
    for (int p = m.crow[k]; p < m.crow[k + 1]; p++) where[(size_t)m.icol[p]] = -1;
  }
  if (with_rhs)      // the right-hand side rides along: X(i) -= (W(i,j)*R(j)) * X(j), ascending j as the forward sweep does
    for (int i = 0; i < n; i++) {
      if (m.diag[i] == m.crow[i]) continue;
      VmEntry E;
      E.tgt = lay.xs(i);
      E.phase = 0;
      for (int p = m.crow[i]; p < m.diag[i]; p++)
        if (m.icol[p] < dh) E.upd.push_back({p, lay.rdiag(m.icol[p]), lay.xs(m.icol[p])});      // columns >= dh: the tail chain
      if (!E.upd.empty()) out.push_back(std::move(E));
    }
  auto scale = [&](int tgt, int aux) {
    if (scale_pairs) {
      scale_pairs->emplace_back(tgt, aux);
      return;
    }
    VmEntry E;
    E.tgt = tgt;
    E.phase = 1;
    E.mulr = aux;
    out.push_back(std::move(E));
  };
  for (int k = 0; k < n; k++)
    for (int p = m.crow[k]; p < m.diag[k]; p++)
      if (m.icol[p] < dh && !(k >= dh && m.icol[p] >= djm)) scale(p, lay.rdiag(m.icol[p]));     // (dense_lu's Schur steps scale theirs)
  if (tail_h >= 0)
    for (int k = tail_h; k < dh; k++)                        // (rows of the dense block: dense_lu stores them scaled)
      for (int p = m.diag[k] + 1; p < m.crow[k + 1]; p++) scale(p, lay.rdiag(k));
  return out;
}

// A long serial accumulation  x -= sum_k (a_k*r_k)*u_k  walked by ONE lane becomes the critical path of its rounds.
// Entries with more than `threshold` updates are cut into partial sums of ~sqrt(n) terms, each accumulated from 0 in
// its own temp cell (possibly by different lanes, in parallel), and combined:  x -= (T_p * 1.0) * (-1.0).
// Same terms, different association.  Returns the number of temp cells used from `first_temp` on.
int split_long_entries(std::vector<VmEntry>& entries, const VmLayout& lay, int threshold, int first_temp) {
  std::vector<VmEntry> out;
  int used = 0;
  for (VmEntry& E : entries) {
    const int n = (int)E.upd.size();
    if (n <= threshold) {
      out.push_back(std::move(E));
      continue;
    }
    int S = 2;
    while (S * S < n) S += 2;                 // even part length ~ sqrt(n): whole two-update records
    VmEntry comb;
    comb.tgt = E.tgt;
    comb.mulr = E.mulr;
    comb.rcp = E.rcp;
    comb.phase = E.phase;
    comb.keep_order = true;
    for (int lo = 0; lo < n; lo += S) {
      if (first_temp + used >= lay.max_temps) throw std::logic_error("out of VM temp cells (raise MAX_TEMPS of the mechanism)");
      VmEntry part;
      part.tgt = lay.temp(first_temp + used);
      used++;
      part.phase = E.phase;
      part.keep_order = E.keep_order;
      part.upd.assign(E.upd.begin() + lo, E.upd.begin() + std::min(n, lo + S));
      comb.upd.push_back({part.tgt, lay.one(), lay.minus_one()});
      out.push_back(std::move(part));
    }
    out.push_back(std::move(comb));
  }
  entries.swap(out);
  return used;
}

// KppSolve_x (gas.f:6206-6608) as one program: phase 0 forward sweep with unit L, phase 1 backward sweep.
// X lives behind Ghimj in VM memory.  (Kept for tests; the kernel runs the head/tail split below.)
std::vector<VmEntry> solve_entries(const MechTables& m, const VmLayout& lay) {
  const int n = m.nvar;
  std::vector<VmEntry> out;
  for (int i = 0; i < n; i++) {
    if (m.diag[i] == m.crow[i]) continue;
    VmEntry E;
    E.tgt = lay.xs(i);
    E.phase = 0;
    for (int p = m.crow[i]; p < m.diag[i]; p++) E.upd.push_back({p, lay.one(), lay.xs(m.icol[p])});
    out.push_back(std::move(E));
  }
  for (int i = n - 1; i >= 0; i--) {
    VmEntry E;
    E.tgt = lay.xs(i);
    E.phase = 1;
    E.keep_order = false;        // see schedule.hpp: readiness order instead of the reference's ascending columns
    E.mulr = lay.rdiag(i);
    for (int p = m.diag[i] + 1; p < m.crow[i + 1]; p++) E.upd.push_back({p, lay.one(), lay.xs(m.icol[p])});
    out.push_back(std::move(E));
  }
  return out;
}

// Forward sweep split at row h: head rows completely, tail rows only their head-column terms (the leading terms of the
// reference's ascending-column order); the tail chain then subtracts the tail-column terms, again ascending.
std::vector<VmEntry> solve_head_fwd_entries(const MechTables& m, const VmLayout& lay, int h) {
  const int n = m.nvar;
  std::vector<VmEntry> out;
  for (int i = 0; i < n; i++) {
    VmEntry E;
    E.tgt = lay.xs(i);
    E.phase = 0;
    for (int p = m.crow[i]; p < m.diag[i]; p++)
      if (m.icol[p] < h) E.upd.push_back({p, lay.one(), lay.xs(m.icol[p])});
    if (!E.upd.empty()) out.push_back(std::move(E));
  }
  return out;
}

// Backward sweep of the head rows, the tail part of X being final already.  A head row couples to up to 42 tail columns;
// walked by one lane those terms make the first round 17 records long in one wave while the others hold 1-2.  All of them
// are available when the program starts (they need only the tail's X), so rows with more than `split_over` tail terms
// first collect them as partial sums in temp cells (phase 0: one extra round, every part independent), and the row's own
// entry (phase 1) starts from those.  Same terms, different association — as in the forward sweep (split_long_entries).
// first_temp / temps_used: the temp cells taken (zeroed by the kernel before every solve).
std::vector<VmEntry> solve_head_bwd_entries(const MechTables& m, const VmLayout& lay, int h, int split_over, int first_temp,
                                            int* temps_used) {
  std::vector<VmEntry> parts, rows;
  int used = 0;
  for (int i = h - 1; i >= 0; i--) {
    VmEntry E;
    E.tgt = lay.xs(i);
    E.phase = 1;
    E.keep_order = false;
    E.mulr = lay.rdiag(i);
    std::vector<VmUpd> tail_terms;
    for (int p = m.diag[i] + 1; p < m.crow[i + 1]; p++) {
      const VmUpd u{p, lay.one(), lay.xs(m.icol[p])};
      (m.icol[p] >= h ? tail_terms : E.upd).push_back(u);
    }
    if (split_over > 0 && (int)tail_terms.size() > split_over) {
      const int n = (int)tail_terms.size();
      int S = 2;
      while (S * S < 2 * n) S += 2;            // part length ~ sqrt(2n), whole two-update records
      std::vector<VmUpd> combine;
      for (int lo = 0; lo < n; lo += S) {
        if (first_temp + used >= lay.max_temps) throw std::logic_error("out of VM temp cells (raise MAX_TEMPS of the mechanism)");
        VmEntry part;
        part.tgt = lay.temp(first_temp + used++);
        part.phase = 0;
        part.keep_order = false;
        part.upd.assign(tail_terms.begin() + lo, tail_terms.begin() + std::min(n, lo + S));
        combine.push_back({part.tgt, lay.one(), lay.minus_one()});
        parts.push_back(std::move(part));
      }
      E.upd.insert(E.upd.begin(), combine.begin(), combine.end());
    } else {
      E.upd.insert(E.upd.begin(), tail_terms.begin(), tail_terms.end());
    }
    rows.push_back(std::move(E));
  }
  if (temps_used) *temps_used = used;
  parts.insert(parts.end(), std::make_move_iterator(rows.begin()), std::make_move_iterator(rows.end()));
  return parts;
}

TailSolve build_tail_solve(const MechTables& m, const VmLayout& lay, int regs) {
  TailSolve T;
  const int n = m.nvar, zero_slot = lay.zero();
  T.regs = regs > 0 ? regs : n > 128 + 64 ? 2 : 1;           // 128-row tail where the mechanism is big enough to leave a head
  T.m = 64 * T.regs;
  if (T.m > n) throw std::invalid_argument("mechanism smaller than one wave");
  T.h = n - T.m;
  if (zero_slot > 0xFFFF) throw std::invalid_argument("tail tables need 16-bit Ghimj slots");
  const uint32_t z = (uint32_t)zero_slot;
  // + VM_LOOKAHEAD_ROWS groups of slack for the kernel's look-ahead loads
  T.fwd.assign(((size_t)T.m / 4 + VM_LOOKAHEAD_ROWS) * 256, z | (z << 16));
  T.bwd.assign(((size_t)T.m / 4 + VM_LOOKAHEAD_ROWS) * 256, z | (z << 16));
  auto put = [&](std::vector<uint32_t>& tab, int group_pos, int lane, int r, int slot) {
    uint32_t& w = tab[((size_t)(group_pos / 4) * 64 + lane) * 4 + group_pos % 4];
    w = r == 0 ? ((w & 0xFFFF0000u) | (uint32_t)slot) : ((w & 0x0000FFFFu) | ((uint32_t)slot << 16));
  };
  for (int ti = 0; ti < T.m; ti++) {
    const int i = T.h + ti, lane = ti % 64, r = ti / 64;
    for (int p = m.crow[i]; p < m.crow[i + 1]; p++) {
      const int c = m.icol[p];
      if (c < T.h || c == i) continue;
      const int q = c - T.h;
      if (c < i) put(T.fwd, q, lane, r, p);                 // L(i, c): used when the forward chain reaches column q
      else put(T.bwd, T.m - 1 - q, lane, r, p);             // U(i, c): used when the backward chain reaches column q
    }
  }
  return T;
}

// Row tables of the dense tail block (schedule.hpp: DenseTail).
DenseTail build_dense_tail(const MechTables& m, const VmLayout& lay, int nd, int kb) {
  DenseTail D;
  const int n = m.nvar;
  if (nd != 64) throw std::invalid_argument("the dense tail block is 64 rows (8 waves x 2 tiles of 16x16)");
  D.nd = nd;
  D.h = n - nd;
  D.kb = kb;
  D.jm = D.h - 4 * kb;
  if (D.jm < 0 || kb < 0) throw std::invalid_argument("more Schur steps than pivots");
  if (D.info_rows() > DenseTail::kInfoRowsMax) throw std::invalid_argument("too many Schur steps for the LDS row table");
  D.row_info.assign((size_t)DenseTail::kInfoRowsMax * 4, 0u);
  auto describe_range = [&](int g, int row, int c0, int c1) {        // slots of `row` with columns in [c0, c1)
    int first = -1, seen = 0;
    uint64_t absent = 0;
    for (int c = c0; c < c1; c++) {
      int slot = -1;
      for (int p = m.crow[row]; p < m.crow[row + 1]; p++)
        if (m.icol[p] == c) slot = p;
      if (slot < 0) { absent |= 1ull << (c - c0); continue; }
      if (first < 0) first = slot;
      if (slot != first + seen) throw std::logic_error("dense tail: a row's slots inside a column range are not contiguous");
      seen++;
    }
    for (int c = c1 - c0; c < 64; c++) absent |= 1ull << c;
    D.row_info[(size_t)g * 4] = (uint32_t)std::max(first, 0);
    D.row_info[(size_t)g * 4 + 1] = (uint32_t)absent;
    D.row_info[(size_t)g * 4 + 2] = (uint32_t)(absent >> 32);
  };
  for (int i = 0; i < nd; i++) {
    describe_range(i, D.h + i, D.h, n);
    // the solves' forward chain over the block (ros3_kernel.hip: dense_fwd_chain) reads a row's columns 1..63 as consecutive cells
    if ((((uint64_t)D.row_info[(size_t)i * 4 + 1] | ((uint64_t)D.row_info[(size_t)i * 4 + 2] << 32)) & ~1ull) != 0)
      throw std::logic_error("dense tail: a row of the block lacks a column other than its first");
    describe_range(64 + i, D.h + i, D.jm, D.h);
  }
  for (int r = 0; r < 4 * kb; r++) describe_range(128 + r, D.jm + r, D.h, n);
  // operand cells of the Schur steps in MFMA lane order (schedule.hpp)
  if (lay.zero() > 0xFFFF) throw std::invalid_argument("Schur cell table needs 16-bit M cells");
  D.schur_cells.assign((size_t)8 * kb * 64, (uint16_t)lay.zero());
  auto cell_or_zero = [&](int g, int c) { const int x = D.cell(g, c); return (uint16_t)(x < 0 ? lay.zero() : x); };
  for (int k = 0; k < kb; k++)
    for (int lane = 0; lane < 64; lane++) {
      const int lrow = lane >> 4, lcol = lane & 15;
      for (int I = 0; I < 4; I++) D.schur_cells[D.schur_w(I, k, lane)] = cell_or_zero(64 + 16 * I + lcol, 4 * k + lrow);
      for (int half = 0; half < 2; half++)
        for (int t = 0; t < 2; t++) D.schur_cells[D.schur_u(half, k, t, lane)] = cell_or_zero(128 + 4 * k + lrow, 16 * (2 * half + t) + lcol);
    }
  return D;
}

// ---------------------------------------------------------------------------------------------------------------
GsumProgram build_gsum_program(const std::vector<std::vector<std::pair<int, double>>>& outputs,
                               const std::vector<int>& slot_of_output, int nq, int nt, uint32_t src_base_bytes,
                               uint32_t zero_cell_bytes) {
  GsumProgram P;
  P.nt = nt;
  P.nw = nt / 64;
  P.nq = nq;
  std::vector<int> output_of_slot((size_t)nq * nt, -1);
  for (size_t o = 0; o < outputs.size(); o++) {
    int s = slot_of_output[o];
    if (s < 0 || s >= nq * nt || output_of_slot[(size_t)s] >= 0) throw std::logic_error("bad gsum slot map");
    output_of_slot[(size_t)s] = (int)o;
  }
  P.wave_base.assign((size_t)P.nw, 0);
  P.rows.assign((size_t)P.nw, 0);
  const float neg_zero = -0.0f;
  uint32_t neg_zero_bits;
  static_assert(sizeof neg_zero_bits == sizeof neg_zero, "float is 32 bits");
  std::memcpy(&neg_zero_bits, &neg_zero, 4);
  auto pad_row = [&]() {
    for (int l = 0; l < 64; l++) {
      for (int k = 0; k < 4; k++) P.recs.push_back(zero_cell_bytes);
      for (int k = 0; k < 4; k++) P.recs.push_back(neg_zero_bits);
    }
  };
  if (zero_cell_bytes & 7u) throw std::logic_error("gsum addresses must leave their low bits to the row marks");
  for (int w = 0; w < P.nw; w++) {
    P.wave_base[(size_t)w] = (uint32_t)(P.recs.size() / 512);
    size_t wave_rows = 0;
    for (int q = 0; q < nq; q++) {
      size_t n = 0;
      for (int l = 0; l < 64; l++) {
        int o = output_of_slot[(size_t)q * nt + w * 64 + l];
        if (o >= 0) n = std::max(n, outputs[(size_t)o].size());
      }
      const size_t rows = std::max<size_t>(1, (n + 3) / 4);       // at least one row: it carries the block's flush mark
      for (size_t r = 0; r < rows; r++)
        for (int l = 0; l < 64; l++) {
          int o = output_of_slot[(size_t)q * nt + w * 64 + l];
          uint32_t addr[4], cbits[4];
          for (size_t k = 0; k < 4; k++) {
            size_t s = r * 4 + k;
            if (o >= 0 && s < outputs[(size_t)o].size()) {
              const auto& term = outputs[(size_t)o][s];
              float cf = (float)term.second;
              if ((double)cf != term.second) throw std::logic_error("stoichiometric coefficient is not a float32 value");
              if (term.first < 0 || term.first > 0xFFFF) throw std::logic_error("gsum source index out of range");
              addr[k] = src_base_bytes + 8u * (uint32_t)term.first;
              std::memcpy(&cbits[k], &cf, 4);
              P.n_terms++;
            } else {            // exact identity: acc + (-0.0f * 0.0) = acc
              addr[k] = zero_cell_bytes;
              cbits[k] = neg_zero_bits;
            }
          }
          if (r == rows - 1) addr[0] |= GS_ROW_FLUSH;      // every lane of the row: the output is complete after this row
          for (int k = 0; k < 4; k++) P.recs.push_back(addr[k]);
          for (int k = 0; k < 4; k++) P.recs.push_back(cbits[k]);
        }
      wave_rows += rows;
    }
    while (wave_rows % GS_ROW_ALIGN) { pad_row(); wave_rows++; }       // the executor walks whole turns of its ring
    if (wave_rows > 0xFFFF) throw std::logic_error("gsum stream too long");
    P.rows[(size_t)w] = (uint16_t)wave_rows;
    P.wave_rows += (int64_t)wave_rows;
    for (int rr = 0; rr < VM_LOOKAHEAD_ROWS; rr++) pad_row();     // slack for the executor's look-ahead loads
  }
  return P;
}

ScaleProgram build_scale_program(const std::vector<std::pair<int, int>>& pairs, const VmLayout& lay, int nt) {
  ScaleProgram P;
  P.nt = nt;
  P.nw = nt / 64;
  P.n_pairs = (int64_t)pairs.size();
  const size_t per_lane = (pairs.size() + (size_t)nt - 1) / (size_t)nt;
  P.nslots = (int)(((per_lane + 1) / 2 + 7) / 8 * 8);
  const uint32_t trash = 8u * (uint32_t)lay.trash(), one = 8u * (uint32_t)lay.one();
  const size_t wave_slots = (size_t)P.nslots + VM_LOOKAHEAD_ROWS;
  P.recs.assign((size_t)P.nw * wave_slots * 64 * 4, 0u);
  for (size_t i = 0; i < P.recs.size(); i += 2) { P.recs[i] = trash; P.recs[i + 1] = one; }      // idle: trash *= 1.0
  for (size_t i = 0; i < pairs.size(); i++) {        // pair i -> thread i % nt, its (i / nt)-th pair
    const int t = (int)(i % (size_t)nt), w = t / 64, l = t % 64;
    const size_t j = i / (size_t)nt, at = (((size_t)w * wave_slots + j / 2) * 64 + (size_t)l) * 4 + 2 * (j % 2);
    P.recs[at] = 8u * (uint32_t)pairs[i].first;
    P.recs[at + 1] = 8u * (uint32_t)pairs[i].second;
  }
  return P;
}

// ---------------------------------------------------------------------------------------------------------------
KernelSchedule build_kernel_schedule(const MechTables& m, int nt, uint32_t ab_base_bytes, int max_temps, int dense_nd, int dense_kb,
                                     uint32_t jb_base_bytes) {
  if (nt % 64 != 0 || nt <= 0 || nt > 1024) throw std::invalid_argument("nt must be a multiple of 64 in (0,1024]");
  if (VmLayout{m.nnz, m.nvar, max_temps}.size() * 8 > 160 * 1024) throw std::invalid_argument("mechanism too large for the LDS VM");
  if (m.nx() > 0xFFFF || m.nb >= 0xFFFF || m.nreact > 0xFFFF) throw std::invalid_argument("mechanism too large");
  KernelSchedule S;
  S.nt = nt;
  S.nw = nt / 64;
  S.ab_base_bytes = ab_base_bytes;
  S.jb_base_bytes = jb_base_bytes ? jb_base_bytes : ab_base_bytes;
  const VmLayout lay{m.nnz, m.nvar, max_temps};                         // M = [Ghimj | XS | 0.0 | 1.0 | trash | -1.0 | R | temps]
  const uint32_t zero_cell_bytes = 8u * (uint32_t)lay.zero();
  S.spt = ceil_div(m.nvar, nt);
  S.rpt = ceil_div(m.nreact, nt);
  const uint64_t one = (uint64_t)(m.nvar + m.nfix);       // X slot of the constant 1.0

  // ---- Fun_x products
  const uint64_t ab_trash = (uint64_t)std::max(m.nreact, m.nb);     // spare cells behind the B products (and the A products where they share the array), one per lane (LdsLayout::AB_TRASH)
  // where Jac_SP's B products have an array of their own (jb_base_bytes: inside the Ghimj area), the A array holds NREACT products and its
  // spare cells come right behind them (LdsLayout::A_CELLS)
  const uint64_t a_trash = jb_base_bytes ? (uint64_t)m.nreact : ab_trash;
  const int spare = nt >= 128 ? 64 : 32;      // ros3_kernel.hpp: spare_cells
  S.a_trash = (int)a_trash;
  S.fun_fac.resize((size_t)S.rpt * nt);
  for (size_t i = 0; i < S.fun_fac.size(); i++) S.fun_fac[i] = one | (one << 16) | (one << 32) | ((a_trash + (i % (size_t)nt) % (size_t)spare) << 48);
  for (int r = 0; r < m.nreact; r++) {
    uint64_t f[3] = {one, one, one};
    int nf = m.a_ptr[r + 1] - m.a_ptr[r];
    if (nf > 3) throw std::logic_error("reaction with more than 3 factors");
    for (int i = 0; i < nf; i++) f[i] = (uint64_t)m.a_fac[(size_t)m.a_ptr[r] + i];
    S.fun_fac[(size_t)r] = f[0] | (f[1] << 16) | (f[2] << 32) | ((uint64_t)r << 48);
  }
  {
    std::vector<std::vector<std::pair<int, double>>> outs((size_t)m.nvar);
    std::vector<int> slot((size_t)m.nvar);
    for (int s = 0; s < m.nvar; s++) {
      slot[(size_t)s] = s;
      for (int p = m.vd_ptr[s]; p < m.vd_ptr[s + 1]; p++) outs[(size_t)s].emplace_back(m.vd_idx[(size_t)p], m.vd_coef[(size_t)p]);
    }
    S.vdot = build_gsum_program(outs, slot, S.spt, nt, ab_base_bytes, zero_cell_bytes);
  }

  // ---- Jac_SP_x products, under the owning reaction
  S.jac_fac.resize((size_t)S.rpt * 3 * nt);
  for (size_t i = 0; i < S.jac_fac.size(); i++) S.jac_fac[i] = one | (one << 16) | (one << 32) | ((ab_trash + (i % (size_t)nt) % (size_t)spare) << 48);
  {
    std::vector<int> used((size_t)m.nreact, 0);
    for (int b = 0; b < m.nb; b++) {
      int r = m.b_rct[(size_t)b];
      int k = used[(size_t)r]++;
      if (k >= 3) throw std::logic_error("more than 3 Jacobian products for one reaction");
      uint64_t f[3] = {one, one, one};
      int nf = m.b_ptr[b + 1] - m.b_ptr[b];
      if (nf > 3) throw std::logic_error("Jacobian product with more than 3 factors");
      for (int i = 0; i < nf; i++) f[i] = (uint64_t)m.b_fac[(size_t)m.b_ptr[b] + i];
      int q = r / nt, t = r % nt;
      S.jac_fac[((size_t)q * 3 + k) * nt + t] = f[0] | (f[1] << 16) | (f[2] << 32) | ((uint64_t)b << 48);
    }
  }
  // ---- JVS construction: structurally non-zero slots become register-resident outputs, balanced over threads
  {
    std::vector<int> nz, zero;
    for (int k = 0; k < m.nnz; k++) (m.jv_ptr[k + 1] > m.jv_ptr[k] ? nz : zero).push_back(k);
    S.n_jnz = (int)nz.size();
    S.n_jzero = (int)zero.size();
    S.jpt = std::max(1, ceil_div(S.n_jnz, nt));
    S.zpt = std::max(1, ceil_div(S.n_jzero, nt));
    std::vector<char> is_diag((size_t)m.nnz, 0);
    for (int s = 0; s < m.nvar; s++) is_diag[(size_t)m.diag[(size_t)s]] = 1;
    std::stable_sort(nz.begin(), nz.end(), [&](int a, int b) {
      return m.jv_ptr[a + 1] - m.jv_ptr[a] > m.jv_ptr[b + 1] - m.jv_ptr[b];
    });
    std::vector<std::vector<std::pair<int, double>>> outs(nz.size());
    std::vector<int> slot(nz.size());
    S.jvs_pos.assign((size_t)S.jpt * nt, POS_NONE);
    for (size_t i = 0; i < nz.size(); i++) {
      int k = nz[i];
      int pass = (int)(i / (size_t)nt), pos = (int)(i % (size_t)nt);
      int t = (pass & 1) ? nt - 1 - pos : pos;
      slot[i] = pass * nt + t;
      for (int p = m.jv_ptr[k]; p < m.jv_ptr[k + 1]; p++) outs[i].emplace_back(m.jv_idx[(size_t)p], m.jv_coef[(size_t)p]);
      S.jvs_pos[(size_t)slot[i]] = (uint16_t)(k | (is_diag[(size_t)k] ? POS_DIAG : 0));
    }
    S.jvs = build_gsum_program(outs, slot, S.jpt, nt, S.jb_base_bytes, zero_cell_bytes);
    S.zero_pos.assign((size_t)S.zpt * nt, POS_NONE);
    // diagonals among the fill-in slots (species whose rate of change does not depend on themselves) first: they land in every
    // thread's slot 0, the only one the kernel treats as a possible diagonal (ros3_kernel.hip: prepare)
    std::stable_sort(zero.begin(), zero.end(), [&](int x, int y) { return is_diag[(size_t)x] > is_diag[(size_t)y]; });
    for (size_t i = 0; i < zero.size(); i++) {
      if (is_diag[(size_t)zero[i]] && i >= (size_t)nt) throw std::logic_error("more fill-in diagonals than threads");
      S.zero_pos[i] = (uint16_t)(zero[i] | (is_diag[(size_t)zero[i]] ? POS_DIAG : 0));
    }
  }
  S.diag_pos.assign((size_t)S.spt * nt, POS_NONE);
  for (int s = 0; s < m.nvar; s++) S.diag_pos[(size_t)s] = (uint16_t)m.diag[(size_t)s];

  if (dense_nd > 0) {
    if (nt < 512) throw std::invalid_argument("the dense tail block needs eight waves");
    S.dense = build_dense_tail(m, lay, dense_nd, dense_kb);
  }
  S.tail = build_tail_solve(m, lay);      // (the tail chain of the solves may reach further up than the dense block)
  // the Schur steps multiply UNSCALED L slots with ROW-SCALED U slots: their pivots must be rows of the solves' tail chain
  if (dense_nd > 0 && S.tail.h > S.dense.jm) throw std::logic_error("the solves' tail chain must cover the dense block and its Schur pivots");
  const int dh = dense_nd > 0 ? S.dense.h : -1, djm = dense_nd > 0 ? S.dense.jm : -1;
  // rounds small enough for wave 0 alone (build_vm_program: local rounds), in records
  // (measured, same box, tools/ab_envs.sh: tot sweeps 0 / 128 / 192 / 512 records: 25 500 / 25 720 / 25 670 / 25 420 timesteps/s, tot LU 0 / 128 / 320:
  //  25 700 / 25 670 / 25 410; aer LU 0 / 64 / 128 / 256: 61 790 / 62 040 / 61 700 / 60 730, aer sweeps 0 / 192: 61 500 / 61 700 — a round walked by one
  //  wave is hardly cheaper than one walked by all: what a round costs is its chain of LDS round trips, not its barrier)
  static const int local_lu = diag_env("MISTRA_DIAG_LOCAL_LU") ? std::atoi(diag_env("MISTRA_DIAG_LOCAL_LU")) : 64;
  static const int local_sweep = diag_env("MISTRA_DIAG_LOCAL_SWEEP") ? std::atoi(diag_env("MISTRA_DIAG_LOCAL_SWEEP")) : 128;
  {
    // the scaling gets its own pass where there is enough of it (tot: 22 cells per thread, +2.7 %); for the small
    // mechanisms the extra pass's start-up costs more than the VM rows it replaces (gas -1.8 %, aer -0.6 %, measured)
    std::vector<std::pair<int, int>> pairs;
    (void)lu_entries(m, lay, true, S.tail.h, &pairs, dh, djm);
    if ((int)pairs.size() >= 16 * nt || dense_nd > 0) {     // (dense_lu reads scaled multipliers: always after a scaling pass)
      S.lu = build_vm_program(lu_entries(m, lay, true, S.tail.h, &pairs, dh, djm), lay, nt, 2, VM_UPD_PER_REC, local_lu);
      pairs.resize(pairs.size() / 2);         // lu_entries appended the same list a second time
      S.lu_scale = build_scale_program(pairs, lay, nt);
    } else {
      S.lu = build_vm_program(lu_entries(m, lay, true, S.tail.h, nullptr), lay, nt, 2, VM_UPD_PER_REC, local_lu);
    }
  }
  S.solve = build_vm_program(solve_entries(m, lay), lay, nt);
  {
    std::vector<VmEntry> fwd = solve_head_fwd_entries(m, lay, S.tail.h);
    // forward: long head-column dot products of the tail rows cut into partial sums (tail rows carry up to 42 terms)
    static const int fwd_split = diag_env("MISTRA_DIAG_FWD_SPLIT") ? std::atoi(diag_env("MISTRA_DIAG_FWD_SPLIT")) : 6;
    S.n_temps = split_long_entries(fwd, lay, fwd_split, 0);
    static const int bwd_split = diag_env("MISTRA_DIAG_BWD_SPLIT") ? std::atoi(diag_env("MISTRA_DIAG_BWD_SPLIT")) : 4;
    int bwd_temps = 0;
    std::vector<VmEntry> bwd = solve_head_bwd_entries(m, lay, S.tail.h, bwd_split, S.n_temps, &bwd_temps);
    S.n_temps += bwd_temps;
    // the sweeps' updates are (L(i,j), 1.0, X(j)): the middle operand is dropped and a record holds three of them
    S.solve_head_fwd = build_vm_program(std::move(fwd), lay, nt, 2, VM_SWEEP_UPD_PER_REC, local_sweep);
    S.solve_head_bwd = build_vm_program(std::move(bwd), lay, nt, 2, VM_SWEEP_UPD_PER_REC, local_sweep);
  }
  return S;
}

std::string describe(const KernelSchedule& s) {
  char buf[2048];
  std::snprintf(buf, sizeof buf,
                "nt=%d spt=%d rpt=%d jpt=%d zpt=%d | vdot: %lld terms, %lld wave-rows | jvs: %lld terms, %lld wave-rows | "
                "LU: %d rounds, %lld updates in %lld items / %lld records, %lld wave-rows, critical %lld, %lld LDS cycles; dense tail block %d rows, %d rank-4 Schur steps | "
                "solve: tail %d rows in registers of one wave; head fwd %d rounds / %lld rows critical, head bwd %d rounds / %lld rows "
                "critical, %d partial-sum cells (whole solve as one VM program: %d rounds, %lld updates, %lld records, critical %lld)",
                s.nt, s.spt, s.rpt, s.jpt, s.zpt, (long long)s.vdot.n_terms, (long long)s.vdot.wave_rows,
                (long long)s.jvs.n_terms, (long long)s.jvs.wave_rows, s.lu.nrounds, (long long)s.lu.n_updates,
                (long long)s.lu.n_items, (long long)s.lu.n_records, (long long)s.lu.wave_rows, (long long)s.lu.crit_rows, (long long)s.lu.lds_cycles, s.dense.nd, s.dense.kb,
                s.tail.m, s.solve_head_fwd.nrounds, (long long)s.solve_head_fwd.crit_rows, s.solve_head_bwd.nrounds,
                (long long)s.solve_head_bwd.crit_rows, s.n_temps, s.solve.nrounds, (long long)s.solve.n_updates, (long long)s.solve.n_records,
                (long long)s.solve.crit_rows);
  return buf;
}

}  // namespace mistra
