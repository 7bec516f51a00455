// Micro-benchmarks for a DPP-based triangular-solve chain (diagnostics only, not product code): v_fmac_f64 with the
// row_newbcast DPP modifier (a lane's value to its 16-lane row in the same instruction), the gfx950 permlane swaps.
//   hipcc --offload-arch=gfx950 -O2 -o dpp dpp.hip && ./dpp          (on the GPU box; '-DNOPS=""' drops the wait states of the timing loops)
// Measured on MI355X: a dependent step 10.1 cycles with one wait state in front of the DPP read (results bit-identical to the
// host's fma chain), 14.1 with two, WRONG results with none; each further off-chain fmac ~4.3 cycles; a lane row's double to
// all four rows by v_permlane16_swap + v_permlane32_swap ~70 cycles, by 2 x ds_bpermute_b32 ~66.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#ifndef NOPS
#define NOPS "s_nop 1\n\t"
#endif
#define STEP(J) NOPS "v_fmac_f64_dpp %[x], %[x], %[l] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define STEP16 STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#define XSTEP(J, Y) NOPS "v_fmac_f64_dpp %[" #Y "], %[x], %[l] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"

// dependent chain, EXTRA further fmacs per step that read the chain value but feed nothing on the chain
template <int EXTRA>
__global__ void k_dpp_chain(double* out, int iters, long long* cyc) {
  int l = threadIdx.x;
  double x = 1.0 + l * 1e-6, m = 1e-9 * (l & 15), y0 = 0, y1 = 0, y2 = 0, y3 = 0;
  long long t0 = clock64();
  for (int i = 0; i < iters; i += 16) {
#define S(J)                                                              \
    asm volatile(STEP(J) : [x] "+v"(x) : [l] "v"(m));                     \
    if (EXTRA > 0) asm volatile(XSTEP(J, y) : [y] "+v"(y0) : [x] "v"(x), [l] "v"(m)); \
    if (EXTRA > 1) asm volatile(XSTEP(J, y) : [y] "+v"(y1) : [x] "v"(x), [l] "v"(m)); \
    if (EXTRA > 2) asm volatile(XSTEP(J, y) : [y] "+v"(y2) : [x] "v"(x), [l] "v"(m)); \
    if (EXTRA > 3) asm volatile(XSTEP(J, y) : [y] "+v"(y3) : [x] "v"(x), [l] "v"(m));
    S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + l] = x + y0 + y1 + y2 + y3;
  if (l == 0) cyc[0] = t1 - t0;
}

// plain (no DPP) dependent fmac chain for comparison, and the chain with an s_nop between
__global__ void k_fmac_chain(double* out, int iters, long long* cyc) {
  int l = threadIdx.x;
  double x = 1.0 + l * 1e-6, m = 1e-9, c = 1.0000001;
  long long t0 = clock64();
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; k++) asm volatile("v_fmac_f64 %[x], %[c], %[l]" : [x] "+v"(x) : [l] "v"(m), [c] "v"(c));
  }
  long long t1 = clock64();
  out[l] = x;
  if (l == 0) cyc[0] = t1 - t0;
}

// one lane-row's 16 values to all four lane-rows: v_permlane16_swap + v_permlane32_swap on both halves of a double
template <int ROW>
__device__ __forceinline__ double row_to_all(double v) {
  unsigned lo = (unsigned)__builtin_bit_cast(unsigned long long, v), hi = (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32);
  // permlane16_swap(a, b): a.row1 <-> b.row0, a.row3 <-> b.row2.   From a = b = v: a = [v0 v0 v2 v2], b = [v1 v1 v3 v3]
  auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  unsigned a = l16[ROW & 1], b = h16[ROW & 1];
  // permlane32_swap(a, b): a.upper32 <-> b.lower32.   From a = b: a = [lower lower], b = [upper upper]
  auto l32 = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  auto h32 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
  unsigned rl = l32[(ROW >> 1) & 1], rh = h32[(ROW >> 1) & 1];
  return __builtin_bit_cast(double, (unsigned long long)rl | ((unsigned long long)rh << 32));
}
template <int ROW>
__global__ void k_row_to_all(double* out) { out[threadIdx.x] = row_to_all<ROW>(100.0 + threadIdx.x); }

__global__ void k_bcast_chain(double* out, int iters, long long* cyc) {     // fmac_dpp -> row_to_all -> fmac_dpp ...
  int l = threadIdx.x;
  double x = 1.0 + l * 1e-6, m = 1e-9;
  long long t0 = clock64();
  for (int i = 0; i < iters; i += 4) {
    asm volatile(STEP(3) : [x] "+v"(x) : [l] "v"(m)); x = row_to_all<0>(x);
    asm volatile(STEP(5) : [x] "+v"(x) : [l] "v"(m)); x = row_to_all<1>(x);
    asm volatile(STEP(7) : [x] "+v"(x) : [l] "v"(m)); x = row_to_all<2>(x);
    asm volatile(STEP(9) : [x] "+v"(x) : [l] "v"(m)); x = row_to_all<3>(x);
  }
  long long t1 = clock64();
  out[l] = x;
  if (l == 0) cyc[0] = t1 - t0;
}
__global__ void k_bperm_chain(double* out, int iters, long long* cyc) {     // fmac_dpp -> ds_bpermute x2 -> ...
  int l = threadIdx.x;
  double x = 1.0 + l * 1e-6, m = 1e-9;
  int src = (l & 15) * 4;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    asm volatile(STEP(3) : [x] "+v"(x) : [l] "v"(m));
    unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    unsigned lo = __builtin_amdgcn_ds_bpermute(src, (int)(unsigned)u), hi = __builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(u >> 32));
    x = __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
  }
  long long t1 = clock64();
  out[l] = x;
  if (l == 0) cyc[0] = t1 - t0;
}
// semantics check of the DPP fmac: x(lane) += x(lane J of the same 16-lane row) * m(lane)
__global__ void k_dpp_sem(double* out) {
  int l = threadIdx.x;
  double x = l, m = 1000.0;
  asm volatile(STEP(5) : [x] "+v"(x) : [l] "v"(m));
  out[l] = x;
}
// a unit-lower-triangular solve inside each 16-lane row: 15 back-to-back dependent DPP steps in ONE asm block, so that
// nothing but the stated s_nop sits between them (a missing wait state shows as a wrong result)
#define TR(J, N) N "v_fmac_f64_dpp %[x], %[x], %[c" #J "] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define TRALL(N) TR(0, N) TR(1, N) TR(2, N) TR(3, N) TR(4, N) TR(5, N) TR(6, N) TR(7, N) TR(8, N) TR(9, N) TR(10, N) TR(11, N) TR(12, N) TR(13, N) TR(14, N)
template <int MODE>
__global__ void k_dpp_trsv(double* out, const double* co, int reps, long long* cyc) {
  int l = threadIdx.x, i = l & 15;
  double x = co[256 + l];
  double c[15];
  for (int j = 0; j < 15; j++) c[j] = i > j ? co[i * 16 + j] : 0.0;
  long long t0 = clock64();
  for (int r = 0; r < reps; r++) {
#define OPS [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), \
            [c8] "v"(c[8]), [c9] "v"(c[9]), [c10] "v"(c[10]), [c11] "v"(c[11]), [c12] "v"(c[12]), [c13] "v"(c[13]), [c14] "v"(c[14])
    if (MODE == 0) asm volatile(TRALL("") : [x] "+v"(x) : OPS);
    if (MODE == 1) asm volatile(TRALL("s_nop 0\n\t") : [x] "+v"(x) : OPS);
    if (MODE == 2) asm volatile(TRALL("s_nop 1\n\t") : [x] "+v"(x) : OPS);
  }
  long long t1 = clock64();
  out[l] = x;
  if (l == 0) cyc[0] = t1 - t0;
}
int main() {
  double* d; long long* dc; long long cyc[8];
  CK(hipMalloc(&d, 1 << 20)); CK(hipMalloc(&dc, 64 * 8));
  std::vector<double> h(64);
  k_dpp_sem<<<1, 64>>>(d); CK(hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost));
  int bad = 0; for (int l = 0; l < 64; l++) bad += h[l] != l + 1000.0 * ((l & ~15) + 5);
  printf("v_fmac_f64_dpp row_newbcast semantics: %s\n", bad ? "UNEXPECTED" : "ok (x += x[row lane J] * m)");
  std::vector<double> co(320); for (int k = 0; k < 256; k++) co[k] = -(0.25 + 0.001 * k);
  for (int k = 0; k < 64; k++) co[256 + k] = 1.0 + 0.01 * k;
  double* dco; CK(hipMalloc(&dco, 2560)); CK(hipMemcpy(dco, co.data(), 2560, hipMemcpyHostToDevice));
  for (int mode = 0; mode < 3; mode++) {
    for (int reps : {1, 1000}) {
      if (mode == 0) k_dpp_trsv<0><<<1, 64>>>(d, dco, reps, dc);
      if (mode == 1) k_dpp_trsv<1><<<1, 64>>>(d, dco, reps, dc);
      if (mode == 2) k_dpp_trsv<2><<<1, 64>>>(d, dco, reps, dc);
      CK(hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
      if (reps > 1) { printf("   %.1f cycles per step\n", (double)cyc[0] / reps / 15); continue; }
      bad = 0;
      for (int r = 0; r < 4; r++) {
        double x[16];
        for (int i = 0; i < 16; i++) x[i] = co[256 + 16 * r + i];
        for (int j = 0; j < 15; j++) for (int i = j + 1; i < 16; i++) x[i] = __builtin_fma(x[j], co[i * 16 + j], x[i]);
        for (int i = 0; i < 16; i++) bad += x[i] != h[16 * r + i];
      }
      printf("16x16 triangular solve, 15 dependent DPP fmacs with %d wait states between: %s (%d of 64 differ from the host's fma chain)", mode, bad ? "WRONG" : "bit-identical", bad);
    }
  }
  for (int row = 0; row < 4; row++) {
    if (row == 0) k_row_to_all<0><<<1, 64>>>(d); if (row == 1) k_row_to_all<1><<<1, 64>>>(d);
    if (row == 2) k_row_to_all<2><<<1, 64>>>(d); if (row == 3) k_row_to_all<3><<<1, 64>>>(d);
    CK(hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost));
    bad = 0; for (int l = 0; l < 64; l++) bad += h[l] != 100.0 + 16 * row + (l & 15);
    printf("row_to_all<%d>: %s  (lane 0 %.0f, lane 17 %.0f, lane 63 %.0f)\n", row, bad ? "WRONG" : "ok", h[0], h[17], h[63]);
  }
  const int iters = 4096;
  for (int nt : {64, 512}) {
    k_fmac_chain<<<1, nt>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
    printf("%3d threads: dependent v_fmac_f64 chain            %.1f cycles per step\n", nt, (double)cyc[0] / iters);
    k_dpp_chain<0><<<1, nt>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
    printf("%3d threads: dependent v_fmac_f64_dpp chain        %.1f cycles per step\n", nt, (double)cyc[0] / iters);
    k_dpp_chain<1><<<1, nt>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
    printf("%3d threads:   + 1 off-chain fmac_dpp per step      %.1f\n", nt, (double)cyc[0] / iters);
    k_dpp_chain<2><<<1, nt>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
    printf("%3d threads:   + 2                                  %.1f\n", nt, (double)cyc[0] / iters);
    k_dpp_chain<3><<<1, nt>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
    printf("%3d threads:   + 3                                  %.1f\n", nt, (double)cyc[0] / iters);
    k_dpp_chain<4><<<1, nt>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
    printf("%3d threads:   + 4                                  %.1f\n", nt, (double)cyc[0] / iters);
  }
  k_bcast_chain<<<1, 64>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
  printf("fmac_dpp -> permlane16/32 swap row broadcast (f64) -> ...   %.1f cycles per round\n", (double)cyc[0] / iters);
  k_bperm_chain<<<1, 64>>>(d, iters, dc); CK(hipMemcpy(cyc, dc, 8, hipMemcpyDeviceToHost));
  printf("fmac_dpp -> 2 x ds_bpermute_b32 -> ...                       %.1f cycles per round\n", (double)cyc[0] / iters);
  CK(hipDeviceSynchronize());
  return 0;
}
