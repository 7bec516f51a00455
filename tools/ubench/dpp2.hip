#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STEP(J, N) N "v_fmac_f64_dpp %[x], %[x], %[l] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
template <int NS, int MODE>
__global__ void k(double* out) {
  int l = threadIdx.x, i = l & 15;
  double x = 1.0 + l;
  double c0 = i > 0 ? 10.0 : 0.0, c1 = i > 1 ? 100.0 : 0.0, c2 = i > 2 ? 1000.0 : 0.0;
  if (MODE == 0) {
    asm volatile(STEP(0, "") : [x] "+v"(x) : [l] "v"(c0));
    if (NS > 1) asm volatile(STEP(1, "") : [x] "+v"(x) : [l] "v"(c1));
    if (NS > 2) asm volatile(STEP(2, "") : [x] "+v"(x) : [l] "v"(c2));
  } else if (MODE == 1) {
    asm volatile(STEP(0, "s_nop 7\n\t") : [x] "+v"(x) : [l] "v"(c0));
    if (NS > 1) asm volatile(STEP(1, "s_nop 7\n\ts_nop 7\n\t") : [x] "+v"(x) : [l] "v"(c1));
    if (NS > 2) asm volatile(STEP(2, "s_nop 7\n\ts_nop 7\n\t") : [x] "+v"(x) : [l] "v"(c2));
  } else {
    double t;
    asm volatile("v_mov_b64_dpp %[t], %[x] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" : [t] "=&v"(t) : [x] "v"(x)); x = __builtin_fma(t, c0, x);
    if (NS > 1) { asm volatile("s_nop 1\n\tv_mov_b64_dpp %[t], %[x] row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" : [t] "=&v"(t) : [x] "v"(x)); x = __builtin_fma(t, c1, x); }
    if (NS > 2) { asm volatile("s_nop 1\n\tv_mov_b64_dpp %[t], %[x] row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" : [t] "=&v"(t) : [x] "v"(x)); x = __builtin_fma(t, c2, x); }
  }
  out[l] = x;
}
int main() {
  double* d; hipMalloc(&d, 4096); std::vector<double> h(64);
#define RUN(NS, MODE) k<NS, MODE><<<1, 64>>>(d); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost); printf("steps %d mode %d:", NS, MODE); for (int i = 0; i < 6; i++) printf(" %.0f", h[i]); printf(" | row1:"); for (int i = 16; i < 21; i++) printf(" %.0f", h[i]); printf("\n");
  RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(2, 1) RUN(3, 1) RUN(2, 2) RUN(3, 2)
  // expected row 0: step0: x_i += 10*x_0 (x0=1): 1 12 13 14 15 16; step1: x_i += 100*x_1(12) for i>1: 1 12 1213 1214 1215 1216; step2: += 1000*1213 for i>2: 1 12 1213 1214214 ...
  printf("expected 3 steps: 1 12 1213 1214214 1214215 1214216\n");
  return 0;
}
