// Micro-benchmarks that size the dense tail-block design (diagnostics only, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_mfma_layout(const double* A, const double* B, double* C) {   // A 16x4 row-major, B 4x16 row-major, C 16x16
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void k_mfma_rate(double* out, int iters, long long* cyc) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  d4 c[NACC];
  for (int i = 0; i < NACC; i++) c[i] = d4{0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__device__ __forceinline__ double __hip_ds_bpermute_dummy(double v, int i) {
  unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, i & 63), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), i & 63);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
template <int NCH>
__global__ void k_tput(double* out, int iters, long long* cyc) {
  double x[NCH];
  for (int k = 0; k < NCH; k++) x[k] = 1.0 + threadIdx.x * 1e-6 + k;
  double y = 1.0000001;
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < NCH; k++) x[k] = __builtin_fma(x[k], y, 1e-9);
  }
  long long t1 = clock64();
  double s = 0; for (int k = 0; k < NCH; k++) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain(double* out, int iters, long long* cyc, double x0) {
  int l = threadIdx.x;
  double x = x0 + l * 1e-9, y = 1.000001;
  long long t[8];
  t[0] = clock64();
#pragma unroll 16
  for (int i = 0; i < iters; i++) x = __builtin_fma(x, y, 1e-9);     // dependent fma chain
  t[1] = clock64();
#pragma unroll 8
  for (int i = 0; i < iters; i++) x = 1.0 / x;                       // dependent IEEE reciprocal chain
  t[2] = clock64();
#pragma unroll 8
  for (int i = 0; i < iters; i++) {                                  // rcp + 2 Newton steps
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    x = __builtin_fma(r, e, r);
  }
  t[3] = clock64();
#pragma unroll 16
  for (int i = 0; i < iters; i++) {                                  // readlane -> fma chain (tail chain step)
    double xq = __hip_ds_bpermute_dummy(x, i);
    x = __builtin_fma(-y, xq, x);
  }
  t[4] = clock64();
#pragma unroll 16
  for (int i = 0; i < iters; i++) x = x * y;                         // dependent mul chain
  t[5] = clock64();
  out[l] = x;
  if (l == 0) for (int k = 0; k < 5; k++) cyc[k] = t[k + 1] - t[k];
}

__global__ void k_lds_barrier(double* out, int iters, long long* cyc) {
  __shared__ double buf[1024];
  int t = threadIdx.x;
  buf[t] = t;
  __syncthreads();
  double x = 0;
  long long t0 = clock64();
#pragma unroll 4
  for (int i = 0; i < iters; i++) {              // write -> barrier -> read of another thread's value -> barrier
    buf[t] = x + 1.0;
    __syncthreads();
    x = buf[(t + 64) & 511];
    __syncthreads();
  }
  long long t1 = clock64();
#pragma unroll 8
  for (int i = 0; i < iters; i++) {              // same-wave LDS round trip: write then read another lane's value, no barrier
    buf[t] = x + 1.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    x = buf[t ^ 1];
  }
  long long t2 = clock64();
#pragma unroll 8
  for (int i = 0; i < iters; i++) __syncthreads();
  long long t3 = clock64();
  out[t] = x;
  if (t == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}

int main() {
  // ---- layout check with asymmetric integer data
  std::vector<double> A(64), B(64), C(256), Cref(256, 0.0);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 7 + k * 3;
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + k * 5 + j * 11;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) Cref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dC; long long* dcyc;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 1 << 20)); CK(hipMalloc(&dcyc, 4096 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  k_mfma_layout<<<1, 64>>>(dA, dB, dC);
  CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; i++) bad += C[i] != Cref[i];
  printf("mfma_f64_16x16x4 layout: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
  long long cyc[64];
  const int iters = 2000;
  // ---- MFMA issue rate: one wave per SIMD (256 threads), two waves per SIMD (512 threads)
  for (int nt : {64, 256, 512}) {
    k_mfma_rate<1><<<1, nt>>>(dC, iters, dcyc); CK(hipMemcpy(cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("mfma f64 16x16x4, %3d threads, 1 dependent accumulator : %.1f cycles per MFMA per wave\n", nt, (double)cyc[0] / iters);
    k_mfma_rate<2><<<1, nt>>>(dC, iters, dcyc); CK(hipMemcpy(cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("mfma f64 16x16x4, %3d threads, 2 accumulators          : %.1f cycles per MFMA per wave\n", nt, (double)cyc[0] / iters / 2);
    k_mfma_rate<4><<<1, nt>>>(dC, iters, dcyc); CK(hipMemcpy(cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("mfma f64 16x16x4, %3d threads, 4 accumulators          : %.1f cycles per MFMA per wave\n", nt, (double)cyc[0] / iters / 4);
  }
  k_chain<<<1, 64>>>(dC, iters, dcyc, 1.5); CK(hipMemcpy(cyc, dcyc, 5 * 8, hipMemcpyDeviceToHost));
  printf("dependent chains, one wave: fma %.1f  1.0/x %.1f  rcp+2NR %.1f  shfl-mul-mul-sub %.1f  mul %.1f cycles per step\n",
         (double)cyc[0] / iters, (double)cyc[1] / iters, (double)cyc[2] / iters, (double)cyc[3] / iters, (double)cyc[4] / iters);
  for (int nt : {64, 256, 512}) {
    k_tput<16><<<1, nt>>>(dC, iters, dcyc); CK(hipMemcpy(cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("independent f64 fma x16, %3d threads: %.2f cycles per fma instruction per wave\n", nt, (double)cyc[0] / iters / 16);
  }
  k_lds_barrier<<<1, 512>>>(dC, iters, dcyc); CK(hipMemcpy(cyc, dcyc, 3 * 8, hipMemcpyDeviceToHost));
  printf("512 threads: write-barrier-read-barrier %.1f  same-wave LDS write->read %.1f  bare barrier %.1f cycles\n",
         (double)cyc[0] / iters, (double)cyc[1] / iters, (double)cyc[2] / iters);
  CK(hipDeviceSynchronize());
  return 0;
}
