// Micro-benchmark (diagnostics only, not product code): how long a wave that waits at s_barrier takes to come back after the
// last wave of the workgroup has arrived, as a function of how long it waited — and the same hand-over through an LDS flag the
// waiting waves poll.  One workgroup of 8 waves per CU-sized grid; wave 0 "works" (a dependent f64 chain) for WORK iterations per
// round, the others have nothing to do.  Reported: cycles from wave 0's arrival (its clock before the barrier / flag store) to the
// moment wave 1 runs again (its clock behind the barrier / behind the poll loop), averaged over the rounds.
//   hipcc --offload-arch=gfx950 -O2 -o barrier_wake barrier_wake.hip && ./barrier_wake          (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ double work(double x, int n) {
  for (int i = 0; i < n; i++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x));
  return x;
}

// MODE 0: s_barrier.  MODE 1: LDS flag, waiting waves poll with s_sleep 1 between polls.  MODE 2: LDS flag, polled without sleeping.
template <int MODE>
__global__ __launch_bounds__(512) void k(long long* out, double* sink, int rounds, int nwork) {
  __shared__ long long t_arrive;
  __shared__ volatile unsigned flag;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  double x = 1.0 + 1e-9 * threadIdx.x;
  long long acc_wake = 0, acc_work = 0;
  for (int r = 1; r <= rounds; r++) {
    if (wave == 0) {
      const long long t0 = clock64();
      x = work(x, nwork);
      const long long t1 = clock64();
      if (lane == 0) t_arrive = t1;
      acc_work += t1 - t0;
      if (MODE == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) flag = (unsigned)r;
      }
    } else {
      if (MODE == 0) {
        __builtin_amdgcn_s_barrier();
      } else {
        unsigned guard = 1u << 22;
        while ((unsigned)__builtin_amdgcn_readfirstlane((int)flag) < (unsigned)r && --guard)
          if (MODE == 1) __builtin_amdgcn_s_sleep(1);
      }
      const long long t2 = clock64();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (wave == 1 && lane == 0) acc_wake += t2 - t_arrive;
    }
    __syncthreads();      // (keeps the rounds apart; not timed)
  }
  if (threadIdx.x == 0) out[blockIdx.x * 2] = acc_work / rounds;
  if (threadIdx.x == 64) out[blockIdx.x * 2 + 1] = acc_wake / rounds;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
  const int nblk = 256, rounds = 200;
  long long* d_out; double* d_sink;
  CK(hipMalloc(&d_out, nblk * 2 * sizeof(long long)));
  CK(hipMalloc(&d_sink, nblk * 512 * sizeof(double)));
  long long h[2 * 256];
  for (int mode = 0; mode < 3; mode++) {
    for (int nwork : {0, 10, 50, 100, 400, 1000, 3000}) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(512), 0, 0, d_out, d_sink, rounds, nwork);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(512), 0, 0, d_out, d_sink, rounds, nwork);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(512), 0, 0, d_out, d_sink, rounds, nwork);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
      double w = 0, k2 = 0;
      for (int b = 0; b < nblk; b++) { w += h[2 * b]; k2 += h[2 * b + 1]; }
      printf("mode %d (%s)  wave 0 works %7.0f cycles  ->  wave 1 back after %7.0f cycles\n", mode,
             mode == 0 ? "s_barrier" : mode == 1 ? "LDS flag, s_sleep 1" : "LDS flag, busy poll", w / nblk, k2 / nblk);
    }
  }
  return 0;
}
