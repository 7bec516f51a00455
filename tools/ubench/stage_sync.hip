// Micro-benchmark (diagnostics only): a "chain" wave that alternates ~WORK cycles of dependent arithmetic with reading a 8-KiB staging
// slot (8 x ds_read_b128), and helper waves that fill the other slot between two workgroup barriers (4 LDS gathers, 2 x ds_write_b128
// each) — the hand-over pattern of the staged tail chain.  Reported: cycles per set for the chain, and what the
// helpers spend working / waiting.
//   hipcc --offload-arch=gfx950 -O2 [-DCHAIN_DPP] [-DHELPER_UNROLL=25] -o stage_sync stage_sync.hip && ./stage_sync          (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int HELPERS>
__global__ __launch_bounds__(512) void k(long long* out, double* sink, int sets, int nwork, int lds_doubles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < lds_doubles; i += 512) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  double* stage = lds + (lds_doubles - 2048);      // two slots of 1024 doubles at the top
  double x = 1.0 + 1e-9 * threadIdx.x;
  long long t_total = 0, t_work = 0, t_wait = 0;
  if (wave == 0) {
    const long long t0 = clock64();
    for (int q = 0; q < sets; q++) {
      wg_barrier();
      const f64x2* s = reinterpret_cast<const f64x2*>(stage + (q & 1) * 1024) + lane;
      f64x2 c[8];
#pragma unroll
      for (int p = 0; p < 8; p++) c[p] = s[p * 64];
#ifdef CHAIN_DPP      // the chain's arithmetic as in the kernel: dependent DPP multiply-adds behind their wait states
      for (int i = 0; i < nwork; i++) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(1e-9));
#else
      for (int i = 0; i < nwork; i++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x));
#endif
#pragma unroll
      for (int p = 0; p < 8; p++) x += c[p][0] * 1e-30 + c[p][1] * 1e-30;
    }
    t_total = clock64() - t0;
  } else {
    const int kk = (wave - 1) & 3;
    unsigned idx = (unsigned)(lane * 37 + wave * 101) % (unsigned)(lds_doubles - 4096);
    double v0 = lds[idx], v1 = lds[idx + 517], v2 = lds[idx + 1033], v3 = lds[idx + 1549];
    long long tp = clock64();
#ifdef HELPER_UNROLL      // the helpers' loop body as straight-line code, each pass new to the instruction buffer (as in the kernel's unrolled form)
#pragma unroll HELPER_UNROLL
#endif
    for (int q = 0; q < sets; q++) {
      if (wave <= HELPERS) {
        f64x2* s = reinterpret_cast<f64x2*>(stage + (q & 1) * 1024) + lane;
        s[(2 * kk) * 64] = f64x2{v0, v1};
        s[(2 * kk + 1) * 64] = f64x2{v2, v3};
        idx = (idx * 5u + 77u) % (unsigned)(lds_doubles - 4096);
        v0 = lds[idx]; v1 = lds[idx + 517]; v2 = lds[idx + 1033]; v3 = lds[idx + 1549];
      }
      const long long ta = clock64();
      wg_barrier();
      const long long tb = clock64();
      t_work += ta - tp; t_wait += tb - ta; tp = tb;
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x * 3] = t_total / sets;
  if (threadIdx.x == 64) { out[blockIdx.x * 3 + 1] = t_work / sets; out[blockIdx.x * 3 + 2] = t_wait / sets; }
  sink[blockIdx.x * 512 + threadIdx.x] = x;
}

int main() {
  const int nblk = 256, sets = 2000;
  long long* d_out; double* d_sink;
  CK(hipMalloc(&d_out, nblk * 3 * sizeof(long long)));
  CK(hipMalloc(&d_sink, nblk * 512 * sizeof(double)));
  long long h[3 * 256];
  for (int big = 0; big < 2; big++) {
    const int lds_doubles = big ? 20000 : 6144;      // 160 KB (one workgroup per CU, like the tot kernel) | 48 KB
    const size_t bytes = (size_t)lds_doubles * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    for (int helpers : {0, 4, 7}) {
      for (int nwork : {0, 50, 100}) {
        if (helpers == 7) hipLaunchKernelGGL(k<7>, dim3(nblk), dim3(512), bytes, 0, d_out, d_sink, sets, nwork, lds_doubles);
        else if (helpers == 4) hipLaunchKernelGGL(k<4>, dim3(nblk), dim3(512), bytes, 0, d_out, d_sink, sets, nwork, lds_doubles);
        else hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(512), bytes, 0, d_out, d_sink, sets, nwork, lds_doubles);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < nblk; i++) { a += h[3 * i]; b += h[3 * i + 1]; c += h[3 * i + 2]; }
        printf("LDS %3zu KB, %d working helpers, chain arithmetic %3d fma: chain %6.0f cycles per set; helper works %5.0f, waits %5.0f\n", bytes / 1024, helpers, nwork,
               a / nblk, b / nblk, c / nblk);
      }
    }
  }
  return 0;
}
