// Microbenchmark: per-CU bandwidth of L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) vs
// global_load_dwordx4 -> VGPR (-> ds_write_b128), 8 waves per workgroup, one workgroup per CU,
// source buffer small enough to stay L2-resident.  Build: hipcc --offload-arch=gfx950 -O3 dma_bw.hip -o dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void *g, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const char *src, size_t src_bytes, int iters, float *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  // each wave streams 1 KiB pieces; pieces of one workgroup are contiguous 8 KiB per step
  size_t off = ((size_t)blockIdx.x * 97 * 8192 + wave * 1024 + lane * 16) % src_bytes;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        glds16(src + off, __builtin_amdgcn_readfirstlane(lds0 + ((it * 8 + u) & 15) * 8192 + wave * 1024));
        off += 8192;
        if (off >= src_bytes) off -= src_bytes;
      }
      if ((it & 1) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // 8-16 pieces in flight per wave
    } else {
      const char *p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        p[u] = src + off;
        off += 8192;
        if (off >= src_bytes) off -= src_bytes;
      }
      u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
      // loads and their wait in ONE statement (outputs early-clobber): the compiler never sees an
      // un-landed destination register
      asm volatile(
          "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %9, off\n\t"
          "global_load_dwordx4 %2, %10, off\n\tglobal_load_dwordx4 %3, %11, off\n\t"
          "global_load_dwordx4 %4, %12, off\n\tglobal_load_dwordx4 %5, %13, off\n\t"
          "global_load_dwordx4 %6, %14, off\n\tglobal_load_dwordx4 %7, %15, off\n\ts_waitcnt vmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
          : "memory");
      if (MODE == 2) {
        char *d = smem + (it & 1) * 65536 + wave * 1024 + lane * 16;
        *reinterpret_cast<u32x4 *>(d + 0 * 8192) = v0; *reinterpret_cast<u32x4 *>(d + 1 * 8192) = v1;
        *reinterpret_cast<u32x4 *>(d + 2 * 8192) = v2; *reinterpret_cast<u32x4 *>(d + 3 * 8192) = v3;
        *reinterpret_cast<u32x4 *>(d + 4 * 8192) = v4; *reinterpret_cast<u32x4 *>(d + 5 * 8192) = v5;
        *reinterpret_cast<u32x4 *>(d + 6 * 8192) = v6; *reinterpret_cast<u32x4 *>(d + 7 * 8192) = v7;
      } else {
        acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x ^ v4.x ^ v5.x ^ v6.x ^ v7.x;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = (float)smem[0] + (float)acc.x;
}

int main() {
  const size_t src_bytes = 2u << 20;  // 2 MiB: L2 resident per XCD
  char *src; float *sink;
  CHECK(hipMalloc(&src, src_bytes));
  CHECK(hipMemset(src, 1, src_bytes));
  CHECK(hipMalloc(&sink, 4096));
  const int iters = 2000, grid = 256;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const char *names[3] = {"LDS-DMA global_load_lds_dwordx4", "global_load_dwordx4 -> VGPR", "global_load_dwordx4 -> VGPR -> ds_write_b128"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(a));
      if (mode == 0) { CHECK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 131072, 0, src, src_bytes, iters, sink); }
      if (mode == 1) { CHECK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 131072, 0, src, src_bytes, iters, sink); }
      if (mode == 2) { CHECK(hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 131072, 0, src, src_bytes, iters, sink); }
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      const double bytes = (double)grid * 8 * iters * 8 * 1024;
      if (rep) printf("%-48s %8.3f ms  %7.1f GB/s per CU  %7.2f TB/s chip  (%.1f B/clk/CU at 2.1 GHz)\n", names[mode], ms,
                      bytes / grid / ms / 1e6, bytes / ms / 1e9, bytes / grid / (ms * 1e-3) / 2.1e9);
    }
  }
  return 0;
}
