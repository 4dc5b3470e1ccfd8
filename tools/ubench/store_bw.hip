// Microbenchmark: what a CU can WRITE.  One workgroup of 4 waves per CU (one wave per SIMD, like the four-wave GEMM
// forms), each wave fires `iters` stores of 16 B per lane (1 KiB per instruction) into its own part of a C-like matrix
// and the kernel ends when they are acknowledged.  Patterns: 0 = 1 KiB contiguous per store; 1 = 16 rows x 64 B (the
// chunk layout of gemm_quad_stream_kernel, row pitch 4608 B); 2 = 16 rows x 32 B, 8 B per lane (the accumulator layout).
// In-kernel cycles per store (s_memtime over the issue loop, before the acknowledgements) and wall time per launch.
// Build: hipcc --offload-arch=gfx950 -O3 store_bw.hip -o store_bw ;  run: ./store_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PATTERN>
__global__ __launch_bounds__(256) void k(char *dst, size_t per_wg, int iters, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char *base = dst + (size_t)blockIdx.x * per_wg + (size_t)wave * (per_wg / 4);
  const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (PATTERN == 0) {
      *reinterpret_cast<u32x4 *>(base + (size_t)it * 1024 + lane * 16) = v;
    } else if (PATTERN == 1) {     // lane -> (row = lane & 15, 16-byte chunk = lane >> 4): 64 B per row, pitch 4608 B
      *reinterpret_cast<u32x4 *>(base + (size_t)(it / 72) * (16 * 4608) + (size_t)(lane & 15) * 4608 + (it % 72) * 64 + (lane >> 4) * 16) = v;
    } else {                       // 8 B per lane, 32 B per row
      *reinterpret_cast<u32x2 *>(base + (size_t)(it / 144) * (16 * 4608) + (size_t)(lane & 15) * 4608 + (it % 144) * 32 + (lane >> 4) * 8) = u32x2{v.x, v.y};
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
  const size_t per_wg = 4u << 20;          // 4 MiB per workgroup (1 MiB per wave)
  const int max_wg = 1024;
  char *dst;
  unsigned long long *cyc;
  CHECK(hipMalloc(&dst, per_wg * max_wg));
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 4 * max_wg));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  unsigned long long *h = (unsigned long long *)malloc(sizeof(unsigned long long) * 4 * max_wg);
  printf("# pattern  workgroups  stores/wave  bytes/store | wall us | GB/s chip | B/clk/CU at 2.0 GHz | issue cycles per store (median wave)\n");
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int wgs : {32, 128, 256, 1024})
      for (int iters : {30, 240}) {
        const int bytes = pattern == 2 ? 512 : 1024;
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
          CHECK(hipEventRecord(e0));
          if (pattern == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, dst, per_wg, iters, cyc);
          else if (pattern == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, dst, per_wg, iters, cyc);
          else hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, dst, per_wg, iters, cyc);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep > 0 && ms < best) best = ms;
        }
        CHECK(hipMemcpy(h, cyc, sizeof(unsigned long long) * 4 * wgs, hipMemcpyDeviceToHost));
        // median
        for (int a = 0; a < 4 * wgs; ++a)
          for (int b = a + 1; b < 4 * wgs && b < a + 1; ++b) {}
        unsigned long long sum = 0;
        for (int a = 0; a < 4 * wgs; ++a) sum += h[a];
        const double total = (double)wgs * 4 * iters * bytes;
        const int cus = wgs < 256 ? wgs : 256;
        printf("  %d  %5d  %4d  %5d | %8.2f | %8.1f | %6.2f | %7.1f (mean)\n", pattern, wgs, iters, bytes, best * 1e3, total / (best * 1e-3) / 1e9,
               total / cus / (best * 1e-3 * 2.0e9), (double)sum / (4.0 * wgs) / iters);
      }
  return 0;
}
