// Microbenchmark: what this MI355X sustains, to put beside the spec peaks DESIGN.md divides by
// (SURVEY.md §8d asks for both).  (1) HBM streaming: read-only, write-only and copy over buffers far
// larger than the 256 MB Infinity Cache.  (2) bare bf16 MFMA loop (v_mfma_f32_16x16x32_bf16, operands in
// registers, one or two waves per SIMD) on random and on zero operands, with the in-kernel clock
// (s_memtime / s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O3 peaks.hip -o peaks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ src, size_t n, unsigned *sink) {
  u32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 v = __builtin_nontemporal_load(src + i);
    a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
  }
  if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) *sink = 1;
}
__global__ __launch_bounds__(256) void k_write(u32x4 *__restrict__ dst, size_t n) {
  const u32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, dst + i);
}
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// 12 independent accumulators per wave, operands fixed in registers
__global__ __launch_bounds__(512) void k_mfma(const uint4 *__restrict__ opnd, int iters, float *sink,
                                              unsigned long long *clk) {
  const int tid = threadIdx.x;
  uint4 a[4], b[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = opnd[(tid * 7 + i) & 4095];
#pragma unroll
  for (int i = 0; i < 3; ++i) b[i] = opnd[(tid * 7 + 4 + i) & 4095];
  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8 *>(&a[i]),
                                                            *reinterpret_cast<bf16x8 *>(&b[j]), acc[i][j], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 1.2345f) *sink = s;
  if (tid == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

static float timeit(void (*fn)(void *), void *ctx, int reps) {
  hipEvent_t s = nullptr, e = nullptr;
  CHECK(hipEventCreate(&s));
  CHECK(hipEventCreate(&e));
  fn(ctx);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(s));
  for (int i = 0; i < reps; ++i) fn(ctx);
  CHECK(hipEventRecord(e));
  CHECK(hipEventSynchronize(e));
  float ms;
  CHECK(hipEventElapsedTime(&ms, s, e));
  return ms / reps;
}

struct Ctx { u32x4 *a, *b; size_t n; unsigned *sink; int mode; uint4 *opnd; int iters, threads, blocks; float *fs; unsigned long long *clk; };
static void run_stream(void *p) {
  Ctx *c = (Ctx *)p;
  const int grid = 256 * 16;
  if (c->mode == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, c->a, c->n, c->sink);
  else if (c->mode == 1) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, c->b, c->n);
  else hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, c->a, c->b, c->n);
}
static void run_mfma(void *p) {
  Ctx *c = (Ctx *)p;
  hipLaunchKernelGGL(k_mfma, dim3(c->blocks), dim3(c->threads), 0, 0, c->opnd, c->iters, c->fs, c->clk);
}

int main() {
  Ctx c{};
  const size_t bytes = (size_t)2 << 30;  // 2 GiB per buffer
  c.n = bytes / 16;
  CHECK(hipMalloc(&c.a, bytes));
  CHECK(hipMalloc(&c.b, bytes));
  CHECK(hipMalloc(&c.sink, 4));
  CHECK(hipMemset(c.a, 1, bytes));
  CHECK(hipMemset(c.b, 2, bytes));
  const char *names[3] = {"read ", "write", "copy "};
  // 2 GiB streams from / to HBM; the smaller working sets are re-touched every launch and show what the
  // 256 MB Infinity Cache (and, at 16 MB, the 8 x 4 MB L2s) add
  const size_t sizes[5] = {bytes, (size_t)512 << 20, (size_t)128 << 20, (size_t)48 << 20, (size_t)16 << 20};
  for (int zi = 0; zi < 5; ++zi)
    for (int m = 0; m < 3; ++m) {
      c.mode = m;
      c.n = sizes[zi] / 16;
      const float ms = timeit(run_stream, &c, zi == 0 ? 10 : 50);
      const double moved = (m == 2 ? 2.0 : 1.0) * sizes[zi];
      printf("stream %s %5zu MiB per buffer: %.4f ms  %.0f GB/s\n", names[m], sizes[zi] >> 20, ms, moved / ms / 1e6);
    }
  // MFMA
  unsigned short *h = (unsigned short *)malloc(4096 * 16);
  CHECK(hipMalloc(&c.opnd, 4096 * 16));
  CHECK(hipMalloc(&c.fs, 4));
  CHECK(hipMalloc(&c.clk, 4096 * 16));
  unsigned long long *hclk = (unsigned long long *)malloc(4096 * 16);
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) {
      const float v = zero ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
      unsigned u;
      memcpy(&u, &v, 4);
      h[i] = (unsigned short)(u >> 16);
    }
    CHECK(hipMemcpy(c.opnd, h, 4096 * 16, hipMemcpyHostToDevice));
    for (int wps = 1; wps <= 2; ++wps) {
      c.threads = 256 * wps;
      c.blocks = 256;
      c.iters = 20000;
      // ~2 s of back-to-back launches so the clock settles (microarch guide, DVFS item 6)
      const float ms = timeit(run_mfma, &c, 40);
      CHECK(hipMemcpy(hclk, c.clk, 256 * 16, hipMemcpyDeviceToHost));
      double ghz = 0;
      for (int i = 0; i < 256; ++i) ghz += (double)hclk[2 * i] / (double)hclk[2 * i + 1] * 0.1;
      ghz /= 256;
      const double fl = 2.0 * 16 * 16 * 32 * 12 * (double)c.iters * (c.threads / 64) * c.blocks;
      printf("MFMA bf16 16x16x32 %s operands, %d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s  in-kernel clock %.2f GHz\n",
             zero ? "zero  " : "random", wps, ms, fl / ms / 1e9, ghz);
    }
  }
  return 0;
}
