// Diagnostic: which XCD (XCC_ID hardware register) does workgroup b of a 1-D grid land on?
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/xcc_map.hip -o tools/ubench/xcc_map ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out) {
  const unsigned v = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID, bits [3:0] = XCC id
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_REG_HW_ID
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)(v & 15); out[2 * blockIdx.x + 1] = (int)hw; }
  // keep the workgroup alive a little so the whole grid is co-resident like a GEMM round
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);
}
int main(int argc, char **argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 512, n = argc > 2 ? atoi(argv[2]) : 256;
  const int lds = argc > 3 ? atoi(argv[3]) : 0;
  int *d;
  hipMalloc(&d, n * 2 * sizeof(int));
  if (lds > 65536) hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(probe, dim3(n), dim3(threads), lds, 0, d);
  std::vector<int> h(2 * n);
  hipMemcpy(h.data(), d, n * 2 * sizeof(int), hipMemcpyDeviceToHost);
  printf("threads %d grid %d lds %d\nblock -> xcc:", threads, n, lds);
  for (int b = 0; b < 64 && b < n; ++b) printf(" %d", h[2 * b]);
  printf("\n");
  int agree = 0;
  for (int b = 0; b < n; ++b) agree += h[2 * b] == (b & 7);
  printf("blocks with xcc == b %% 8: %d of %d\n", agree, n);
  return 0;
}
