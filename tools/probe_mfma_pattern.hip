// Probe: f64 MFMA rate with the GEMM kernel's register pattern (TM x TN accumulators, TM + TN operands)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template <int TM, int TN>
__global__ void __launch_bounds__(256, 2) pat(double* out, int iters, double x) {
  d4 acc[TM][TN];
  for (int a = 0; a < TM; a++) for (int b = 0; b < TN; b++) acc[a][b] = d4{0, 0, 0, 0};
  double af[TM], bf[TN];
  for (int a = 0; a < TM; a++) af[a] = x + threadIdx.x * 1e-3 + a;
  for (int b = 0; b < TN; b++) bf[b] = x - threadIdx.x * 1e-3 - b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
  }
  double s = 0;
  for (int a = 0; a < TM; a++) for (int b = 0; b < TN; b++) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int TM, int TN>
int run(double* out, int grid, const char* name) {
  int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  pat<TM, TN><<<grid, 256>>>(out, 100, 1.0); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); pat<TM, TN><<<grid, 256>>>(out, iters, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-10s grid %4d: %8.3f ms %6.1f TFLOP/s\n", name, grid, ms, (double)grid * 4 * iters * TM * TN * 2048.0 / ms * 1e-9);
  return 0;
}
int main() {
  double* out; CK(hipMalloc(&out, 256 * 8 * 4096));
  for (int g : {256, 512}) {
    run<8, 2>(out, g, "8x2"); run<4, 4>(out, g, "4x4"); run<2, 2>(out, g, "2x2"); run<1, 8>(out, g, "1x8"); run<4, 2>(out, g, "4x2"); run<2, 1>(out, g, "2x1"); run<1, 1>(out, g, "1x1");
  }
  return 0;
}
