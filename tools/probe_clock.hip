// Probe: sustained shader clock under a long f64 MFMA load, and its dependence on the operand data.
// clock64() counts shader cycles, wall_clock64() the constant 100 MHz timer: their ratio over a long loop is the
// effective frequency the power manager grants.  Operands: zeros / constants / pseudo-random mantissas.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void __launch_bounds__(256, 2) mfma_load(double* out, long long* clk, int iters, int mode) {
  constexpr int TM = 8, TN = 2;
  d4 acc[TM][TN];
  for (int a = 0; a < TM; a++) for (int b = 0; b < TN; b++) acc[a][b] = d4{0, 0, 0, 0};
  double af[TM], bf[TN];
  unsigned long long s = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 977 * blockIdx.x);
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5; };
  for (int a = 0; a < TM; a++) af[a] = mode == 0 ? 0.0 : (mode == 1 ? 1.0 : rnd());
  for (int b = 0; b < TN; b++) bf[b] = mode == 0 ? 0.0 : (mode == 1 ? 1.0 : rnd() * 1e-3);
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double t = 0;
  for (int a = 0; a < TM; a++) for (int b = 0; b < TN; b++) t += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  const int grid = 512, iters = 40000;
  double* out; long long* clk; CK(hipMalloc(&out, 256 * 8 * grid)); CK(hipMalloc(&clk, 16 * grid));
  long long h[2 * grid];
  const char* names[3] = {"zeros", "ones", "random"};
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 3; mode++) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      mfma_load<<<grid, 256>>>(out, clk, iters, mode);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
      double sc = 0, sw = 0;
      for (int b = 0; b < grid; b++) { sc += h[2 * b]; sw += h[2 * b + 1]; }
      const double ghz = sc / sw * 0.1;   // wall clock = 100 MHz
      const double tf = (double)grid * 4 * iters * 16 * 2048.0 / ms * 1e-9;
      printf("%-7s %8.2f ms  %6.1f TFLOP/s  shader clock %.3f GHz  cycles/MFMA/SIMD %.1f\n", names[mode], ms, tf, ghz,
             sc / grid / ((double)iters * 16) / 2.0);
    }
  return 0;
}
