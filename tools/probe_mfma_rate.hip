// Probe: sustained v_mfma_f64_16x16x4_f64 / v_mfma_f32_32x32x2_f32 / v_fma_f64 rates with in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int NACC>
__global__ void __launch_bounds__(256) rate_f64(double* out, unsigned long long* stamps, int iters, double x) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = x + threadIdx.x * 1e-3, b = x - threadIdx.x * 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_f32(float* out, unsigned long long* stamps, int iters, float x) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
  float a = x + threadIdx.x * 1e-3f, b = x - threadIdx.x * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void __launch_bounds__(256) rate_fma64(double* out, unsigned long long* stamps, int iters, double x) {
  double acc[16];
  for (int i = 0; i < 16; i++) acc[i] = i;
  double a = x + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += acc[i];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, 256 * 8 * 8192));
  unsigned long long* st; CK(hipMalloc(&st, 16 * 8192));
  std::vector<unsigned long long> hs(2 * 8192);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* name, int grid, double flops, float ms, double instr_per_wave) {
    hipMemcpy(hs.data(), st, 16 * grid, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (int i = 0; i < grid; i++) { clk.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 100.0); cyc.push_back((double)hs[2 * i] / instr_per_wave); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    printf("%-28s grid %4d: %8.3f ms %7.1f TFLOP/s  in-kernel clk %.0f MHz  cyc/instr/wave %.1f\n", name, grid, ms, flops / ms * 1e-9, clk[grid / 2], cyc[grid / 2]);
  };
  for (int rep = 0; rep < 1; rep++) {
    for (int wpc = 1; wpc <= 8; wpc *= 2) {
      int grid = ncu * wpc, iters = 30000;
      float ms;
      rate_f64<4><<<grid, 256>>>(out, st, 1000, 1.0); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); rate_f64<4><<<grid, 256>>>(out, st, iters, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      report("mfma_f64_16x16x4 nacc4", grid, (double)grid * 4 * iters * 4 * 2048.0, ms, (double)iters * 4);
      CK(hipEventRecord(e0)); rate_f64<8><<<grid, 256>>>(out, st, iters / 2, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      report("mfma_f64_16x16x4 nacc8", grid, (double)grid * 4 * (iters / 2) * 8 * 2048.0, ms, (double)(iters / 2) * 8);
      CK(hipEventRecord(e0)); rate_f32<4><<<grid, 256>>>((float*)out, st, iters, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      report("mfma_f32_32x32x2 nacc4", grid, (double)grid * 4 * iters * 4 * 4096.0, ms, (double)iters * 4);
      CK(hipEventRecord(e0)); rate_fma64<<<grid, 256>>>(out, st, iters, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      report("v_fma_f64 x16", grid, (double)grid * 256 * iters * 16 * 2.0, ms, (double)iters * 16);
    }
  }
  return 0;
}
