// Probe: v_mfma_f64_16x16x4_f64 operand/result lane maps (exact integer data, asymmetric B)
// and sustained issue rate. Build: hipcc --offload-arch=gfx950 -O3 probe_mfma_f64.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void layout(const double* A, const double* B, double* C) {
  // A is 16x4 row-major, B is 4x16 row-major, C is 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate(double* out, int iters, double x) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = x + threadIdx.x * 1e-3, b = x - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<double> A(64), B(64), C(256), R(256);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = i * 3 + k * 7 + 1;
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = k * 11 + j * j + 2;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
  double *dA, *dB, *dC;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout<<<1, 64>>>(dA, dB, dC);
  CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; i++) bad += (C[i] != R[i]);
  printf("layout mismatches: %d of 256\n", bad);
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* out; CK(hipMalloc(&out, 256 * 8 * 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu++) {
    int grid = p.multiProcessorCount * wgs_per_cu, iters = 2000;
    auto run = [&](int nacc) {
      if (nacc == 1) rate<1><<<grid, 256>>>(out, iters, 1.0);
      if (nacc == 4) rate<4><<<grid, 256>>>(out, iters, 1.0);
      if (nacc == 16) rate<16><<<grid, 256>>>(out, iters, 1.0);
    };
    for (int nacc : {1, 4, 16}) {
      run(nacc); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); run(nacc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)grid * 4 * iters * nacc * 2048.0;
      printf("wg/cu %d nacc %2d: %.3f ms  %.1f TFLOP/s f64\n", wgs_per_cu, nacc, ms, flops / ms * 1e-9);
    }
  }
  return 0;
}
