// Probe: issue cost (cycles per wave64 instruction, one wavefront per SIMD) of the float64 vector instructions the
// covariance / contraction kernels are made of, measured with s_memtime around long unrolled chains of independent
// operations.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_f64_ops tools/probe_f64_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

enum { OP_FMA, OP_MUL, OP_ADD, OP_MAX, OP_RSQ, OP_RCP, OP_RNDNE, OP_LDEXP, OP_CVT_I32, OP_CVT_F32, OP_SQRT, OP_FMA32, OP_EXP32, OP_COUNT };

template <int OP>
__global__ void __launch_bounds__(256) rate(double* out, unsigned long long* stamps, int iters, double x) {
  double acc[16];
  for (int i = 0; i < 16; i++) acc[i] = 1.0 + i * 0.01 + threadIdx.x * 1e-6;
  float facc[16];
  for (int i = 0; i < 16; i++) facc[i] = 1.0f + i * 0.01f;
  const double a = x + 1e-9, b = 1e-9 * x;
  int iacc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == OP_FMA) acc[i] = __builtin_fma(acc[i], a, b);
      if (OP == OP_MUL) acc[i] = acc[i] * a;
      if (OP == OP_ADD) acc[i] = acc[i] + b;
      if (OP == OP_MAX) acc[i] = fmax(acc[i], b + i);
      if (OP == OP_RSQ) acc[i] = __builtin_amdgcn_rsq(acc[i]);
      if (OP == OP_RCP) acc[i] = __builtin_amdgcn_rcp(acc[i]);
      if (OP == OP_RNDNE) acc[i] = __builtin_rint(acc[i] * a);          // (mul + rndne: subtract OP_MUL)
      if (OP == OP_LDEXP) acc[i] = ldexp(acc[i], (int)threadIdx.x & 1);
      if (OP == OP_CVT_I32) { iacc += (int)acc[i]; }
      if (OP == OP_CVT_F32) { facc[i] += (float)acc[i]; }
      if (OP == OP_SQRT) acc[i] = __builtin_amdgcn_sqrt(acc[i]);
      if (OP == OP_FMA32) facc[i] = __builtin_fmaf(facc[i], (float)a, (float)b);
      if (OP == OP_EXP32) facc[i] = __builtin_amdgcn_exp2f(facc[i]);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = iacc;
  for (int i = 0; i < 16; i++) s += acc[i] + facc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, 256 * 8 * 4096));
  unsigned long long* st; CK(hipMalloc(&st, 8 * 4096));
  std::vector<unsigned long long> hs(4096);
  const char* names[OP_COUNT] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_rsq_f64", "v_rcp_f64", "v_mul+v_rndne_f64", "v_ldexp_f64",
                                 "v_cvt_i32_f64 (+add)", "v_cvt_f32_f64 (+add)", "v_sqrt_f64", "v_fma_f32", "v_exp_f32"};
  const int iters = 2000;
  for (int wpc = 1; wpc <= 2; wpc++) {        // workgroups per CU: 1 -> one wavefront per SIMD, 2 -> two
    const int grid = ncu * wpc;
    printf("---- %d wavefront(s) per SIMD ----\n", wpc);
#define RUN(OP) { rate<OP><<<grid, 256>>>(out, st, 10, 1.0); CK(hipDeviceSynchronize()); rate<OP><<<grid, 256>>>(out, st, iters, 1.0); CK(hipDeviceSynchronize()); \
      CK(hipMemcpy(hs.data(), st, 8 * grid, hipMemcpyDeviceToHost)); std::vector<double> c; for (int i = 0; i < grid; i++) c.push_back((double)hs[i] / (16.0 * iters)); \
      std::sort(c.begin(), c.end()); printf("%-24s %6.2f s_memtime ticks per instruction per wavefront (x%d resident)\n", names[OP], c[grid / 2], wpc); }
    RUN(OP_FMA) RUN(OP_MUL) RUN(OP_ADD) RUN(OP_MAX) RUN(OP_RSQ) RUN(OP_RCP) RUN(OP_RNDNE) RUN(OP_LDEXP) RUN(OP_CVT_I32) RUN(OP_CVT_F32) RUN(OP_SQRT)
    RUN(OP_FMA32) RUN(OP_EXP32)
  }
  return 0;
}
