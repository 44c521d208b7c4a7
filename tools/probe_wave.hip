// probe_wave.hip — exact-integer reproducibility probe of gemm_wave.hip's kernels (measurement tool, never shipped):
//   op(A) = lower-triangular ones (role 1) / dense ones (role 3), B[k][n] = (k + 1) * 4096 + (n & 1023): every product and
//   partial sum is an exact integer below 2^53, so any deviation names the k that went wrong.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gpitch_amd/csrc tools/probe_wave.hip -o tools/probe_wave
#include "../gpitch_amd/csrc/gemm_wave.hip"
#include <vector>
#include <cstdio>
#include <cstring>
#include <cmath>
gp_status launch_gemm_batched(gp_handle, const GemmProblem*, int, int, int, const GemmFlags&) { return GP_OK; }
const GpSwitches& gp_switches() { static const GpSwitches s; return s; }
bool gemm_strip_fused_contraction_ok(int, int, int) { return false; }
GpTimerScope::GpTimerScope(gp_handle h_, int w) : h(h_), which(w) {}
GpTimerScope::~GpTimerScope() {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int role = argc > 1 ? atoi(argv[1]) : 1, M = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 32768;
  const int batch = argc > 4 ? atoi(argv[4]) : 8, reps = argc > 5 ? atoi(argv[5]) : 20;
  gp_handle_s hs; hs.device = 0; hs.stream = nullptr;
  std::vector<double> hA((size_t)M * M), hB((size_t)M * N), hv((size_t)N, 1.0);
  for (int i = 0; i < M; i++) for (int k = 0; k < M; k++) hA[(size_t)i * M + k] = (role == 3 || k <= i) ? 1.0 : 0.0;
  for (int k = 0; k < M; k++) for (int n = 0; n < N; n++) hB[(size_t)k * N + n] = (double)(k + 1) * 4096.0 + (double)(n & 1023);
  double *dA, *dB, *dC, *dv; GemmProblem* dP;
  CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dB, hB.size() * 8 * batch)); CK(hipMalloc(&dC, hB.size() * 8 * batch)); CK(hipMalloc(&dv, N * 8));
  CK(hipMalloc(&dP, sizeof(GemmProblem) * batch));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, hv.data(), N * 8, hipMemcpyHostToDevice));
  std::vector<GemmProblem> hp(batch);
  for (int b = 0; b < batch; b++) {
    CK(hipMemcpy(dB + (size_t)b * M * N, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
    GemmProblem p; memset(&p, 0, sizeof(p));
    p.A = dA; p.lda = M; p.B = dB + (size_t)b * M * N; p.ldb = N; p.C = dC + (size_t)b * M * N; p.ldc = N; p.M = M; p.N = N; p.K = M; p.v1 = dv;
    hp[b] = p;
  }
  CK(hipMemcpy(dP, hp.data(), sizeof(GemmProblem) * batch, hipMemcpyHostToDevice));
  GemmFlags f; f.role = role; f.uniform_aligned = 1; f.rows64_ok = 1; f.epilogue = EPI_STORE; f.big_tiles = 1;
  if (role == 1) f.triA = TRI_LOWER;
  if (role == 3) { f.scale_mode = 1; f.alpha = 1.0; }
  std::vector<double> out((size_t)M * N * batch);
  long total_bad = 0;
  for (int r = 0; r < reps; r++) {
    CK(hipMemset(dC, 0xff, out.size() * 8));
    gp_status st = GP_OK;
    if (!launch_gemm_wave(&hs, dP, batch, M, N, f, &st) || st != GP_OK) { printf("launch refused / failed: %s\n", hs.last_error.c_str()); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dC, out.size() * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int b = 0; b < batch; b++)
      for (int i = 0; i < M; i++) {
        const int kend = (role == 3) ? M : i + 1;
        const double sk = 4096.0 * ((double)kend * (kend + 1) / 2.0);
        for (int n = 0; n < N; n++) {
          const double want = sk + (double)kend * (double)(n & 1023), got = out[((size_t)b * M + i) * N + n];
          if (got != want) {
            if (bad < 12) printf("  rep %d gp %d row %3d (tile %d, +%2d) col %5d (group %3d, wave %d, +%2d): got - want = %.17g  (/4096 = %.6f)\n", r, b, i, i / 64, i % 64, n,
                                 n / 256, (n / 64) & 3, n % 64, got - want, (got - want) / 4096.0);
            bad++;
          }
        }
      }
    if (bad) printf("rep %d: %ld wrong entries\n", r, bad);
    total_bad += bad;
  }
  printf("role %d M %d N %d batch %d reps %d: %ld wrong entries in total\n", role, M, N, batch, reps, total_bad);
  return 0;
}
