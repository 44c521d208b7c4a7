// sgpr.hip — SGPRSS plan: Titsias collapsed bound with a sum-of-pitch kernel, sparse predictions and the
// per-source exact-GP posterior.  Mirrors gpitch/sgpr_ss.py:29-114 and GPflow-0.5 SGPR.build_predict.
//   Kuf = sum_p K_p(Z,X); L = chol(Kuu + jitter I); A = L^-1 Kuf / sigma; B = A A^T + I; LB = chol(B);
//   c = LB^-1 (A err) / sigma; bound = -N/2 log 2pi - sum log diag LB - N/2 log s2 - |err|^2/(2 s2)
//                                     + |c|^2/2 - sum Kdiag/(2 s2) + tr(A A^T)/2  [- 1000 sum |v_p|]
// Everything is assembled from the same kernels as the Pdgp path (cov.hip, chol.hip, gemm.hip); the only
// new device code is the handful of small scalar/vector kernels below.
#include "engine.h"
#include <string.h>

struct gp_sgpr_plan_s {
  gp_handle h = nullptr;
  int P = 0, maxN = 0, M = 0, reg = 0;
  int f32 = 0;                 // gp_sgpr_set_precision: Kuf, A, Kuf_bar strips in float32 (gemm_f32.hip)
  double jitter = 1e-6;
  std::vector<int> ktype, m;
  std::vector<int64_t> off_theta;
  int64_t nparams = 0;
  int maxm = 0;
  void* ws = nullptr; size_t ws_bytes = 0;
  // workspace
  double *L = nullptr, *W = nullptr, *Kuf = nullptr, *A = nullptr, *H = nullptr, *LB = nullptr, *WB = nullptr;
  double *feat = nullptr, *s1 = nullptr, *s2 = nullptr, *dot = nullptr, *u = nullptr, *c = nullptr, *slabs = nullptr;
  // backward
  double *E2 = nullptr, *T1 = nullptr, *T2 = nullptr, *Wbar = nullptr, *R = nullptr, *Binv = nullptr, *G = nullptr;
  double *ubar = nullptr, *Lu = nullptr, *alpha = nullptr, *ones = nullptr, *hyp = nullptr, *hyp_uu = nullptr;
  size_t hyp_stride = 0;    // doubles between the per-kernel partial record sets in hyp
  size_t hyp_uu_stride = 0; // the same for the Kuu-side record sets in hyp_uu
  std::vector<char> h_tail_fin;   // the batched gradient tail's finish items as last uploaded (sgpr_backward)
  double *upart = nullptr;  // [max(nsplit, 2)][M] row-dot partials of u = A' y, fused into the H = A' A'^T launch
  double *red = nullptr;    // [2][SG_RED_BLOCKS] partial sums of the two N-long reductions (tr H, sum y^2)
  double *scal = nullptr;   // [0] bound, [1] sum err^2, [2] sum colsumsq(A'), [3] kdiag total per point, [4] dF/dkd, [5] dF/ds
  char* d_desc = nullptr; std::vector<char> h_desc[2];   // two descriptor blocks (training pass / prediction pass)
  int nsplit = 2;
  // M > 256: Kuu and B are factored by one resident launch and inverted block by block (chol.hip: chol_inverse_blocked_*),
  // descriptors prepared when the workspace is set
  void *chol_ws_kuu = nullptr, *chol_ws_b = nullptr; size_t chol_ws_bytes = 0; bool chol_blocked = false;
  // bound+gradient evaluations are launch-bound at window sizes (N ~ 2001): once the device descriptors match
  // the argument pointers the whole kernel sequence is captured into a hipGraph and replayed (L-BFGS-B calls it
  // dozens of times per window with the same buffers)
  struct EvalKey {
    const double *params = nullptr, *X = nullptr, *Y = nullptr, *Z = nullptr; double *grad = nullptr; int N = -1;
    bool operator==(const EvalKey& o) const {
      return params == o.params && X == o.X && Y == o.Y && Z == o.Z && grad == o.grad && N == o.N;
    }
  };
  EvalKey desc_key; bool desc_valid = false;   // what descriptor slots 0/1 on the device currently describe
  EvalKey graph_key; hipGraphExec_t gexec = nullptr;
  bool skip_upload = false;                    // set while re-enqueueing with valid device descriptors
  int graphs = 1;                              // gp_sgpr_set_graphs
  int64_t n_eager = 0, n_captured = 0, n_replayed = 0;
  const double* grad_Y = nullptr; int grad_N = -1;     // data of the last bound + gradient evaluation (gp_sgpr_residual_grad)
  // frame-sharded evaluation: what gp_sgpr_bound_begin staged for gp_sgpr_bound_end
  int staged_N = -1; const double* staged_params = nullptr; const double* staged_X = nullptr;
  ~gp_sgpr_plan_s() { if (gexec) (void)hipGraphExecDestroy(gexec); }
};

static inline int64_t ldN64(int N) { return (N + 1) & ~1; }
// any other entry point rewrites the device descriptor blocks the recorded graph reads
static inline void sg_invalidate(gp_sgpr_plan_s* p) {
  p->grad_Y = nullptr; p->grad_N = -1;
  p->desc_valid = false;
  if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
}
static const size_t SG_DESC_BYTES = 16 * 1024;

// ---- small kernels ----------------------------------------------------------------------------------
// C[i][i] += mul * s[0] + add   (s may be null)
__global__ void __launch_bounds__(256) add_diag_kernel(double* __restrict__ C, int M, int64_t ld,
                                                       const double* __restrict__ s, double mul, double add) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < M) C[(int64_t)i * ld + i] += (s ? mul * s[0] : 0.0) + add;
}

// B = H / s2 + I
__global__ void __launch_bounds__(256) sgpr_B_kernel(const double* __restrict__ H, double* __restrict__ B, int M,
                                                     const double* __restrict__ s2) {
  const double inv = 1.0 / s2[0];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)M * M; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    B[idx] = H[idx] * inv + (i == j ? 1.0 : 0.0);
  }
}

// The two N-long sums of a bound evaluation — sum over the [rb][N] column partials of A'^2 (= tr H) and sum y^2 — as
// SG_RED_BLOCKS partial sums each in ONE launch (grid (SG_RED_BLOCKS, 2)) and one small launch that adds them in a fixed
// order: one block walking 262 144 doubles took 0.34 ms of a 4.5-ms evaluation at N = 65536.
#define SG_RED_BLOCKS 128
__global__ void __launch_bounds__(256) sgpr_sums_partial_kernel(const double* __restrict__ s1, int64_t n1,
                                                                const double* __restrict__ y, int64_t n2,
                                                                double* __restrict__ part) {
  const bool sq = (blockIdx.y == 1);
  const double* v = sq ? y : s1;
  const int64_t n = sq ? n2 : n1;
  // contiguous chunk per block (multiple of 256 elements), fixed order inside it
  const int64_t per = ((n + SG_RED_BLOCKS - 1) / SG_RED_BLOCKS + 255) & ~(int64_t)255;
  const int64_t beg = (int64_t)blockIdx.x * per, end = (beg + per < n) ? beg + per : n;
  double a = 0.0;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) { const double x = v[i]; a = sq ? fma(x, x, a) : a + x; }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.y * SG_RED_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// scal[2] = sum of the first SG_RED_BLOCKS partials (tr H), scal[1] = sum of the second (sum y^2): wavefront w does row w
__global__ void __launch_bounds__(128) sgpr_sums_final_kernel(const double* __restrict__ part, double* __restrict__ scal) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  double a = 0.0;
  for (int i = l; i < SG_RED_BLOCKS; i += 64) a += part[w * SG_RED_BLOCKS + i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if (l == 0) scal[w == 0 ? 2 : 1] = a;
}

// c = WB (u / s2): one wavefront per row (coalesced reads of the lower-triangular row)
__global__ void __launch_bounds__(256) sgpr_c_kernel(const double* __restrict__ WB, const double* __restrict__ u,
                                                     double* __restrict__ c, int M, const double* __restrict__ params) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (i >= M) return;
  double acc = 0.0;
  for (int k = l; k <= i; k += 64) acc = fma(WB[(int64_t)i * M + k], u[k], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if (l == 0) c[i] = acc / params[0];
}

// r[n] = sum_i A'[i][n] dFdu[i] - y[n] / s2 = d bound / d err_n (err = Y - mean_function(X), sgpr_ss.py:40): what a trainable
// mean function's parameters are differentiated through.  One thread per frame, the strip read once (coalesced rows).
__global__ void __launch_bounds__(256) sgpr_resid_grad_kernel(const double* __restrict__ A, int64_t ld, int a_f32,
                                                              const double* __restrict__ dFdu, const double* __restrict__ y,
                                                              int M, int N, const double* __restrict__ params,
                                                              double* __restrict__ r) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  double acc = 0.0;
  if (a_f32) {
    const float* Af = reinterpret_cast<const float*>(A);
    for (int i = 0; i < M; i++) acc = fma((double)Af[(int64_t)i * ld + n], dFdu[i], acc);
  } else {
    for (int i = 0; i < M; i++) acc = fma(A[(int64_t)i * ld + n], dFdu[i], acc);
  }
  r[n] = acc - y[n] / params[0];
}

// The bound scalar from c = WB (u / s2) (sgpr_c_kernel), diag(LB) and the scalars.  One block.
__global__ void __launch_bounds__(256) sgpr_finish_kernel(const double* __restrict__ LB,
                                                          const double* __restrict__ c, int M,
                                                          int N, const double* __restrict__ params, int P,
                                                          const int* __restrict__ toff, const int* __restrict__ ktype,
                                                          const int* __restrict__ km, int reg,
                                                          double* __restrict__ scal) {
  __shared__ double red[256];
  const double s2 = params[0];
  double csq = 0.0, logd = 0.0;
  for (int i = threadIdx.x; i < M; i += 256) {
    const double acc = c[i];
    csq = fma(acc, acc, csq);
    logd += log(LB[(int64_t)i * M + i]);
  }
  red[threadIdx.x] = csq; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  csq = red[0]; __syncthreads();
  red[threadIdx.x] = logd; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  logd = red[0];
  if (threadIdx.x == 0) {
    double kd = 0.0, vabs = 0.0;
    for (int p = 0; p < P; p++) {
      const double* th = params + toff[p];
      double v = th[0];
      vabs += fabs(v);
      if (gp_kern_kdiag_energy(ktype[p])) {
        double s = 0.0;
        for (int q = 0; q < km[p]; q++) s += th[2 + q];
        v *= s;
      }
      kd += v;
    }
    const double LOG2PI = 1.8378770664093453;
    double b = -0.5 * N * LOG2PI;          // sgpr_ss.py:56
    b += -logd;                            // :57  (output_dim = 1)
    b -= 0.5 * N * log(s2);                // :58
    b += -0.5 * scal[1] / s2;              // :59
    b += 0.5 * csq;                        // :60
    b += -0.5 * (N * kd) / s2;             // :61
    b += 0.5 * scal[2] / s2;               // :62  tr(A A^T) with A = A'/sigma
    if (reg) b -= 1000.0 * vabs;           // :64-68
    scal[0] = b;
    scal[3] = kd;
  }
}

// mean[n] = sum_rb dot[rb][n];  var[n] = kd + sum_rb s2[rb][n] - sum_rb s1[rb][n]   (SGPR.build_predict)
// src mode: var[n] = kd - sum_rb s1[rb][n]   (sgpr_ss.py:101)
__global__ void __launch_bounds__(256) predict_finish_kernel(const double* __restrict__ dot, const double* __restrict__ s1,
                                                             const double* __restrict__ s2p, int rb, int n,
                                                             const double* __restrict__ kd, double* __restrict__ mean,
                                                             double* __restrict__ var) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  double d = 0.0, a = 0.0, b = 0.0;
  for (int r = 0; r < rb; r++) {
    d += dot[(int64_t)r * n + j];
    a += s1[(int64_t)r * n + j];
    if (s2p) b += s2p[(int64_t)r * n + j];
  }
  mean[j] = d;
  var[j] = s2p ? (kd[0] + b) - a : kd[0] - a;
}

// ---- host side ----------------------------------------------------------------------------------------
// Each kernel of the sum keeps its own (Z | X) feature table for a whole bound + gradient evaluation: built once in
// the forward pass, read by the Kuu build, the Kuf build and both hyper-gradient contractions.
static size_t sgpr_feat_stride(const gp_sgpr_plan_s* p) {
  return gp_align_up(kernel_build_feat_ws_doubles(p->maxm > 0 ? p->maxm : 1, p->M, p->maxN), 32);
}

static bool sgpr_chol_blocked(const gp_sgpr_plan_s* p) { return p->M > 256 && p->M <= 1024 && (p->M % 2) == 0; }
static size_t sgpr_ws_doubles(const gp_sgpr_plan_s* p) {
  size_t d = 0;
  auto add = [&](size_t c) { d += gp_align_up(c * sizeof(double), 256) / sizeof(double); };
  const size_t M = p->M, strip = gp_strip_doubles(M, p->maxN, p->f32 != 0);
  const int rb = (p->M + 63) / 64;                   // column-partial rows: one per 64-row tile (gemm_wave.hip; the 128-row forms use half)
  for (int i = 0; i < 5; i++) add(M * M);           // L, W, H, LB, WB
  add(strip); add(strip);                            // Kuf, A
  add(sgpr_feat_stride(p) * p->P);                   // one feature table per kernel of the sum
  add((size_t)rb * p->maxN); add((size_t)rb * p->maxN); add((size_t)rb * p->maxN);
  add(M); add(M); add(64); add((size_t)(p->nsplit > 2 ? p->nsplit : 2) * M); add(2 * SG_RED_BLOCKS);
  add((size_t)p->nsplit * M * M);
  for (int i = 0; i < 6; i++) add(M * M);            // E2, T1, T2, Wbar, R, Binv
  add(strip); add(M); add(M); add(M); add(p->maxN);
  {
    const size_t ns = hyper_num_sums(p->maxm);
    add(ns * hyper_kuf_records(p->maxN, (int)M) * (size_t)(p->P > 0 ? p->P : 1));     // one record set per kernel of the sum (fused contraction)
    add(ns * hyper_kuf_records((int)M, (int)M) * (size_t)(p->P > 0 ? p->P : 1));
  }
  if (sgpr_chol_blocked(p)) { add(chol_inverse_blocked_workspace_bytes(p->M) / sizeof(double) + 1); add(chol_inverse_blocked_workspace_bytes(p->M) / sizeof(double) + 1); }
  return d;
}

extern "C" {

gp_status gp_sgpr_create(gp_handle h, const gp_sgpr_config* cfg, gp_sgpr_plan* out) {
  if (!h || !out) return GP_ERR_BAD_ARG;
  *out = nullptr;
  if (!cfg || cfg->num_kernels < 1 || cfg->max_N < 1 || cfg->M < 1 || !cfg->kern_type || !cfg->partials)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_create: bad config");
  gp_sgpr_plan p = new gp_sgpr_plan_s();
  p->h = h; p->P = cfg->num_kernels; p->maxN = cfg->max_N; p->M = cfg->M; p->reg = cfg->reg; p->jitter = cfg->jitter;
  int64_t off = 1;
  for (int i = 0; i < p->P; i++) {
    const int t = cfg->kern_type[i], m = cfg->partials[i];
    const bool sm = gp_kern_has_partials(t);
    if (t < 0 || t > GP_KERN_LAST || (sm && (m < 1 || m > 32)) || (!sm && m != 0)) {
      delete p;
      return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_create: bad kernel config");
    }
    p->ktype.push_back(t); p->m.push_back(m); p->off_theta.push_back(off);
    off += GP_THETA_LEN(m);
    if (m > p->maxm) p->maxm = m;
  }
  p->nparams = off;
  p->nsplit = gemm_nt_nsplit(p->M, p->maxN, 1);
  *out = p;
  return GP_OK;
}

gp_status gp_sgpr_destroy(gp_sgpr_plan p) { delete p; return GP_OK; }
gp_status gp_sgpr_set_precision(gp_sgpr_plan p, int32_t bits) {
  if (!p || (bits != 32 && bits != 64)) return GP_ERR_BAD_ARG;
  if (p->ws) return gp_fail(p->h, GP_ERR_BAD_ARG, "gp_sgpr_set_precision: call it before gp_sgpr_set_workspace");
  p->f32 = (bits == 32);
  return GP_OK;
}
int64_t gp_sgpr_num_params(gp_sgpr_plan p) { return p ? p->nparams : 0; }
size_t gp_sgpr_workspace_bytes(gp_sgpr_plan p) { return p ? sgpr_ws_doubles(p) * sizeof(double) + 2 * SG_DESC_BYTES + 4096 : 0; }

gp_status gp_sgpr_set_workspace(gp_sgpr_plan p, void* workspace, size_t bytes) {
  if (!p) return GP_ERR_BAD_ARG;
  sg_invalidate(p);
  if (!workspace || bytes < gp_sgpr_workspace_bytes(p) || (((uintptr_t)workspace) & 255))
    return gp_fail(p->h, GP_ERR_WORKSPACE, "gp_sgpr_set_workspace: workspace too small or not 256-byte aligned");
  GpArena ar(workspace, bytes);
  const size_t M = p->M, strip = gp_strip_doubles(M, p->maxN, p->f32 != 0);
  const int rb = (p->M + 63) / 64;
  p->d_desc = ar.take<char>(2 * SG_DESC_BYTES);
  p->L = ar.take<double>(M * M); p->W = ar.take<double>(M * M); p->H = ar.take<double>(M * M);
  p->LB = ar.take<double>(M * M); p->WB = ar.take<double>(M * M);
  p->Kuf = ar.take<double>(strip); p->A = ar.take<double>(strip);
  p->feat = ar.take<double>(sgpr_feat_stride(p) * p->P);
  p->s1 = ar.take<double>((size_t)rb * p->maxN); p->s2 = ar.take<double>((size_t)rb * p->maxN);
  p->dot = ar.take<double>((size_t)rb * p->maxN);
  p->u = ar.take<double>(M); p->c = ar.take<double>(M); p->scal = ar.take<double>(64);
  p->upart = ar.take<double>((size_t)(p->nsplit > 2 ? p->nsplit : 2) * M); p->red = ar.take<double>(2 * SG_RED_BLOCKS);
  p->slabs = ar.take<double>((size_t)p->nsplit * M * M);
  p->E2 = ar.take<double>(M * M); p->T1 = ar.take<double>(M * M); p->T2 = ar.take<double>(M * M);
  p->Wbar = ar.take<double>(M * M); p->R = ar.take<double>(M * M); p->Binv = ar.take<double>(M * M);
  p->G = ar.take<double>(strip); p->ubar = ar.take<double>(M); p->Lu = ar.take<double>(M); p->alpha = ar.take<double>(M);
  p->ones = ar.take<double>(p->maxN);
  {
    const size_t ns = hyper_num_sums(p->maxm);
    p->hyp_stride = ns * hyper_kuf_records(p->maxN, (int)M);
    p->hyp = ar.take<double>(p->hyp_stride * (size_t)(p->P > 0 ? p->P : 1));
    p->hyp_uu_stride = ns * hyper_kuf_records((int)M, (int)M);
    p->hyp_uu = ar.take<double>(p->hyp_uu_stride * (size_t)(p->P > 0 ? p->P : 1));
  }
  p->chol_blocked = sgpr_chol_blocked(p);
  if (p->chol_blocked) {
    p->chol_ws_bytes = chol_inverse_blocked_workspace_bytes(p->M);
    p->chol_ws_kuu = ar.take<char>(p->chol_ws_bytes);
    p->chol_ws_b = ar.take<char>(p->chol_ws_bytes);
  }
  if (!ar.ok) return gp_fail(p->h, GP_ERR_WORKSPACE, "gp_sgpr_set_workspace: arena exhausted");
  p->ws = workspace; p->ws_bytes = bytes;
  if (p->chol_blocked) {
    GP_CHECK(chol_inverse_blocked_prepare(p->h, p->L, p->W, p->M, p->M, p->chol_ws_kuu, p->chol_ws_bytes));
    GP_CHECK(chol_inverse_blocked_prepare(p->h, p->LB, p->WB, p->M, p->M, p->chol_ws_b, p->chol_ws_bytes));
  }
  return GP_OK;
}

}  // extern "C"

static DevKern sg_kern(const gp_sgpr_plan_s* p, const double* params, int i) {
  return DevKern{p->ktype[i], p->m[i], params + p->off_theta[i]};
}

// device descriptor upload helper: returns device pointers for up to 6 GemmProblems + 3 int arrays
struct SgDesc { GemmProblem* probs; int* toff; int* ktype; int* km; };
// (`extra`: a blob placed at SG_EXTRA_OFF of the block — the batched gradient tail's item arrays, sgpr_backward)
static const size_t SG_EXTRA_OFF = 6144, SG_TAIL_MAX = 16, SG_FIN_OFF = SG_EXTRA_OFF + 4096;
static gp_status sg_upload(gp_sgpr_plan p, const std::vector<GemmProblem>& probs, SgDesc* out, int slot,
                           const void* extra = nullptr, size_t extra_bytes = 0) {
  const size_t nb = probs.size() * sizeof(GemmProblem);
  const size_t off_int = gp_align_up(20 * sizeof(GemmProblem), 256);
  if (off_int + 3 * 256 * sizeof(int) > SG_EXTRA_OFF || p->P > 256 || probs.size() > 20 || SG_EXTRA_OFF + extra_bytes > SG_FIN_OFF)
    return gp_fail(p->h, GP_ERR_UNSUPPORTED, "sgpr: too many kernels/problems for the descriptor block");
  std::vector<char>& hd = p->h_desc[slot];
  char* dd = p->d_desc + (size_t)slot * SG_DESC_BYTES;
  hd.assign(SG_DESC_BYTES, 0);
  memcpy(hd.data(), probs.data(), nb);
  int* hi = (int*)(hd.data() + off_int);
  for (int i = 0; i < p->P; i++) { hi[i] = (int)p->off_theta[i]; hi[256 + i] = p->ktype[i]; hi[512 + i] = p->m[i]; }
  if (extra && extra_bytes) memcpy(hd.data() + SG_EXTRA_OFF, extra, extra_bytes);
  if (slot == 1 && p->h_tail_fin.size() <= SG_DESC_BYTES - SG_FIN_OFF && !p->h_tail_fin.empty())
    memcpy(hd.data() + SG_FIN_OFF, p->h_tail_fin.data(), p->h_tail_fin.size());   // (keeps the block's last finish items: re-sent with it)
  if (!p->skip_upload)
    GP_HIP_CHECK(p->h, hipMemcpyAsync(dd, hd.data(), SG_DESC_BYTES, hipMemcpyHostToDevice, p->h->stream));
  out->probs = (GemmProblem*)dd;
  out->toff = (int*)(dd + off_int); out->ktype = out->toff + 256; out->km = out->toff + 512;
  return GP_OK;
}

// device pointers of a descriptor block without touching it
static void sg_desc_ptrs(gp_sgpr_plan p, int slot, SgDesc* out) {
  const size_t off_int = gp_align_up(20 * sizeof(GemmProblem), 256);
  char* dd = p->d_desc + (size_t)slot * SG_DESC_BYTES;
  out->probs = (GemmProblem*)dd;
  out->toff = (int*)(dd + off_int); out->ktype = out->toff + 256; out->km = out->toff + 512;
}

// front part over THIS caller's frames (all of them, or one rank's slice of a frame-sharded window):
// L, W, Kuf, A' = W Kuf (+ colsumsq), H = A' A'^T, u = A' y, sum y^2, sum colsumsq
static gp_status sgpr_local(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int N,
                            const double* Z, SgDesc* desc) {
  gp_handle h = p->h;
  const int M = p->M;
  const int f32 = p->f32;
  const int64_t ld = gp_strip_ld(N, f32 != 0);
  const bool wave_a = !f32 && M > 64 && gemm_wave_takes(1, M, N, 1);      // A' = W Kuf in gemm_wave.hip's form: a partial row per 64 rows
  const int rb = wave_a ? M / 64 : gemm_rowblocks(M, 1);
  std::vector<GemmProblem> probs(3);
  memset(probs.data(), 0, probs.size() * sizeof(GemmProblem));
  { GemmProblem& r = probs[0]; r.A = p->W; r.lda = M; r.B = p->Kuf; r.ldb = ld; r.C = p->A; r.ldc = ld; r.M = M; r.N = N; r.K = M; r.o0 = p->s1; }
  { GemmProblem& r = probs[1]; r.A = p->A; r.lda = ld; r.B = p->A; r.ldb = ld; r.C = p->H; r.ldc = M; r.M = M; r.N = M; r.K = N; r.o2 = p->slabs;
    r.v2 = Y; r.o1 = p->upart; r.o0 = p->u; }      // u = A' y fused into the product's first tile column (slab_reduce sums the K-slices)
  { GemmProblem& r = probs[2]; r.A = p->A; r.lda = ld; r.M = M; r.N = N; r.v0 = Y; r.o0 = p->u; r.a_f32 = f32; }
  // Kuu, Kuf: GPflow Add kernel = sum over kern_list (sgpr_ss.py:42-43)
  // (all-Mercer sums of up to eight kernels are built in one pass per matrix: launch_kernel_build_sum)
  std::vector<DevKern> kerns(p->P);
  std::vector<double*> feats(p->P);
  // the 2 P spectral-mixture feature tables (Z and X of every kernel) from ONE launch when the kernels share a padded partial
  // count: ten 4-us launches with their gaps were 0.1 ms of a 3-ms evaluation
  std::vector<FeatItem> fitems;
  bool one_feat_launch = p->P > 1 && (size_t)2 * p->P * sizeof(FeatItem) <= SG_FIN_OFF - SG_EXTRA_OFF;
  for (int i = 0; i < p->P; i++) {
    kerns[i] = sg_kern(p, params, i);
    feats[i] = p->feat + (size_t)i * sgpr_feat_stride(p);
    one_feat_launch = one_feat_launch && gp_kern_is_mercer(kerns[i].type) && sm_mpad(kerns[i].m) == sm_mpad(kerns[0].m);
  }
  if (one_feat_launch)
    for (int i = 0; i < p->P; i++) {
      const int mp = sm_mpad(kerns[i].m);
      fitems.push_back(FeatItem{kerns[i], Z, feats[i], M, 0});
      fitems.push_back(FeatItem{kerns[i], X, feats[i] + gp_align_up((size_t)2 * mp * M, 32), N, 0});
    }
  GP_CHECK(sg_upload(p, probs, desc, 0, fitems.empty() ? nullptr : fitems.data(), fitems.size() * sizeof(FeatItem)));
  if (one_feat_launch)
    GP_CHECK(launch_sm_features_items(h, (const FeatItem*)(p->d_desc + SG_EXTRA_OFF), 2 * p->P, N > M ? N : M, sm_mpad(kerns[0].m), nullptr, 0));
  else
    for (int i = 0; i < p->P; i++) GP_CHECK(launch_sm_features(h, kerns[i], Z, M, X, N, feats[i]));
  {
    gp_status st = GP_OK;
    if (launch_kernel_build_sum(h, kerns.data(), feats.data(), p->P, Z, M, nullptr, M, p->L, M, p->jitter, 0, &st)) GP_CHECK(st);
    else
      for (int i = 0; i < p->P; i++)
        GP_CHECK(launch_kernel_build(h, kerns[i], Z, M, nullptr, M, p->L, M, i > 0, i == 0 ? p->jitter : 0.0, feats[i], 1));
  }
  // Long batches: the factorisation of Kuu (one workgroup, then a chain of small launches: 0.6 ms at M = 512) goes to the
  // handle's helper stream and the Kuf strip build (device-filling, 0.3 ms at N = 65536) runs beside it, as in the Pdgp
  // engine (engine.hip: cond_batch_run); they meet before A' = W Kuf.  Event fork / join only: it records into a hipGraph.
  const bool forked = (N >= 4096) && gp_aux_fork(h);
  auto build_kuf = [&]() -> gp_status {
    gp_status st = GP_OK;
    if (launch_kernel_build_sum(h, kerns.data(), feats.data(), p->P, Z, M, X, N, p->Kuf, ld, 0.0, f32, &st)) return st;
    for (int i = 0; i < p->P; i++)
      GP_CHECK(launch_kernel_build(h, kerns[i], Z, M, X, N, p->Kuf, ld, i > 0, 0.0, feats[i], 1, f32));
    return GP_OK;
  };
  bool kuf_built = false;
  {
    // A workgroup cluster where it takes the shape (chol_cluster.hip): all its 16 workgroups have to be resident to make
    // progress, so it is enqueued FIRST and on the main stream, and the device-filling strip build takes the helper
    // stream behind it (launched the other way round the cluster's workgroups waited for the build to drain: 0.62 ms
    // instead of 0.36).  Otherwise one workgroup + blocked inverse, or one fused launch, on the helper stream.
    gp_status st = GP_OK;
    hipStream_t helper = h->stream;
    if (forked) h->stream = h->main_stream_saved;
    const bool cluster = launch_cholesky_cluster_single(h, p->L, p->W, M, M, 0, &st);
    if (forked) h->stream = helper;
    if (cluster) {
      if (st == GP_OK) st = build_kuf();
      kuf_built = true;
    } else {
      st = p->chol_blocked ? chol_inverse_blocked_run(h, M, M, p->chol_ws_kuu, p->chol_ws_bytes)
                           : launch_cholesky_inverse_single(h, p->L, p->W, M, M);
    }
    if (forked) { gp_status s2 = gp_aux_end(h); if (st == GP_OK) st = s2; }
    GP_CHECK(st);
  }
  if (!kuf_built) GP_CHECK(build_kuf());
  GP_CHECK(gp_aux_join(h));
  { GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = (M > 64); f.role = (M > 64) ? 1 : 0;   /* one row-block either way: the 64-tiles double the workgroups of a window-sized product */ f.timer = GP_TIMER_COND_A; f.epilogue = EPI_STORE | EPI_COLSUMSQ; f.uniform_aligned = 1;   /* one problem, arena buffers, ld = gp_strip_ld */ f.rows64_ok = wave_a ? 1 : 0;
    if (f32) { f.role = 1; GP_CHECK(launch_gemm_f32_role(h, desc->probs + 0, 1, M, N, f)); }
    else GP_CHECK(launch_gemm_batched(h, desc->probs + 0, 1, M, N, f)); }
  hipLaunchKernelGGL(sgpr_sums_partial_kernel, dim3(SG_RED_BLOCKS, 2), dim3(256), 0, h->stream, p->s1, (int64_t)rb * N, Y,
                     (int64_t)N, p->red);
  hipLaunchKernelGGL(sgpr_sums_final_kernel, dim3(1), dim3(128), 0, h->stream, p->red, p->scal);
  if (f32) GP_CHECK(launch_gemm_f32_nt_reduce_batched(h, desc->probs + 1, 1, M, N, p->nsplit, 1, 0, 1.0, 1));
  else GP_CHECK(launch_gemm_nt_reduce_batched(h, desc->probs + 1, 1, M, N, p->nsplit, 1, 0, 1.0, 1));
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// back part on the window-wide H, u, sum y^2, tr(H): B, LB, WB, c and the bound scalar (Ntotal frames)
static gp_status sgpr_global(gp_sgpr_plan p, const double* params, int Ntotal, const SgDesc* desc) {
  gp_handle h = p->h;
  const int M = p->M;
  hipLaunchKernelGGL(sgpr_B_kernel, dim3(64), dim3(256), 0, h->stream, p->H, p->LB, M, params);
  {
    gp_status st = GP_OK;
    if (launch_cholesky_cluster_single(h, p->LB, p->WB, M, M, 0, &st)) GP_CHECK(st);
    else if (p->chol_blocked) GP_CHECK(chol_inverse_blocked_run(h, M, M, p->chol_ws_b, p->chol_ws_bytes));
    else GP_CHECK(launch_cholesky_inverse_single(h, p->LB, p->WB, M, M));
  }
  hipLaunchKernelGGL(sgpr_c_kernel, dim3((M + 3) / 4), dim3(256), 0, h->stream, p->WB, p->u, p->c, M, params);
  hipLaunchKernelGGL(sgpr_finish_kernel, dim3(1), dim3(256), 0, h->stream, p->LB, p->c, M, Ntotal, params,
                     p->P, desc->toff, desc->ktype, desc->km, p->reg, p->scal);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

static gp_status sgpr_common(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int N,
                             const double* Z, SgDesc* desc) {
  GP_CHECK(sgpr_local(p, params, X, Y, N, Z, desc));
  return sgpr_global(p, params, N, desc);
}

// With ubar = WB^T c = Binv u / s:  Bbar = dF/dB = -1/2 Binv - 1/2 ubar ubar^T,  dF/du = ubar / s,
// E2 = 2 dF/dH = (2/s) Bbar + I/s = (I - Binv - ubar ubar^T) / s          (see gp_sgpr_bound_grad)
__global__ void __launch_bounds__(256) sgpr_E2_kernel(const double* __restrict__ Binv, const double* __restrict__ ubar,
                                                      double* __restrict__ E2, int M, const double* __restrict__ s2) {
  const double inv = 1.0 / s2[0];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)M * M; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    E2[idx] = inv * ((i == j ? 1.0 : 0.0) - Binv[idx] - ubar[i] * ubar[j]);
  }
}

// noise-variance gradient and dF/dkd (one block):
//   dF/ds = -tr(Bbar H)/s^2 - ubar.u/s^2 - tr(H)/(2 s^2) - N/(2s) + |y|^2/(2 s^2) + N kd/(2 s^2),
//   tr(Bbar H) = -1/2 tr(Binv H) - 1/2 ubar^T H ubar ;   dF/dkd = -N/(2s)      (ubar = Binv u / s)
// (the two M x M sums come as per-block partials from sgpr_noise_partial_kernel: one block walking M^2 elements with an
// index division each took 0.37 ms at M = 512, a tenth of the whole evaluation)
#define SG_NOISE_BLOCKS 64
__global__ void __launch_bounds__(256) sgpr_noise_partial_kernel(const double* __restrict__ Binv, const double* __restrict__ H,
                                                                 const double* __restrict__ ubar, int M,
                                                                 double* __restrict__ part) {
  __shared__ double red[2][256];
  double t_bh = 0.0, t_uhu = 0.0;
  for (int i = blockIdx.x; i < M; i += gridDim.x) {
    const double ui = ubar[i];
    const double* __restrict__ hr = H + (int64_t)i * M;
    const double* __restrict__ br = Binv + (int64_t)i * M;
    double s_bh = 0.0, s_uh = 0.0;
    for (int j = threadIdx.x; j < M; j += 256) {
      const double h = hr[j];
      s_bh = fma(br[j], h, s_bh);
      s_uh = fma(ubar[j], h, s_uh);
    }
    t_bh += s_bh; t_uhu = fma(ui, s_uh, t_uhu);
  }
  red[0][threadIdx.x] = t_bh; red[1][threadIdx.x] = t_uhu;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) for (int q = 0; q < 2; q++) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = red[0][0]; part[2 * blockIdx.x + 1] = red[1][0]; }
}
__global__ void __launch_bounds__(256) sgpr_noise_grad_kernel(const double* __restrict__ part, int nblocks,
                                                              const double* __restrict__ ubar, const double* __restrict__ u,
                                                              int M, int N, const double* __restrict__ s2,
                                                              double* __restrict__ scal, double* __restrict__ g_noise) {
  __shared__ double red[3][256];
  double t_bh = 0.0, t_uhu = 0.0, t_uu = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) { t_bh += part[2 * b]; t_uhu += part[2 * b + 1]; }
  for (int i = threadIdx.x; i < M; i += 256) t_uu = fma(ubar[i], u[i], t_uu);
  red[0][threadIdx.x] = t_bh; red[1][threadIdx.x] = t_uhu; red[2][threadIdx.x] = t_uu;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) for (int q = 0; q < 3; q++) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double s = s2[0];
    const double trBH = -0.5 * red[0][0] - 0.5 * red[1][0];
    const double trH = scal[2];   // sum colsumsq(A') = tr(H)
    double g = -trBH / (s * s) - red[2][0] / (s * s) - 0.5 * trH / (s * s) - 0.5 * N / s + 0.5 * scal[1] / (s * s) +
               0.5 * N * scal[3] / (s * s);
    g_noise[0] = g;
    scal[4] = -0.5 * N / s;
    scal[5] = g;
  }
}

// d(-1000 sum |v_p|)/dv_p
__global__ void sgpr_reg_grad_kernel(const double* __restrict__ params, double* __restrict__ grad, const int* __restrict__ toff, int P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) { const double v = params[toff[p]]; grad[toff[p]] -= 1000.0 * (v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0)); }
}

// v[i] /= s[0]
__global__ void __launch_bounds__(256) div_scalar_kernel(double* __restrict__ v, int n, const double* __restrict__ s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] /= s[0];
}

__global__ void __launch_bounds__(256) fill_kernel(double* __restrict__ v, int n, double x) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = x;
}

extern "C" {

/* gradient of the collapsed bound w.r.t. [noise_var | theta_0 | ... ] (what TF autodiff hands to L-BFGS-B in
 * SGPRSS.optimize: transcription.py:283, separation.py:298).  Z is a DataHolder (sgpr_ss.py:26): no gradient. */
}  // extern "C"

// backward pass given the forward state; N = this caller's frames, Ntotal = the window's.  The terms that do not
// depend on which frames the caller holds (noise, Kdiag, Kuu side, L1 penalty) are added only when
// include_replicated != 0, so that a sum of the gradient vectors over the ranks of a frame-sharded window is exact.
static gp_status sgpr_backward(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int N, int Ntotal,
                               const double* Z, double* grad, int include_replicated, const SgDesc& d) {
  gp_handle h = p->h;
  const int M = p->M;
  const int f32 = p->f32;
  const int64_t ld = gp_strip_ld(N, f32 != 0);
  GP_HIP_CHECK(h, hipMemsetAsync(grad, 0, (size_t)p->nparams * sizeof(double), h->stream));
  enum { Q_BINV = 0, Q_UBAR, Q_EH, Q_WBAR, Q_LU, Q_RANK1, Q_R, Q_ALPHA, Q_G, Q_T2, Q_LBAR, Q_P, Q_T3, Q_S, Q_COUNT };
  std::vector<GemmProblem> pr(Q_COUNT);
  memset(pr.data(), 0, pr.size() * sizeof(GemmProblem));
  auto sq = [&](int q) -> GemmProblem& { GemmProblem& r = pr[q]; r.M = M; r.N = M; r.K = M; r.lda = M; r.ldb = M; r.ldc = M; return r; };
  { GemmProblem& r = sq(Q_BINV); r.A = p->WB; r.B = p->WB; r.C = p->Binv; }
  { GemmProblem& r = sq(Q_UBAR); r.A = p->WB; r.v0 = p->c; r.o0 = p->ubar; }
  { GemmProblem& r = sq(Q_EH); r.A = p->E2; r.B = p->H; r.C = p->T1; }
  { GemmProblem& r = sq(Q_WBAR); r.A = p->T1; r.B = p->L; r.C = p->Wbar; }
  { GemmProblem& r = sq(Q_LU); r.A = p->L; r.v0 = p->u; r.o0 = p->Lu; }
  { GemmProblem& r = sq(Q_RANK1); r.C = p->Wbar; r.v0 = p->ubar; r.v1 = p->Lu; }
  { GemmProblem& r = sq(Q_R); r.A = p->W; r.B = p->E2; r.C = p->R; }
  { GemmProblem& r = sq(Q_ALPHA); r.A = p->W; r.v0 = p->ubar; r.o0 = p->alpha; }
  { GemmProblem& r = sq(Q_G); r.A = p->R; r.B = p->A; r.ldb = ld; r.N = N; r.v1 = p->ones; r.C = p->G; r.ldc = ld; }
  { GemmProblem& r = sq(Q_T2); r.A = p->W; r.B = p->Wbar; r.C = p->T2; }
  { GemmProblem& r = sq(Q_LBAR); r.A = p->T2; r.B = p->W; r.C = p->T1; }
  { GemmProblem& r = sq(Q_P); r.A = p->L; r.B = p->T1; r.C = p->T2; }
  { GemmProblem& r = sq(Q_T3); r.A = p->W; r.B = p->T2; r.C = p->H; }     // H is free once E2 H and the noise terms are done
  { GemmProblem& r = sq(Q_S); r.A = p->H; r.B = p->W; r.C = p->E2; }
  // The gradient's tail — per kernel of the sum a Kuu-side contraction and two finish launches, 15 launches of a few
  // workgroups for five kernels — as ONE contraction launch over an item array and ONE finish launch (bwd.hip's item forms,
  // as the Pdgp plan uses them) when the kernels share type and partial count and the Kuf side went through the fused pass.
  bool tail_items = include_replicated && p->P >= 1 && (size_t)p->P <= SG_TAIL_MAX && sizeof(HyperItem) * SG_TAIL_MAX <= 4096 &&
                    sizeof(HyperFinishItem) * SG_TAIL_MAX <= SG_DESC_BYTES - SG_FIN_OFF;
  for (int i = 1; i < p->P; i++) if (p->ktype[i] != p->ktype[0] || p->m[i] != p->m[0]) tail_items = false;
  std::vector<HyperItem> uu_items(tail_items ? p->P : 0);
  for (int i = 0; i < (int)uu_items.size(); i++) {
    HyperItem& iu = uu_items[i];
    memset(&iu, 0, sizeof(iu));
    iu.k = sg_kern(p, params, i); iu.x1 = Z; iu.n1 = M; iu.x2 = Z; iu.n2 = M; iu.G = p->E2; iu.ldg = M; iu.symmetric = 1;
    iu.partials = p->hyp_uu + (size_t)i * p->hyp_uu_stride;
    if (gp_kern_is_mercer(p->ktype[i])) { iu.f1 = p->feat + (size_t)i * sgpr_feat_stride(p); iu.f2 = iu.f1; }
  }
  SgDesc d2;
  GP_CHECK(sg_upload(p, pr, &d2, 1, uu_items.data(), uu_items.size() * sizeof(HyperItem)));
  GemmProblem* D = d2.probs;
  char* const d_block1 = (char*)d2.probs;
  GemmFlags f;
  // Binv = WB^T WB ; ubar = WB^T c (= Binv u / s)
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D + Q_BINV, 1, M, M, f));
  GP_CHECK(launch_matvec_batched(h, D + Q_UBAR, 1, M, 1));
  hipLaunchKernelGGL(sgpr_E2_kernel, dim3(64), dim3(256), 0, h->stream, p->Binv, p->ubar, p->E2, M, params);
  {
    const int nb = M < SG_NOISE_BLOCKS ? M : SG_NOISE_BLOCKS;       // partials in the (free by now) split-K slab buffer
    hipLaunchKernelGGL(sgpr_noise_partial_kernel, dim3(nb), dim3(256), 0, h->stream, p->Binv, p->H, p->ubar, M, p->slabs);
    hipLaunchKernelGGL(sgpr_noise_grad_kernel, dim3(1), dim3(256), 0, h->stream, p->slabs, nb, p->ubar, p->u, M, Ntotal,
                       params, p->scal, include_replicated ? grad : p->scal + 6);
  }
  // from here on ubar holds dF/du = ubar / s
  hipLaunchKernelGGL(div_scalar_kernel, dim3((M + 255) / 256), dim3(256), 0, h->stream, p->ubar, M, params);
  // Wbar = tril(E2 H L^T + ubar (L u)^T)
  f = GemmFlags();
  GP_CHECK(launch_gemm_batched(h, D + Q_EH, 1, M, M, f));
  f = GemmFlags(); f.transB = 1; f.triB = TRI_UPPER; f.triC = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D + Q_WBAR, 1, M, M, f));
  GP_CHECK(launch_matvec_batched(h, D + Q_LU, 1, M, 0));
  GP_CHECK(launch_rank1_tril_batched(h, D + Q_RANK1, 1, M));
  // R = W^T E2 ; alpha = W^T ubar ; Kuf_bar = R A' + alpha y^T
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER;
  GP_CHECK(launch_gemm_batched(h, D + Q_R, 1, M, M, f));
  GP_CHECK(launch_matvec_batched(h, D + Q_ALPHA, 1, M, 1));
  // Kuu side (same Cholesky-adjoint chain as the Pdgp backward): five dependent M x M products that end in Kuu_bar (in E2).
  // Nothing on the Kuf side reads what they write (T1, T2, H, E2 — R = W^T E2 is already there), so for long batches the
  // chain runs on the helper stream underneath the Kuf_bar product and its contraction.
  auto kuu_chain = [&]() -> gp_status {
    GemmFlags f;
    f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, D + Q_T2, 1, M, M, f));
    f = GemmFlags(); f.transB = 1; f.triB = TRI_UPPER; f.triC = TRI_LOWER; f.alpha = -1.0;
    GP_CHECK(launch_gemm_batched(h, D + Q_LBAR, 1, M, M, f));
    f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, D + Q_P, 1, M, M, f));
    GP_CHECK(launch_phi_batched(h, D + Q_P, 1, M));
    f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, D + Q_T3, 1, M, M, f));
    f = GemmFlags(); f.triB = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, D + Q_S, 1, M, M, f));
    return GP_OK;
  };
  const bool forked = (N >= 4096) && gp_aux_fork(h);
  if (forked) {
    gp_status st = kuu_chain();
    gp_status s2 = gp_aux_end(h);
    GP_CHECK(st); GP_CHECK(s2);
  }
  hipLaunchKernelGGL(fill_kernel, dim3((N + 255) / 256), dim3(256), 0, h->stream, p->ones, N, 1.0);
  f = GemmFlags(); f.big_tiles = (M > 64); f.scale_mode = 1; f.timer = GP_TIMER_KUF_BAR; f.role = (M > 64) ? 3 : 0; f.uniform_aligned = 1;
  if (f32) { f.role = 3; GP_CHECK(launch_gemm_f32_role(h, D + Q_G, 1, M, N, f)); }
  else GP_CHECK(launch_gemm_batched(h, D + Q_G, 1, M, N, f));
  if (!forked) GP_CHECK(kuu_chain());
  // every kernel of the sum sees the same Kuf_bar / Kuu_bar (K = sum_p K_p)
  // (Kuf side: all-Mercer sums of up to six kernels with at most four partials each go through ONE pass over Kuf_bar)
  bool fused = false;
  int np_fused = 0;
  {
    std::vector<DevKern> kerns(p->P);
    std::vector<double*> feats(p->P), parts(p->P);
    for (int i = 0; i < p->P; i++) {
      kerns[i] = sg_kern(p, params, i);
      feats[i] = p->feat + (size_t)i * sgpr_feat_stride(p);
      parts[i] = p->hyp + (size_t)i * p->hyp_stride;
    }
    gp_status st = GP_OK;
    fused = launch_hyper_contract_sum(h, kerns.data(), feats.data(), parts.data(), p->P, Z, M, X, N, p->G, ld, p->alpha, Y, f32,
                                      &np_fused, &st);
    if (fused) GP_CHECK(st);
  }
  GP_CHECK(gp_aux_join(h));      // Kuu_bar is needed from here on
  if (fused && tail_items) {
    int np_uu = 0;
    GP_CHECK(launch_hyper_contract_items(h, p->ktype[0], p->m[0], (const HyperItem*)(d_block1 + SG_EXTRA_OFF), p->P, M, M, 0, &np_uu));
    std::vector<HyperFinishItem> fin(p->P);
    for (int i = 0; i < p->P; i++) {
      HyperFinishItem& it = fin[i];
      memset(&it, 0, sizeof(it));
      it.k = sg_kern(p, params, i); it.p_uf = p->hyp + (size_t)i * p->hyp_stride; it.np_uf = np_fused;
      it.p_uu = p->hyp_uu + (size_t)i * p->hyp_uu_stride; it.np_uu = np_uu; it.gv_sum = p->scal + 4;
      it.g_theta = grad + p->off_theta[i]; it.n1 = M;
    }
    const size_t bytes = fin.size() * sizeof(HyperFinishItem);
    if (p->h_tail_fin.size() != bytes || memcmp(p->h_tail_fin.data(), fin.data(), bytes) != 0) {
      if (p->skip_upload) return gp_fail(h, GP_ERR_UNSUPPORTED, "sgpr: finish items changed under a recorded launch sequence");
      p->h_tail_fin.assign((const char*)fin.data(), (const char*)fin.data() + bytes);
      GP_HIP_CHECK(h, hipMemcpyAsync(d_block1 + SG_FIN_OFF, p->h_tail_fin.data(), bytes, hipMemcpyHostToDevice, h->stream));
    }
    GP_CHECK(launch_hyper_finish_items(h, (const HyperFinishItem*)(d_block1 + SG_FIN_OFF), p->P, 2 + 2 * p->m[0]));
  } else
  for (int i = 0; i < p->P; i++) {
    DevKern k = sg_kern(p, params, i);
    double* feat = p->feat + (size_t)i * sgpr_feat_stride(p);   // this kernel's (Z | X) features, from the forward pass
    int np_uf = np_fused, np_uu = 0;
    double* hyp_i = fused ? p->hyp + (size_t)i * p->hyp_stride : p->hyp;
    if (!fused)
      GP_CHECK(launch_hyper_contract(h, k, Z, M, X, N, p->G, ld, p->alpha, Y, 0, feat, hyp_i, &np_uf, nullptr, nullptr, 0, f32));
    GP_CHECK(launch_hyper_finish(h, k, hyp_i, np_uf, include_replicated ? p->scal + 4 : nullptr, grad + p->off_theta[i],
                                 nullptr, 0, M, nullptr));
    if (include_replicated) {
      GP_CHECK(launch_hyper_contract(h, k, Z, M, Z, M, p->E2, M, nullptr, nullptr, 1, feat, p->hyp_uu, &np_uu, nullptr));
      GP_CHECK(launch_hyper_finish(h, k, p->hyp_uu, np_uu, nullptr, grad + p->off_theta[i], nullptr, 0, M, nullptr));
    }
  }
  if (p->reg && include_replicated)
    hipLaunchKernelGGL(sgpr_reg_grad_kernel, dim3(1), dim3(256), 0, h->stream, params, grad, d.toff, p->P);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// every launch of one bound + gradient evaluation, in stream order (no host synchronisation inside)
static gp_status sgpr_enqueue_bound_grad(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int N,
                                         const double* Z, double* grad) {
  SgDesc d;
  GP_CHECK(sgpr_common(p, params, X, Y, N, Z, &d));
  return sgpr_backward(p, params, X, Y, N, N, Z, grad, 1, d);
}

// frame-sharded window: exchange layout [H (M x M) | u (M) | sum y^2 | tr(H)]
static size_t sg_xchg_doubles(const gp_sgpr_plan_s* p) { return (size_t)p->M * p->M + p->M + 2; }

extern "C" {

// ---- one window sharded over its frames: each rank holds a slice of (X, Y) ------------------
int64_t gp_sgpr_exchange_doubles(gp_sgpr_plan p) { return p ? (int64_t)sg_xchg_doubles(p) : 0; }

gp_status gp_sgpr_bound_begin(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                              const double* Z, double* exchange) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  sg_invalidate(p);
  p->staged_N = -1;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgpr_bound_begin: workspace not set");
  if (!params || !X || !Y || !Z || !exchange || N < 1 || N > p->maxN) return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_bound_begin: bad argument");
  SgDesc d;
  GP_CHECK(sgpr_local(p, params, X, Y, N, Z, &d));
  const size_t mm = (size_t)p->M * p->M;
  GP_HIP_CHECK(h, hipMemcpyAsync(exchange, p->H, mm * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(exchange + mm, p->u, p->M * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(exchange + mm + p->M, p->scal + 1, 2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  p->staged_N = N; p->staged_params = params; p->staged_X = X;
  return GP_OK;
}

gp_status gp_sgpr_bound_end(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                            int64_t N_total, const double* Z, const double* exchange, double* bound_dev,
                            double* bound_host, double* grad, int32_t include_replicated) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!params || !X || !Y || !Z || !exchange || N_total < N || N_total > 2147483647LL)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_bound_end: bad argument");
  if (p->staged_N != N || p->staged_params != params || p->staged_X != X)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_bound_end: no matching gp_sgpr_bound_begin (same params, X and N required)");
  p->staged_N = -1;
  const size_t mm = (size_t)p->M * p->M;
  GP_HIP_CHECK(h, hipMemcpyAsync(p->H, exchange, mm * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(p->u, exchange + mm, p->M * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(p->scal + 1, exchange + mm + p->M, 2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  SgDesc d;
  sg_desc_ptrs(p, 0, &d);
  GP_CHECK(sgpr_global(p, params, (int)N_total, &d));
  if (grad) { GP_CHECK(sgpr_backward(p, params, X, Y, N, (int)N_total, Z, grad, include_replicated, d)); p->grad_Y = Y; p->grad_N = N; }   // (gp_sgpr_residual_grad: this rank's slice)
  if (bound_dev) GP_HIP_CHECK(h, hipMemcpyAsync(bound_dev, p->scal, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (bound_host) {
    GP_HIP_CHECK(h, hipMemcpyAsync(bound_host, p->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return check_not_pd(h);
  }
  return GP_OK;
}

gp_status gp_sgpr_set_graphs(gp_sgpr_plan p, int32_t enable) {
  if (!p) return GP_ERR_BAD_ARG;
  p->graphs = enable ? 1 : 0;
  if (!enable && p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
  return GP_OK;
}

gp_status gp_sgpr_eval_counts(gp_sgpr_plan p, int64_t* eager, int64_t* captured, int64_t* replayed) {
  if (!p) return GP_ERR_BAD_ARG;
  if (eager) *eager = p->n_eager;
  if (captured) *captured = p->n_captured;
  if (replayed) *replayed = p->n_replayed;
  return GP_OK;
}

// d bound / d err (N doubles on the device) of the evaluation gp_sgpr_bound_grad has just run with the same params / Y / N:
// A'^T (dF/du) - err / sigma^2, from the strip and the dF/du vector that evaluation left in the workspace.
gp_status gp_sgpr_residual_grad(gp_sgpr_plan p, const double* params, const double* Y, int32_t N, double* r_dev) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws || !params || !Y || !r_dev || N < 1 || N > p->maxN) return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_residual_grad: bad argument");
  if (p->grad_Y != Y || p->grad_N != N)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_residual_grad: call it right after gp_sgpr_bound_grad with the same Y and N");
  hipLaunchKernelGGL(sgpr_resid_grad_kernel, dim3((N + 255) / 256), dim3(256), 0, h->stream, p->A, gp_strip_ld(N, p->f32 != 0),
                     p->f32, p->ubar, Y, p->M, N, params, r_dev);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

gp_status gp_sgpr_bound_grad(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                             const double* Z, double* bound_dev, double* bound_host, double* grad) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgpr_bound_grad: workspace not set");
  if (!params || !X || !Y || !Z || !grad || N < 1 || N > p->maxN) return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_bound_grad: bad argument");
  gp_sgpr_plan_s::EvalKey key;
  key.params = params; key.X = X; key.Y = Y; key.Z = Z; key.grad = grad; key.N = N;
  // capture needs a real stream (not the legacy null stream) and no event timers inside the sequence
  const bool can_graph = p->graphs && h->stream != nullptr && !h->timers_on;
  if (can_graph && p->gexec && key == p->graph_key) {
    GP_HIP_CHECK(h, hipGraphLaunch(p->gexec, h->stream));
    p->n_replayed++;
  } else if (can_graph && p->desc_valid && key == p->desc_key) {
    // second evaluation with the same buffers: the descriptors are on the device already, record the launches
    if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    hipGraph_t graph = nullptr;
    GP_HIP_CHECK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    p->skip_upload = true;
    gp_status st = sgpr_enqueue_bound_grad(p, params, X, Y, N, Z, grad);
    p->skip_upload = false;
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (st != GP_OK || e != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      p->graphs = 0;                       // fall back to eager launches for the rest of this plan's life
      p->desc_valid = false;
      GP_CHECK(sgpr_enqueue_bound_grad(p, params, X, Y, N, Z, grad));
      p->desc_key = key; p->desc_valid = true;
      p->n_eager++;
    } else {
      e = hipGraphInstantiate(&p->gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) { p->gexec = nullptr; return gp_fail(h, GP_ERR_HIP, "hipGraphInstantiate failed"); }
      p->graph_key = key;
      GP_HIP_CHECK(h, hipGraphLaunch(p->gexec, h->stream));
      p->n_captured++;
    }
  } else {
    p->desc_valid = false;
    GP_CHECK(sgpr_enqueue_bound_grad(p, params, X, Y, N, Z, grad));
    p->desc_key = key; p->desc_valid = true;
    p->n_eager++;
  }
  p->grad_Y = Y; p->grad_N = N;
  if (bound_dev) GP_HIP_CHECK(h, hipMemcpyAsync(bound_dev, p->scal, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (bound_host) {
    GP_HIP_CHECK(h, hipMemcpyAsync(bound_host, p->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return check_not_pd(h);
  }
  return GP_OK;
}

}  // extern "C"

extern "C" {

gp_status gp_sgpr_bound(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                        const double* Z, double* bound_dev, double* bound_host) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  sg_invalidate(p);
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgpr_bound: workspace not set");
  if (!params || !X || !Y || !Z || N < 1 || N > p->maxN) return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_bound: bad argument");
  SgDesc d;
  GP_CHECK(sgpr_common(p, params, X, Y, N, Z, &d));
  if (bound_dev) GP_HIP_CHECK(h, hipMemcpyAsync(bound_dev, p->scal, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (bound_host) {
    GP_HIP_CHECK(h, hipMemcpyAsync(bound_host, p->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return check_not_pd(h);
  }
  return GP_OK;
}

static gp_status sgpr_predict_f_impl(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                     const double* Z, const double* Xnew, int32_t n, double* mean, double* var, double* cov);

gp_status gp_sgpr_predict_f(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                            const double* Z, const double* Xnew, int32_t n, double* mean, double* var) {
  return sgpr_predict_f_impl(p, params, X, Y, N, Z, Xnew, n, mean, var, nullptr);
}

// full_cov = True of SGPR.build_predict: cov (n x n, row-major) = K_sum(Xnew) + tmp2^T tmp2 - tmp1^T tmp1; `var` still
// receives the diagonal form (it is the finish kernel's output).  float64 plans only.
gp_status gp_sgpr_predict_f_full(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                 const double* Z, const double* Xnew, int32_t n, double* mean, double* var, double* cov) {
  if (!p) return GP_ERR_BAD_ARG;
  if (!cov) return gp_fail(p->h, GP_ERR_BAD_ARG, "gp_sgpr_predict_f_full: cov is null");
  if (p->f32) return gp_fail(p->h, GP_ERR_UNSUPPORTED, "gp_sgpr_predict_f_full: float64 plans only");
  return sgpr_predict_f_impl(p, params, X, Y, N, Z, Xnew, n, mean, var, cov);
}

static gp_status sgpr_predict_f_impl(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                     const double* Z, const double* Xnew, int32_t n, double* mean, double* var, double* cov) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  sg_invalidate(p);
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgpr_predict_f: workspace not set");
  if (!params || !X || !Y || !Z || !Xnew || !mean || !var || N < 1 || N > p->maxN || n < 1 || n > p->maxN)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_predict_f: bad argument");
  SgDesc d;
  GP_CHECK(sgpr_common(p, params, X, Y, N, Z, &d));
  const int M = p->M;
  const int f32 = p->f32;
  const int64_t ld = gp_strip_ld(n, f32 != 0);
  const int rb = gemm_rowblocks(M, 1);
  // tmp1 = W Kus (stored in A; colsumsq -> s1); tmp2 = WB tmp1 (colsumsq -> s2, dot with c -> dot)
  std::vector<GemmProblem> probs(4);
  memset(probs.data(), 0, probs.size() * sizeof(GemmProblem));
  { GemmProblem& r = probs[0]; r.A = p->W; r.lda = M; r.B = p->Kuf; r.ldb = ld; r.C = p->A; r.ldc = ld; r.M = M; r.N = n; r.K = M; r.o0 = p->s1; }
  { GemmProblem& r = probs[1]; r.A = p->WB; r.lda = M; r.B = p->A; r.ldb = ld; r.M = M; r.N = n; r.K = M; r.v0 = p->c; r.o0 = p->s2; r.o1 = p->dot;
    if (cov) { r.C = p->Kuf; r.ldc = ld; } }      // full covariance: tmp2 is kept (Kus is spent once tmp1 exists)
  { GemmProblem& r = probs[2]; r.A = p->A; r.lda = ld; r.B = p->A; r.ldb = ld; r.C = cov; r.ldc = n; r.M = n; r.N = n; r.K = M; }
  { GemmProblem& r = probs[3]; r.A = p->Kuf; r.lda = ld; r.B = p->Kuf; r.ldb = ld; r.C = cov; r.ldc = n; r.M = n; r.N = n; r.K = M; }
  SgDesc d2;
  GP_CHECK(sg_upload(p, probs, &d2, 1));
  for (int i = 0; i < p->P; i++) {
    DevKern k = sg_kern(p, params, i);
    GP_CHECK(launch_kernel_build(h, k, Z, M, Xnew, n, p->Kuf, ld, i > 0, 0.0, p->feat, 0, f32));
  }
  { GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = 1; f.role = 1; f.timer = GP_TIMER_COND_A; f.epilogue = EPI_STORE | EPI_COLSUMSQ;
    if (f32) GP_CHECK(launch_gemm_f32_role(h, d2.probs + 0, 1, M, n, f));
    else GP_CHECK(launch_gemm_batched(h, d2.probs + 0, 1, M, n, f)); }
  { GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = 1; f.role = 1; f.timer = GP_TIMER_COND_A;
    f.epilogue = EPI_COLSUMSQ | EPI_COLDOT | (cov ? EPI_STORE : 0);
    if (f32) GP_CHECK(launch_gemm_f32_role(h, d2.probs + 1, 1, M, n, f));
    else GP_CHECK(launch_gemm_batched(h, d2.probs + 1, 1, M, n, f)); }
  hipLaunchKernelGGL(predict_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, p->dot, p->s1, p->s2, rb, n,
                     p->scal + 3, mean, var);
  GP_HIP_CHECK(h, hipGetLastError());
  if (cov) {
    for (int i = 0; i < p->P; i++)
      GP_CHECK(launch_kernel_build(h, sg_kern(p, params, i), Xnew, n, nullptr, n, cov, n, i > 0, 0.0, p->feat));
    GemmFlags f; f.transA = 1; f.alpha = -1.0; f.beta = 1.0;
    GP_CHECK(launch_gemm_batched(h, d2.probs + 2, 1, n, n, f));
    f = GemmFlags(); f.transA = 1; f.beta = 1.0;
    GP_CHECK(launch_gemm_batched(h, d2.probs + 3, 1, n, n, f));
  }
  return check_not_pd(h);
}

size_t gp_sgpr_predict_source_workspace_bytes(int32_t N, int32_t n) {
  if (N < 1 || n < 1) return 256;
  const size_t ld = ldN64(n);
  const int rb = gemm_rowblocks(N, 1);
  size_t d = 0;
  auto add = [&](size_t c) { d += gp_align_up(c * sizeof(double), 256) / sizeof(double); };
  add((size_t)N * ldN64(N)); add((size_t)N * ldN64(N));     // K -> L, W
  add((size_t)N * ld); add((size_t)N * ld);   // Kx, A
  d += cholesky_large_workspace_bytes(N) / sizeof(double) + 64;
  add(kernel_build_feat_ws_doubles(32, N, n > N ? n : N));
  add((size_t)rb * n); add((size_t)rb * n); add(N); add(64);
  return d * sizeof(double) + SG_DESC_BYTES + 4096;
}

static gp_status sgpr_predict_source_impl(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                          const double* Xnew, int32_t n, double* mean, double* var, double* cov,
                                          void* workspace, size_t workspace_bytes);

gp_status gp_sgpr_predict_source(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                 const double* Xnew, int32_t n, double* mean, double* var, void* workspace,
                                 size_t workspace_bytes) {
  return sgpr_predict_source_impl(p, params, X, Y, N, Xnew, n, mean, var, nullptr, workspace, workspace_bytes);
}

// full_cov = True of SGPRSS.build_predict_source (sgpr_ss.py:95-99): cov [P][n][n], cov_p = K_sum(Xnew) - A_p^T A_p
gp_status gp_sgpr_predict_source_full(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                      const double* Xnew, int32_t n, double* mean, double* var, double* cov,
                                      void* workspace, size_t workspace_bytes) {
  if (!p) return GP_ERR_BAD_ARG;
  if (!cov) return gp_fail(p->h, GP_ERR_BAD_ARG, "gp_sgpr_predict_source_full: cov is null");
  return sgpr_predict_source_impl(p, params, X, Y, N, Xnew, n, mean, var, cov, workspace, workspace_bytes);
}

static gp_status sgpr_predict_source_impl(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                          const double* Xnew, int32_t n, double* mean, double* var, double* cov,
                                          void* workspace, size_t workspace_bytes) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  sg_invalidate(p);
  if (!params || !X || !Y || !Xnew || !mean || !var || N < 1 || n < 1)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgpr_predict_source: bad argument");
  if (!workspace || workspace_bytes < gp_sgpr_predict_source_workspace_bytes(N, n) || (((uintptr_t)workspace) & 255))
    return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgpr_predict_source: workspace too small");
  GpArena ar(workspace, workspace_bytes);
  const int64_t ld = ldN64(n);
  const int rb = gemm_rowblocks(N, 1);
  char* d_desc = ar.take<char>(SG_DESC_BYTES);
  const int64_t ldL = ldN64(N);      // even leading dimension: 16-byte operand loads in the GEMMs
  double* L = ar.take<double>((size_t)N * ldL);
  double* W = ar.take<double>((size_t)N * ldL);
  void* chol_ws = ar.take<char>(cholesky_large_workspace_bytes(N));
  double* Kx = ar.take<double>((size_t)N * ld);
  double* A = ar.take<double>((size_t)N * ld);
  double* feat = ar.take<double>(kernel_build_feat_ws_doubles(32, N, n > N ? n : N));
  double* s1 = ar.take<double>((size_t)rb * n);
  double* dot = ar.take<double>((size_t)rb * n);
  double* V = ar.take<double>(N);
  double* scal = ar.take<double>(64);
  if (!ar.ok) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgpr_predict_source: arena exhausted");
  // K = K_sum(X) + sigma^2 I ; L = chol(K) ; W = L^-1 ; V = W y   (sgpr_ss.py:88-90)
  for (int i = 0; i < p->P; i++)
    GP_CHECK(launch_kernel_build(h, sg_kern(p, params, i), X, N, nullptr, N, L, ldL, i > 0, 0.0, feat));
  hipLaunchKernelGGL(add_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, h->stream, L, N, ldL, params, 1.0, 0.0);
  if (N > 512) {
    // N = 2001 per window: blocked over the GEMM kernels instead of one workgroup (72 ms -> a few ms)
    GP_CHECK(launch_cholesky_large(h, L, W, N, ldL, chol_ws, cholesky_large_workspace_bytes(N)));
  } else {
    GP_CHECK(launch_cholesky_single(h, L, N, ldL));
    GP_CHECK(launch_tri_inverse_single(h, L, W, N, ldL));
  }
  std::vector<GemmProblem> probs(2 + (cov ? p->P : 0));
  memset(probs.data(), 0, probs.size() * sizeof(GemmProblem));
  { GemmProblem& r = probs[0]; r.A = W; r.lda = ldL; r.M = N; r.v0 = Y; r.o0 = V; }
  { GemmProblem& r = probs[1]; r.A = W; r.lda = ldL; r.B = Kx; r.ldb = ld; r.C = A; r.ldc = ld; r.M = N; r.N = n; r.K = N;
    r.v0 = V; r.o0 = s1; r.o1 = dot; }
  for (int i = 0; cov && i < p->P; i++) {
    GemmProblem& r = probs[2 + i];
    r.A = A; r.lda = ld; r.B = A; r.ldb = ld; r.C = cov + (size_t)i * n * n; r.ldc = n; r.M = n; r.N = n; r.K = N;
  }
  if (probs.size() * sizeof(GemmProblem) > SG_DESC_BYTES) return gp_fail(h, GP_ERR_UNSUPPORTED, "gp_sgpr_predict_source_full: too many kernels");
  GP_HIP_CHECK(h, hipMemcpyAsync(d_desc, probs.data(), probs.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // probs is a stack object
  GemmProblem* dp = (GemmProblem*)d_desc;
  GP_CHECK(launch_matvec_batched(h, dp + 0, 1, N, 0));
  // kd = Kdiag of the SUM kernel (sgpr_ss.py:101), computed by the bound's finish kernel formula
  {
    double kd = 0.0;  // host-side: needs theta -> fetch the few scalars (predict_s is not on the training path)
    std::vector<double> th(p->nparams);
    GP_HIP_CHECK(h, hipMemcpyAsync(th.data(), params, p->nparams * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < p->P; i++) {
      const double* t = th.data() + p->off_theta[i];
      double v = t[0];
      if (gp_kern_kdiag_energy(p->ktype[i])) {
        double s = 0.0;
        for (int q = 0; q < p->m[i]; q++) s += t[2 + q];
        v *= s;
      }
      kd += v;
    }
    GP_HIP_CHECK(h, hipMemcpyAsync(scal, &kd, sizeof(double), hipMemcpyHostToDevice, h->stream));
    GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  for (int i = 0; i < p->P; i++) {
    // Kx = K_i(X, Xnew); A = W Kx; mean_i = A^T V; var_i = Kdiag_sum - sum A^2   (sgpr_ss.py:92-103)
    GP_CHECK(launch_kernel_build(h, sg_kern(p, params, i), X, N, Xnew, n, Kx, ld, 0, 0.0, feat));
    GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = 1; f.role = 1; f.timer = GP_TIMER_COND_A;
    f.epilogue = EPI_COLSUMSQ | EPI_COLDOT | (cov ? EPI_STORE : 0);
    GP_CHECK(launch_gemm_batched(h, dp + 1, 1, N, n, f));
    hipLaunchKernelGGL(predict_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, dot, s1,
                       (const double*)nullptr, rb, n, scal, mean + (size_t)i * n, var + (size_t)i * n);
    if (cov) {     // cov_i = K_sum(Xnew) - A^T A
      double* ci = cov + (size_t)i * n * n;
      for (int q = 0; q < p->P; q++)
        GP_CHECK(launch_kernel_build(h, sg_kern(p, params, q), Xnew, n, nullptr, n, ci, n, q > 0, 0.0, feat));
      GemmFlags g; g.transA = 1; g.alpha = -1.0; g.beta = 1.0;
      GP_CHECK(launch_gemm_batched(h, dp + 2 + i, 1, n, n, g));
    }
  }
  GP_HIP_CHECK(h, hipGetLastError());
  return check_not_pd(h);
}

}  // extern "C"
