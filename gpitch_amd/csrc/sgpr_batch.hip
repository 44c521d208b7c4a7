// sgpr_batch.hip — W independent SGPRSS windows evaluated by ONE launch sequence.
//
// The reference fits hundreds of small windows (N = 2001 frames, M ~ 64 inducing points) one after another
// (gpitch/transcription.py:253-288 AMT.optimize, gpitch/separation.py:279-313 SoSp.optimize); every L-BFGS-B
// evaluation of one window is the collapsed bound of gpitch/sgpr_ss.py:29-71 and its gradient: ~50 dependent launches
// of grids that occupy a few CUs each (0.59 ms per evaluation on an MI355X, nearly all of it launch latency).  Windows
// are independent, so this plan carries W of them through the same ~50 launches: every kernel below runs over a
// window index (blockIdx.z / a batch of GemmProblems / item arrays), with per-window parameters, data, workspaces and
// results.  The arithmetic per window is exactly sgpr.hip's (same kernels, same operation order).
//
// All windows share N, M and the kernel structure (number of kernels, their types and partial counts); parameters,
// X, Y, Z differ.  Layout: params [W][nparams], X / Y [W][N], Z [W][M], bound [W], grad [W][nparams], contiguous.
#include "engine.h"
#include <string.h>

struct SgbWin {            // per-window device pointers (device array, indexed by the window slot)
  const double* params; const double* Y;
  double *H, *LB, *WB, *s1, *u, *c, *scal, *E2, *Binv, *ubar, *grad;
};

struct gp_sgprb_plan_s {
  gp_handle h = nullptr;
  int P = 0, N = 0, M = 0, W = 0, reg = 0;
  double jitter = 1e-6;
  std::vector<int> ktype, m;
  std::vector<int64_t> off_theta;
  int64_t nparams = 0;
  int maxm = 0, nsplit = 2, rb = 1;
  void* ws = nullptr; size_t ws_bytes = 0;
  // per-window workspace blocks (window w at base + w * stride)
  double* wsd = nullptr; size_t win_doubles = 0;
  size_t o_L, o_W, o_H, o_LB, o_WB, o_Kuf, o_A, o_G, o_feat, o_s1, o_u, o_c, o_scal, o_slabs, o_E2, o_T1, o_T2, o_Wbar, o_R,
      o_Binv, o_ubar, o_Lu, o_alpha, o_hyp, o_hyp_uu;
  size_t feat_stride = 0;
  double* ones = nullptr;
  int* d_toff = nullptr; int* d_ktype = nullptr; int* d_km = nullptr;
  char* d_desc = nullptr; size_t desc_bytes = 0; std::vector<char> h_desc;
  // descriptor offsets
  size_t off_win = 0, off_ptr_L = 0, off_ptr_W = 0, off_ptr_LB = 0, off_ptr_WB = 0, off_M = 0, off_ld = 0;
  size_t off_feat = 0, off_cov_uu = 0, off_cov_uf = 0, off_hy_uf = 0, off_hy_uu = 0, off_fin = 0;
  enum { F_A = 0, F_H, F_U, Q_BINV, Q_UBAR, Q_EH, Q_WBAR, Q_LU, Q_RANK1, Q_R, Q_ALPHA, Q_G, Q_T2, Q_LBAR, Q_P, Q_T3, Q_S, Q_COUNT };
  size_t off_prob[Q_COUNT] = {0};
  // cache: what the uploaded descriptors describe
  const double *k_params = nullptr, *k_X = nullptr, *k_Y = nullptr, *k_Z = nullptr; double *k_grad = nullptr, *k_bound = nullptr;
  bool desc_valid = false;
  hipGraphExec_t gexec = nullptr; int graphs = 1; int64_t n_eager = 0, n_captured = 0, n_replayed = 0; int graph_count = -1;
  int np_uf = 0, np_uu = 0;
  char* d_pred_desc = nullptr; size_t pred_desc_bytes = 0;    // descriptors of gp_sgprb_predict_f (uploaded per call)
  ~gp_sgprb_plan_s() { if (gexec) (void)hipGraphExecDestroy(gexec); }
};
typedef gp_sgprb_plan_s* gp_sgprb_plan_t;

// ---- window-batched small kernels (grid.x or grid.y = window) -------------------------------------------------------
// scal[1] = sum y^2; scal[2] = sum over [rb][N] column sums of squares (tr H)
__global__ void __launch_bounds__(256) sgb_pre_kernel(const SgbWin* __restrict__ wins, int N, int rb) {
  const SgbWin w = wins[blockIdx.x];
  __shared__ double red[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) a = fma(w.Y[i], w.Y[i], a);
  for (int64_t i = threadIdx.x; i < (int64_t)rb * N; i += 256) b += w.s1[i];
  red[threadIdx.x] = a; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  a = red[0]; __syncthreads();
  red[threadIdx.x] = b; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { w.scal[1] = a; w.scal[2] = red[0]; }
}

// B = H / s2 + I
__global__ void __launch_bounds__(256) sgb_B_kernel(const SgbWin* __restrict__ wins, int M) {
  const SgbWin w = wins[blockIdx.y];
  const double inv = 1.0 / w.params[0];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)M * M; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    w.LB[idx] = w.H[idx] * inv + (i == j ? 1.0 : 0.0);
  }
}

// c = WB (u / s2) and the bound scalar (sgpr_ss.py:56-68); one block per window.  Same arithmetic as sgpr_finish_kernel.
__global__ void __launch_bounds__(256) sgb_finish_kernel(const SgbWin* __restrict__ wins, int M, int N, int P,
                                                         const int* __restrict__ toff, const int* __restrict__ ktype,
                                                         const int* __restrict__ km, int reg, double* __restrict__ bound) {
  const SgbWin w = wins[blockIdx.x];
  __shared__ double red[256];
  const double s2 = w.params[0];
  double csq = 0.0, logd = 0.0;
  for (int i = threadIdx.x; i < M; i += 256) {
    double acc = 0.0;
    for (int k = 0; k <= i; k++) acc = fma(w.WB[(int64_t)i * M + k], w.u[k], acc);
    acc /= s2;
    w.c[i] = acc;
    csq = fma(acc, acc, csq);
    logd += log(w.LB[(int64_t)i * M + i]);
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
      const double* th = w.params + toff[p];
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
    double b = -0.5 * N * LOG2PI;
    b += -logd;
    b -= 0.5 * N * log(s2);
    b += -0.5 * w.scal[1] / s2;
    b += 0.5 * csq;
    b += -0.5 * (N * kd) / s2;
    b += 0.5 * w.scal[2] / s2;
    if (reg) b -= 1000.0 * vabs;
    w.scal[0] = b;
    w.scal[3] = kd;
    bound[blockIdx.x] = b;
  }
}

// E2 = (I - Binv - ubar ubar^T) / s
__global__ void __launch_bounds__(256) sgb_E2_kernel(const SgbWin* __restrict__ wins, int M) {
  const SgbWin w = wins[blockIdx.y];
  const double inv = 1.0 / w.params[0];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)M * M; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    w.E2[idx] = inv * ((i == j ? 1.0 : 0.0) - w.Binv[idx] - w.ubar[i] * w.ubar[j]);
  }
}

// noise-variance gradient and dF/dkd, then ubar <- ubar / s (= dF/du); one block per window (sgpr_noise_grad_kernel +
// div_scalar_kernel of sgpr.hip); grad is zeroed here first (its other entries are accumulated by the finish kernel)
__global__ void __launch_bounds__(256) sgb_noise_grad_kernel(const SgbWin* __restrict__ wins, int M, int N, int64_t nparams,
                                                             int P, const int* __restrict__ toff, int reg) {
  const SgbWin w = wins[blockIdx.x];
  __shared__ double red[3][256];
  double t_bh = 0.0, t_uhu = 0.0, t_uu = 0.0;
  for (int64_t idx = threadIdx.x; idx < (int64_t)M * M; idx += 256) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    const double hh = w.H[idx];
    t_bh = fma(w.Binv[idx], hh, t_bh);
    t_uhu = fma(w.ubar[i] * w.ubar[j], hh, t_uhu);
  }
  for (int i = threadIdx.x; i < M; i += 256) t_uu = fma(w.ubar[i], w.u[i], t_uu);
  red[0][threadIdx.x] = t_bh; red[1][threadIdx.x] = t_uhu; red[2][threadIdx.x] = t_uu;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) for (int q = 0; q < 3; q++) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  const double s = w.params[0];
  for (int64_t i = threadIdx.x; i < nparams; i += 256) w.grad[i] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double trBH = -0.5 * red[0][0] - 0.5 * red[1][0];
    const double trH = w.scal[2];
    const double g = -trBH / (s * s) - red[2][0] / (s * s) - 0.5 * trH / (s * s) - 0.5 * N / s + 0.5 * w.scal[1] / (s * s) +
                     0.5 * N * w.scal[3] / (s * s);
    w.grad[0] = g;
    w.scal[4] = -0.5 * N / s;
    w.scal[5] = g;
    if (reg)       // d(-1000 sum |v_p|)/dv_p
      for (int p = 0; p < P; p++) { const double v = w.params[toff[p]]; w.grad[toff[p]] -= 1000.0 * (v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0)); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < M; i += 256) w.ubar[i] /= s;
}

// ---- host side ----------------------------------------------------------------------------------------------------
static inline int64_t sgb_ld(int N) { return (N + 1) & ~1; }
static size_t sgb_feat_stride(const gp_sgprb_plan_s* p) {
  return gp_align_up(kernel_build_feat_ws_doubles(p->maxm > 0 ? p->maxm : 1, p->M, p->N), 32);
}

static void sgb_layout(gp_sgprb_plan_s* p) {
  size_t d = 0;
  auto add = [&](size_t c) { size_t o = d; d += gp_align_up(c * sizeof(double), 256) / sizeof(double); return o; };
  const size_t M = p->M, ld = sgb_ld(p->N);
  p->rb = gemm_rowblocks(p->M, p->M > 64 ? 1 : 0);
  p->feat_stride = sgb_feat_stride(p);
  p->o_L = add(M * M); p->o_W = add(M * M); p->o_H = add(M * M); p->o_LB = add(M * M); p->o_WB = add(M * M);
  p->o_Kuf = add(M * ld); p->o_A = add(M * ld); p->o_G = add(M * ld);
  p->o_feat = add(p->feat_stride * p->P);
  p->o_s1 = add((size_t)p->rb * p->N); p->o_u = add(M); p->o_c = add(M); p->o_scal = add(64);
  p->o_slabs = add((size_t)p->nsplit * M * M);
  p->o_E2 = add(M * M); p->o_T1 = add(M * M); p->o_T2 = add(M * M); p->o_Wbar = add(M * M); p->o_R = add(M * M);
  p->o_Binv = add(M * M); p->o_ubar = add(M); p->o_Lu = add(M); p->o_alpha = add(M);
  const size_t ns = hyper_num_sums(p->maxm);
  p->o_hyp = add(ns * hyper_kuf_records(p->N, (int)M) * p->P);
  p->o_hyp_uu = add(ns * hyper_kuf_records((int)M, (int)M) * p->P);
  p->win_doubles = d;
}

struct SgbPredWin;
static size_t sgb_pred_desc_bytes(const gp_sgprb_plan_s* p);
static size_t sgb_desc_bytes(const gp_sgprb_plan_s* p) {
  const size_t W = p->W, P = p->P;
  size_t b = 0;
  auto add = [&](size_t c) { b += gp_align_up(c, 256); };
  add(W * sizeof(SgbWin));
  for (int i = 0; i < 4; i++) add(W * sizeof(double*));
  add(W * sizeof(int)); add(W * sizeof(int));
  add(2 * W * P * sizeof(FeatItem));
  add(W * P * sizeof(CovItem)); add(W * P * sizeof(CovItem));
  add(W * P * sizeof(HyperItem)); add(W * P * sizeof(HyperItem));
  add(W * P * sizeof(HyperFinishItem));
  for (int q = 0; q < gp_sgprb_plan_s::Q_COUNT; q++) add(W * sizeof(GemmProblem));
  return b;
}

extern "C" {

gp_status gp_sgprb_create(gp_handle h, const gp_sgpr_config* cfg, int32_t num_windows, gp_sgprb_plan_t* out) {
  if (!h || !out) return GP_ERR_BAD_ARG;
  *out = nullptr;
  if (!cfg || cfg->num_kernels < 1 || cfg->num_kernels > 256 || cfg->max_N < 1 || cfg->M < 1 || !cfg->kern_type ||
      !cfg->partials || num_windows < 1)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgprb_create: bad config");
  if (cfg->M > 256) return gp_fail(h, GP_ERR_UNSUPPORTED, "gp_sgprb_create: window-batched plans take M <= 256 (use gp_sgpr_* for large windows)");
  gp_sgprb_plan_t p = new gp_sgprb_plan_s();
  p->h = h; p->P = cfg->num_kernels; p->N = cfg->max_N; p->M = cfg->M; p->W = num_windows; p->reg = cfg->reg; p->jitter = cfg->jitter;
  int64_t off = 1;
  for (int i = 0; i < p->P; i++) {
    const int t = cfg->kern_type[i], m = cfg->partials[i];
    const bool sm = gp_kern_has_partials(t);
    if (t < 0 || t > GP_KERN_LAST || (sm && (m < 1 || m > 32)) || (!sm && m != 0)) {
      delete p;
      return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgprb_create: bad kernel config");
    }
    p->ktype.push_back(t); p->m.push_back(m); p->off_theta.push_back(off);
    off += GP_THETA_LEN(m);
    if (m > p->maxm) p->maxm = m;
  }
  p->nparams = off;
  p->nsplit = gemm_nt_nsplit(p->M, p->N, p->W);
  sgb_layout(p);
  *out = p;
  return GP_OK;
}

gp_status gp_sgprb_destroy(gp_sgprb_plan_t p) { delete p; return GP_OK; }
int64_t gp_sgprb_num_params(gp_sgprb_plan_t p) { return p ? p->nparams : 0; }
int32_t gp_sgprb_num_windows(gp_sgprb_plan_t p) { return p ? p->W : 0; }

size_t gp_sgprb_workspace_bytes(gp_sgprb_plan_t p) {
  if (!p) return 0;
  return (size_t)p->W * p->win_doubles * sizeof(double) + gp_align_up((size_t)p->N * sizeof(double), 256) + 3 * 1024 +
         sgb_desc_bytes(p) + sgb_pred_desc_bytes(p) + 8192;
}

gp_status gp_sgprb_set_workspace(gp_sgprb_plan_t p, void* workspace, size_t bytes) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!workspace || bytes < gp_sgprb_workspace_bytes(p) || (((uintptr_t)workspace) & 255))
    return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_set_workspace: workspace too small or not 256-byte aligned");
  if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
  GpArena ar(workspace, bytes);
  p->desc_bytes = sgb_desc_bytes(p);
  p->d_desc = ar.take<char>(p->desc_bytes);
  p->pred_desc_bytes = sgb_pred_desc_bytes(p);
  p->d_pred_desc = ar.take<char>(p->pred_desc_bytes);
  p->ones = ar.take<double>(p->N);
  p->d_toff = ar.take<int>(256); p->d_ktype = ar.take<int>(256); p->d_km = ar.take<int>(256);
  p->wsd = ar.take<double>((size_t)p->W * p->win_doubles);
  if (!ar.ok) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_set_workspace: arena exhausted");
  p->ws = workspace; p->ws_bytes = bytes;
  p->desc_valid = false;
  // constants: the all-ones column scale of Kuf_bar = R A', the kernel structure tables
  std::vector<double> ones(p->N, 1.0);
  GP_HIP_CHECK(h, hipMemcpyAsync(p->ones, ones.data(), p->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  std::vector<int> ti(768, 0);
  for (int i = 0; i < p->P; i++) { ti[i] = (int)p->off_theta[i]; ti[256 + i] = p->ktype[i]; ti[512 + i] = p->m[i]; }
  GP_HIP_CHECK(h, hipMemcpyAsync(p->d_toff, ti.data(), 256 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(p->d_ktype, ti.data() + 256, 256 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(p->d_km, ti.data() + 512, 256 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));       // the staging vectors are stack objects
  return GP_OK;
}

}  // extern "C"

// (re)build every descriptor for the given argument pointers; uploaded once per pointer set
static gp_status sgb_upload(gp_sgprb_plan_t p, const double* params, const double* X, const double* Y, const double* Z,
                            double* grad) {
  gp_handle h = p->h;
  const int W = p->W, P = p->P, M = p->M, N = p->N;
  const int64_t ld = sgb_ld(N);
  p->h_desc.assign(p->desc_bytes, 0);
  size_t off = 0;
  auto region = [&](size_t bytes) { size_t o = off; off += gp_align_up(bytes, 256); return o; };
  p->off_win = region(W * sizeof(SgbWin));
  p->off_ptr_L = region(W * sizeof(double*)); p->off_ptr_W = region(W * sizeof(double*));
  p->off_ptr_LB = region(W * sizeof(double*)); p->off_ptr_WB = region(W * sizeof(double*));
  p->off_M = region(W * sizeof(int)); p->off_ld = region(W * sizeof(int));
  p->off_feat = region(2 * (size_t)W * P * sizeof(FeatItem));
  p->off_cov_uu = region((size_t)W * P * sizeof(CovItem)); p->off_cov_uf = region((size_t)W * P * sizeof(CovItem));
  p->off_hy_uf = region((size_t)W * P * sizeof(HyperItem)); p->off_hy_uu = region((size_t)W * P * sizeof(HyperItem));
  p->off_fin = region((size_t)W * P * sizeof(HyperFinishItem));
  for (int q = 0; q < gp_sgprb_plan_s::Q_COUNT; q++) p->off_prob[q] = region(W * sizeof(GemmProblem));
  if (off > p->desc_bytes) return gp_fail(h, GP_ERR_WORKSPACE, "sgprb: descriptor block too small");
  char* hd = p->h_desc.data();
  SgbWin* wins = (SgbWin*)(hd + p->off_win);
  double** pL = (double**)(hd + p->off_ptr_L); double** pW = (double**)(hd + p->off_ptr_W);
  double** pLB = (double**)(hd + p->off_ptr_LB); double** pWB = (double**)(hd + p->off_ptr_WB);
  int* pM = (int*)(hd + p->off_M); int* pld = (int*)(hd + p->off_ld);
  // item arrays are kernel-major: [p][w], so that one launch per kernel of the sum takes a contiguous run of W items
  FeatItem* feat = (FeatItem*)(hd + p->off_feat);
  CovItem* cuu = (CovItem*)(hd + p->off_cov_uu); CovItem* cuf = (CovItem*)(hd + p->off_cov_uf);
  HyperItem* hyf = (HyperItem*)(hd + p->off_hy_uf); HyperItem* hyu = (HyperItem*)(hd + p->off_hy_uu);
  HyperFinishItem* fin = (HyperFinishItem*)(hd + p->off_fin);
  const size_t ns = hyper_num_sums(p->maxm);
  const size_t rec_uf = hyper_kuf_records(N, M), rec_uu = hyper_kuf_records(M, M);
  typedef gp_sgprb_plan_s PL;
  for (int w = 0; w < W; w++) {
    double* b = p->wsd + (size_t)w * p->win_doubles;
    const double* par = params + (size_t)w * p->nparams;
    const double* Xw = X + (size_t)w * N; const double* Yw = Y + (size_t)w * N; const double* Zw = Z + (size_t)w * M;
    double* gw = grad + (size_t)w * p->nparams;
    SgbWin& sw = wins[w];
    sw.params = par; sw.Y = Yw; sw.H = b + p->o_H; sw.LB = b + p->o_LB; sw.WB = b + p->o_WB; sw.s1 = b + p->o_s1;
    sw.u = b + p->o_u; sw.c = b + p->o_c; sw.scal = b + p->o_scal; sw.E2 = b + p->o_E2; sw.Binv = b + p->o_Binv;
    sw.ubar = b + p->o_ubar; sw.grad = gw;
    pL[w] = b + p->o_L; pW[w] = b + p->o_W; pLB[w] = b + p->o_LB; pWB[w] = b + p->o_WB; pM[w] = M; pld[w] = M;
    auto prob = [&](int q) -> GemmProblem& {
      GemmProblem& r = *((GemmProblem*)(hd + p->off_prob[q]) + w);
      memset(&r, 0, sizeof(r));
      r.M = M; r.N = M; r.K = M; r.lda = M; r.ldb = M; r.ldc = M;
      return r;
    };
    double *L = b + p->o_L, *Wm = b + p->o_W, *H = b + p->o_H, *WB = b + p->o_WB, *Kuf = b + p->o_Kuf, *A = b + p->o_A,
           *G = b + p->o_G, *E2 = b + p->o_E2, *T1 = b + p->o_T1, *T2 = b + p->o_T2, *Wbar = b + p->o_Wbar, *R = b + p->o_R,
           *Binv = b + p->o_Binv, *ubar = b + p->o_ubar, *Lu = b + p->o_Lu, *alpha = b + p->o_alpha, *u = b + p->o_u,
           *c = b + p->o_c;
    { GemmProblem& r = prob(PL::F_A); r.A = Wm; r.B = Kuf; r.ldb = ld; r.C = A; r.ldc = ld; r.N = N; r.o0 = b + p->o_s1; }
    { GemmProblem& r = prob(PL::F_H); r.A = A; r.lda = ld; r.B = A; r.ldb = ld; r.C = H; r.K = N; r.o2 = b + p->o_slabs; }
    { GemmProblem& r = prob(PL::F_U); r.A = A; r.lda = ld; r.N = N; r.v0 = Yw; r.o0 = u; }
    { GemmProblem& r = prob(PL::Q_BINV); r.A = WB; r.B = WB; r.C = Binv; }
    { GemmProblem& r = prob(PL::Q_UBAR); r.A = WB; r.v0 = c; r.o0 = ubar; }
    { GemmProblem& r = prob(PL::Q_EH); r.A = E2; r.B = H; r.C = T1; }
    { GemmProblem& r = prob(PL::Q_WBAR); r.A = T1; r.B = L; r.C = Wbar; }
    { GemmProblem& r = prob(PL::Q_LU); r.A = L; r.v0 = u; r.o0 = Lu; }
    { GemmProblem& r = prob(PL::Q_RANK1); r.C = Wbar; r.v0 = ubar; r.v1 = Lu; }
    { GemmProblem& r = prob(PL::Q_R); r.A = Wm; r.B = E2; r.C = R; }
    { GemmProblem& r = prob(PL::Q_ALPHA); r.A = Wm; r.v0 = ubar; r.o0 = alpha; }
    { GemmProblem& r = prob(PL::Q_G); r.A = R; r.B = A; r.ldb = ld; r.N = N; r.v1 = p->ones; r.C = G; r.ldc = ld; }
    { GemmProblem& r = prob(PL::Q_T2); r.A = Wm; r.B = Wbar; r.C = T2; }
    { GemmProblem& r = prob(PL::Q_LBAR); r.A = T2; r.B = Wm; r.C = T1; }
    { GemmProblem& r = prob(PL::Q_P); r.A = L; r.B = T1; r.C = T2; }
    { GemmProblem& r = prob(PL::Q_T3); r.A = Wm; r.B = T2; r.C = H; }
    { GemmProblem& r = prob(PL::Q_S); r.A = H; r.B = Wm; r.C = E2; }
    for (int i = 0; i < P; i++) {
      DevKern k{p->ktype[i], p->m[i], par + p->off_theta[i]};
      double* ft = b + p->o_feat + (size_t)i * p->feat_stride;
      const int mp = sm_mpad(k.m);
      const size_t idx = (size_t)i * W + w;
      feat[idx] = FeatItem{k, Zw, ft, M, 0};
      feat[(size_t)W * P + idx] = FeatItem{k, Xw, ft + gp_align_up((size_t)2 * mp * M, 32), N, 0};
      cov_item_fill(&cuu[idx], k, Zw, M, nullptr, M, L, M, i > 0, i == 0 ? p->jitter : 0.0, ft);
      cov_item_fill(&cuf[idx], k, Zw, M, Xw, N, Kuf, ld, i > 0, 0.0, ft);
      double* hy = b + p->o_hyp + (size_t)i * ns * rec_uf;
      double* hyuu = b + p->o_hyp_uu + (size_t)i * ns * rec_uu;
      const bool mer = gp_kern_is_mercer(k.type);
      HyperItem& a = hyf[idx];
      memset(&a, 0, sizeof(a));
      a.k = k; a.x1 = Zw; a.n1 = M; a.x2 = Xw; a.n2 = N; a.G = G; a.ldg = ld; a.alpha = alpha; a.gm = Yw; a.symmetric = 0;
      a.f1 = mer ? ft : nullptr; a.f2 = mer ? ft + gp_align_up((size_t)2 * mp * M, 32) : nullptr; a.partials = hy;
      HyperItem& c2 = hyu[idx];
      memset(&c2, 0, sizeof(c2));
      c2.k = k; c2.x1 = Zw; c2.n1 = M; c2.x2 = Zw; c2.n2 = M; c2.G = E2; c2.ldg = M; c2.symmetric = 1;
      c2.f1 = mer ? ft : nullptr; c2.f2 = mer ? ft : nullptr; c2.partials = hyuu;
      HyperFinishItem& f = fin[idx];
      memset(&f, 0, sizeof(f));
      f.k = k; f.p_uf = hy; f.p_uu = hyuu; f.gv_sum = b + p->o_scal + 4; f.g_theta = gw + p->off_theta[i]; f.n1 = M;
      // np_uf / np_uu are launch-geometry constants, filled in by the caller before the upload
    }
  }
  return GP_OK;
}

static gp_status sgb_enqueue(gp_sgprb_plan_t p, int count, double* bound_dev, bool with_grad) {
  gp_handle h = p->h;
  const int W = count, P = p->P, M = p->M, N = p->N;
  typedef gp_sgprb_plan_s PL;
  char* dd = p->d_desc;
  const SgbWin* wins = (const SgbWin*)(dd + p->off_win);
  auto D = [&](int q) { return (const GemmProblem*)(dd + p->off_prob[q]); };
  const FeatItem* feat = (const FeatItem*)(dd + p->off_feat);
  const CovItem* cuu = (const CovItem*)(dd + p->off_cov_uu); const CovItem* cuf = (const CovItem*)(dd + p->off_cov_uf);
  // items are kernel-major with stride p->W; a partial batch (count < W) takes the first `count` of each run
  for (int i = 0; i < P; i++) {
    if (gp_kern_is_mercer(p->ktype[i])) {
      const int mp = sm_mpad(p->m[i]);
      GP_CHECK(launch_sm_features_items(h, feat + (size_t)i * p->W, W, M, mp, nullptr, 0));
      GP_CHECK(launch_sm_features_items(h, feat + (size_t)p->W * P + (size_t)i * p->W, W, N, mp, nullptr, 0));
    }
    GP_CHECK(launch_kernel_build_items(h, p->ktype[i], p->m[i], cuu + (size_t)i * p->W, W, M, M, nullptr, 0));
  }
  GP_CHECK(launch_cholesky_inverse_batched(h, (double* const*)(dd + p->off_ptr_L), (double* const*)(dd + p->off_ptr_W),
                                           (const int*)(dd + p->off_M), (const int*)(dd + p->off_ld), W, M));
  for (int i = 0; i < P; i++)
    GP_CHECK(launch_kernel_build_items(h, p->ktype[i], p->m[i], cuf + (size_t)i * p->W, W, M, N, nullptr, 0));
  { GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = (M > 64); f.role = (M > 64) ? 1 : 0; f.timer = GP_TIMER_COND_A;
    f.epilogue = EPI_STORE | EPI_COLSUMSQ;
    GP_CHECK(launch_gemm_batched(h, D(PL::F_A), W, M, N, f)); }
  hipLaunchKernelGGL(sgb_pre_kernel, dim3(W), dim3(256), 0, h->stream, wins, N, p->rb);
  GP_CHECK(launch_gemm_nt_reduce_batched(h, D(PL::F_H), W, M, N, p->nsplit, 1, 0, 1.0));
  GP_CHECK(launch_rowdot_batched(h, D(PL::F_U), W, M));
  hipLaunchKernelGGL(sgb_B_kernel, dim3(16, W), dim3(256), 0, h->stream, wins, M);
  GP_CHECK(launch_cholesky_inverse_batched(h, (double* const*)(dd + p->off_ptr_LB), (double* const*)(dd + p->off_ptr_WB),
                                           (const int*)(dd + p->off_M), (const int*)(dd + p->off_ld), W, M));
  hipLaunchKernelGGL(sgb_finish_kernel, dim3(W), dim3(256), 0, h->stream, wins, M, N, P, p->d_toff, p->d_ktype, p->d_km,
                     p->reg, bound_dev);
  GP_HIP_CHECK(h, hipGetLastError());
  if (!with_grad) return GP_OK;
  // ---- backward (sgpr.hip: sgpr_backward, window-batched) ----
  GemmFlags f;
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_BINV), W, M, M, f));
  GP_CHECK(launch_matvec_batched(h, D(PL::Q_UBAR), W, M, 1));
  hipLaunchKernelGGL(sgb_E2_kernel, dim3(16, W), dim3(256), 0, h->stream, wins, M);
  hipLaunchKernelGGL(sgb_noise_grad_kernel, dim3(W), dim3(256), 0, h->stream, wins, M, N, p->nparams, P, p->d_toff, p->reg);
  f = GemmFlags();
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_EH), W, M, M, f));
  f = GemmFlags(); f.transB = 1; f.triB = TRI_UPPER; f.triC = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_WBAR), W, M, M, f));
  GP_CHECK(launch_matvec_batched(h, D(PL::Q_LU), W, M, 0));
  GP_CHECK(launch_rank1_tril_batched(h, D(PL::Q_RANK1), W, M));
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_R), W, M, M, f));
  GP_CHECK(launch_matvec_batched(h, D(PL::Q_ALPHA), W, M, 1));
  f = GemmFlags(); f.big_tiles = (M > 64); f.scale_mode = 1; f.timer = GP_TIMER_KUF_BAR; f.role = (M > 64) ? 3 : 0;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_G), W, M, N, f));
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_T2), W, M, M, f));
  f = GemmFlags(); f.transB = 1; f.triB = TRI_UPPER; f.triC = TRI_LOWER; f.alpha = -1.0;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_LBAR), W, M, M, f));
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_P), W, M, M, f));
  GP_CHECK(launch_phi_batched(h, D(PL::Q_P), W, M));
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_T3), W, M, M, f));
  f = GemmFlags(); f.triB = TRI_LOWER;
  GP_CHECK(launch_gemm_batched(h, D(PL::Q_S), W, M, M, f));
  const HyperItem* hyf = (const HyperItem*)(dd + p->off_hy_uf);
  const HyperItem* hyu = (const HyperItem*)(dd + p->off_hy_uu);
  const HyperFinishItem* fin = (const HyperFinishItem*)(dd + p->off_fin);
  int maxblocks = 0;
  for (int i = 0; i < P; i++) {
    int a = 0, b2 = 0;
    GP_CHECK(launch_hyper_contract_items(h, p->ktype[i], p->m[i], hyf + (size_t)i * p->W, W, M, N, 0, &a));
    GP_CHECK(launch_hyper_contract_items(h, p->ktype[i], p->m[i], hyu + (size_t)i * p->W, W, M, M, 0, &b2));
    if (a != p->np_uf || b2 != p->np_uu) return gp_fail(h, GP_ERR_HIP, "sgprb: contraction geometry changed");
    const int blocks = 2 + 2 * p->m[i];
    if (blocks > maxblocks) maxblocks = blocks;
  }
  for (int i = 0; i < P; i++) GP_CHECK(launch_hyper_finish_items(h, fin + (size_t)i * p->W, W, maxblocks));
  return GP_OK;
}

extern "C" {

gp_status gp_sgprb_set_graphs(gp_sgprb_plan_t p, int32_t enable) {
  if (!p) return GP_ERR_BAD_ARG;
  p->graphs = enable ? 1 : 0;
  if (!enable && p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
  return GP_OK;
}

gp_status gp_sgprb_eval_counts(gp_sgprb_plan_t p, int64_t* eager, int64_t* captured, int64_t* replayed) {
  if (!p) return GP_ERR_BAD_ARG;
  if (eager) *eager = p->n_eager;
  if (captured) *captured = p->n_captured;
  if (replayed) *replayed = p->n_replayed;
  return GP_OK;
}

/* bound (and gradient when grad != NULL) of the first `count` windows.  Asynchronous: results are on the device when the
 * handle's stream reaches this point. */
gp_status gp_sgprb_bound_grad(gp_sgprb_plan_t p, const double* params, const double* X, const double* Y, const double* Z,
                              int32_t count, double* bound_dev, double* grad) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_bound_grad: workspace not set");
  if (!params || !X || !Y || !Z || !bound_dev || count < 1 || count > p->W)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgprb_bound_grad: bad argument");
  const bool same = p->desc_valid && p->k_params == params && p->k_X == X && p->k_Y == Y && p->k_Z == Z && p->k_grad == grad &&
                    p->k_bound == bound_dev;   // (bound_dev is baked into the captured finish launch)
  if (!same) {
    if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    GP_CHECK(sgb_upload(p, params, X, Y, Z, grad ? grad : (double*)p->wsd));
    // launch geometry of the contractions (what launch_hyper_contract_items will report)
    {
      const int wr_uf = ((int64_t)p->M * p->N >= (1 << 20)) ? 32 : 8, wr_uu = ((int64_t)p->M * p->M >= (1 << 20)) ? 32 : 8;
      p->np_uf = ((p->N + 255) / 256) * ((p->M + wr_uf - 1) / wr_uf);
      p->np_uu = ((p->M + 255) / 256) * ((p->M + wr_uu - 1) / wr_uu);
      HyperFinishItem* fin = (HyperFinishItem*)(p->h_desc.data() + p->off_fin);
      for (size_t i = 0; i < (size_t)p->W * p->P; i++) { fin[i].np_uf = p->np_uf; fin[i].np_uu = p->np_uu; }
    }
    GP_HIP_CHECK(h, hipMemcpyAsync(p->d_desc, p->h_desc.data(), p->desc_bytes, hipMemcpyHostToDevice, h->stream));
    p->k_params = params; p->k_X = X; p->k_Y = Y; p->k_Z = Z; p->k_grad = grad; p->k_bound = bound_dev; p->desc_valid = true;
    GP_CHECK(sgb_enqueue(p, count, bound_dev, grad != nullptr));
    p->n_eager++;
    return GP_OK;
  }
  const bool can_graph = p->graphs && h->stream != nullptr && !h->timers_on;
  if (can_graph && p->gexec && p->graph_count == count) {
    GP_HIP_CHECK(h, hipGraphLaunch(p->gexec, h->stream));
    p->n_replayed++;
    return GP_OK;
  }
  if (can_graph) {
    if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    hipGraph_t graph = nullptr;
    GP_HIP_CHECK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    gp_status st = sgb_enqueue(p, count, bound_dev, grad != nullptr);
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (st == GP_OK && e == hipSuccess && graph) {
      e = hipGraphInstantiate(&p->gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e == hipSuccess) {
        p->graph_count = count;
        GP_HIP_CHECK(h, hipGraphLaunch(p->gexec, h->stream));
        p->n_captured++;
        return GP_OK;
      }
      p->gexec = nullptr;
    } else if (graph) {
      (void)hipGraphDestroy(graph);
    }
    (void)hipGetLastError();
    p->graphs = 0;            // eager launches for the rest of this plan's life
  }
  GP_CHECK(sgb_enqueue(p, count, bound_dev, grad != nullptr));
  p->n_eager++;
  return GP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Predictions of W fitted windows in one launch sequence — the rest of the loop body of SoSp.optimize
// (gpitch/separation.py:300-313: model.predict_f(X_i), model.predict_s(X_i) after every window's optimisation).
//   predict_f : GPflow 0.5 SGPR.build_predict (full_cov = False) from the window's collapsed-bound state
//               (W = chol(Kuu)^-1, WB = chol(B)^-1, c): tmp1 = W Kus, tmp2 = WB tmp1, mean = tmp2^T c,
//               var = Kdiag_sum + sum tmp2^2 - sum tmp1^2
//   predict_s : the EXACT GP of sgpr_ss.py:73-106 on the window's N frames: L = chol(K_sum(X) + s2 I) (N x N, N = 2001 per
//               window), V = L^-1 y, per source A = L^-1 K_p(X, Xnew), mean_p = A^T V, var_p = Kdiag_sum - sum A^2.
// Same kernels and operation order per window as gp_sgpr_predict_f / gp_sgpr_predict_source (sgpr.hip); what changes is
// that every launch carries all windows: one window's predict_s is ~100 dependent launches of small grids (7 ms, nearly
// all latency), W windows cost the arithmetic.
struct SgbPredWin {
  const double* params; double* L; int64_t ldL; double* scal;
  const double* dot; const double* s1; const double* s2; const double* kd;
  double* mean; double* var;
};

// L[i][i] += noise variance; scal[0] = Kdiag of the SUM kernel (sgpr_ss.py:101)
__global__ void __launch_bounds__(256) sgb_pred_prep_kernel(const SgbPredWin* __restrict__ wins, int N, int P,
                                                            const int* __restrict__ toff, const int* __restrict__ ktype,
                                                            const int* __restrict__ km) {
  const SgbPredWin w = wins[blockIdx.x];
  const double s2 = w.params[0];
  for (int i = threadIdx.x; i < N; i += 256) w.L[(int64_t)i * w.ldL + i] += s2;
  if (threadIdx.x == 0) {
    double kd = 0.0;
    for (int p = 0; p < P; p++) {
      const double* th = w.params + toff[p];
      double v = th[0];
      if (gp_kern_kdiag_energy(ktype[p])) {
        double s = 0.0;
        for (int q = 0; q < km[p]; q++) s += th[2 + q];
        v *= s;
      }
      kd += v;
    }
    w.scal[0] = kd;
  }
}

// mean[j] = sum_rb dot[rb][j];  var[j] = kd + sum_rb s2[rb][j] - sum_rb s1[rb][j]  (s2 null: kd - sum s1); grid (n / 256, window)
__global__ void __launch_bounds__(256) sgb_pred_finish_kernel(const SgbPredWin* __restrict__ wins, int rb, int n,
                                                              int64_t out_off) {
  const SgbPredWin w = wins[blockIdx.y];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  double d = 0.0, a = 0.0, b = 0.0;
  for (int r = 0; r < rb; r++) {
    d += w.dot[(int64_t)r * n + j];
    a += w.s1[(int64_t)r * n + j];
    if (w.s2) b += w.s2[(int64_t)r * n + j];
  }
  w.mean[out_off + j] = d;
  w.var[out_off + j] = w.s2 ? (w.kd[0] + b) - a : w.kd[0] - a;
}

static inline int64_t sgb_ld64(int n) { return (n + 1) & ~1; }
static size_t sgb_pred_desc_bytes(const gp_sgprb_plan_s* p) {
  const size_t W = p->W, P = p->P;
  return gp_align_up(W * P * sizeof(FeatItem), 256) + gp_align_up(W * P * sizeof(CovItem), 256) +
         2 * gp_align_up(W * sizeof(GemmProblem), 256) + gp_align_up(W * sizeof(SgbPredWin), 256);
}

extern "C" {

// mean, var: [count][n].  Xnew: [count][n], n <= N.  Runs the forward pass of the bound at `params` first (same descriptors
// and recorded launch sequence as gp_sgprb_bound_grad), then the prediction launches.
gp_status gp_sgprb_predict_f(gp_sgprb_plan_t p, const double* params, const double* X, const double* Y, const double* Z,
                             const double* Xnew, int32_t n, int32_t count, double* mean, double* var) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_predict_f: workspace not set");
  if (!params || !X || !Y || !Z || !Xnew || !mean || !var || count < 1 || count > p->W || n < 1 || n > p->N)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgprb_predict_f: bad argument (1 <= n <= N, 1 <= count <= num_windows)");
  const int W = count, P = p->P, M = p->M, N = p->N;
  const int64_t ld = sgb_ld(N);            // the strips of the plan keep their leading dimension
  const int rb = p->rb;
  // forward state of every window: W, WB, c, scal[3] = Kdiag_sum.  (The bound values nobody asked for land in the first
  // window's G strip, which a forward-only evaluation leaves free and which the prediction overwrites afterwards.)
  GP_CHECK(gp_sgprb_bound_grad(p, params, X, Y, Z, count, p->wsd + p->o_G, nullptr));
  // descriptors of the prediction launches (host-built, uploaded per call: predictions are not in the training loop)
  const size_t nfeat = (size_t)W * P, ncov = (size_t)W * P;
  const size_t bytes = gp_align_up(nfeat * sizeof(FeatItem), 256) + gp_align_up(ncov * sizeof(CovItem), 256) +
                       2 * gp_align_up((size_t)W * sizeof(GemmProblem), 256) + gp_align_up((size_t)W * sizeof(SgbPredWin), 256);
  std::vector<char> hd(bytes, 0);
  size_t off = 0;
  auto region = [&](size_t b) { size_t o = off; off += gp_align_up(b, 256); return o; };
  const size_t o_feat = region(nfeat * sizeof(FeatItem)), o_cov = region(ncov * sizeof(CovItem));
  const size_t o_p1 = region((size_t)W * sizeof(GemmProblem)), o_p2 = region((size_t)W * sizeof(GemmProblem));
  const size_t o_win = region((size_t)W * sizeof(SgbPredWin));
  FeatItem* feat = (FeatItem*)(hd.data() + o_feat);
  CovItem* cov = (CovItem*)(hd.data() + o_cov);
  GemmProblem* p1 = (GemmProblem*)(hd.data() + o_p1);
  GemmProblem* p2 = (GemmProblem*)(hd.data() + o_p2);
  SgbPredWin* wins = (SgbPredWin*)(hd.data() + o_win);
  for (int w = 0; w < W; w++) {
    double* b = p->wsd + (size_t)w * p->win_doubles;
    const double* par = params + (size_t)w * p->nparams;
    const double* Zw = Z + (size_t)w * M;
    const double* Xn = Xnew + (size_t)w * n;
    double *Wm = b + p->o_W, *WB = b + p->o_WB, *Kus = b + p->o_Kuf, *A = b + p->o_A, *c = b + p->o_c;
    double* s1 = b + p->o_s1;                         // [rb][n]
    double* s2 = b + p->o_G;                          // [rb][n]   (G: M x ld doubles, free here)
    double* dot = s2 + (size_t)rb * N;
    for (int i = 0; i < P; i++) {
      DevKern k{p->ktype[i], p->m[i], par + p->off_theta[i]};
      double* ft = b + p->o_feat + (size_t)i * p->feat_stride;     // [Z features | frame features]: the frame half is rebuilt for Xnew
      const int mp = sm_mpad(k.m);
      const size_t idx = (size_t)i * W + w;
      feat[idx] = FeatItem{k, Xn, ft + gp_align_up((size_t)2 * mp * M, 32), n, 0};
      cov_item_fill(&cov[idx], k, Zw, M, Xn, n, Kus, ld, i > 0, 0.0, ft);
    }
    { GemmProblem& r = p1[w]; memset(&r, 0, sizeof(r));
      r.A = Wm; r.lda = M; r.B = Kus; r.ldb = ld; r.C = A; r.ldc = ld; r.M = M; r.N = n; r.K = M; r.o0 = s1; }
    { GemmProblem& r = p2[w]; memset(&r, 0, sizeof(r));
      r.A = WB; r.lda = M; r.B = A; r.ldb = ld; r.M = M; r.N = n; r.K = M; r.v0 = c; r.o0 = s2; r.o1 = dot; }
    SgbPredWin& sw = wins[w];
    memset(&sw, 0, sizeof(sw));
    sw.params = par; sw.dot = dot; sw.s1 = s1; sw.s2 = s2; sw.kd = b + p->o_scal + 3;
    sw.mean = mean + (size_t)w * n; sw.var = var + (size_t)w * n;
  }
  if (bytes > p->pred_desc_bytes) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_predict_f: descriptor block too small");
  char* dd = p->d_pred_desc;
  GP_HIP_CHECK(h, hipMemcpyAsync(dd, hd.data(), bytes, hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));       // hd is a stack object
  const FeatItem* dfeat = (const FeatItem*)(dd + o_feat);
  const CovItem* dcov = (const CovItem*)(dd + o_cov);
  for (int i = 0; i < P; i++) {
    if (gp_kern_is_mercer(p->ktype[i]))
      GP_CHECK(launch_sm_features_items(h, dfeat + (size_t)i * W, W, n, sm_mpad(p->m[i]), nullptr, 0));
    GP_CHECK(launch_kernel_build_items(h, p->ktype[i], p->m[i], dcov + (size_t)i * W, W, M, n, nullptr, 0));
  }
  { GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = (M > 64); f.role = (M > 64) ? 1 : 0; f.timer = GP_TIMER_COND_A;
    f.epilogue = EPI_STORE | EPI_COLSUMSQ;
    GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(dd + o_p1), W, M, n, f)); }
  { GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = (M > 64); f.role = (M > 64) ? 1 : 0; f.timer = GP_TIMER_COND_A;
    f.epilogue = EPI_COLSUMSQ | EPI_COLDOT;
    GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(dd + o_p2), W, M, n, f)); }
  hipLaunchKernelGGL(sgb_pred_finish_kernel, dim3((n + 255) / 256, W), dim3(256), 0, h->stream,
                     (const SgbPredWin*)(dd + o_win), rb, n, (int64_t)0);
  GP_HIP_CHECK(h, hipGetLastError());
  // (the frame halves of the feature tables and the Kuf strips now hold Xnew's: every evaluation rebuilds them first)
  return check_not_pd(h);
}

size_t gp_sgprb_predict_source_workspace_bytes(gp_sgprb_plan_t p, int32_t count, int32_t n) {
  if (!p || count < 1 || n < 1) return 256;
  const int N = p->N;
  const size_t ldL = sgb_ld64(N), ld = sgb_ld64(n);
  const int rb = gemm_rowblocks(N, 1);
  size_t d = 0;
  auto add = [&](size_t c) { d += gp_align_up(c * sizeof(double), 256) / sizeof(double); };
  add((size_t)N * ldL); add((size_t)N * ldL);     // K -> L, W = L^-1
  add((size_t)N * ld);                            // K_p(X, Xnew)
  add(kernel_build_feat_ws_doubles(p->maxm > 0 ? p->maxm : 1, N, n > N ? n : N));
  add((size_t)rb * n); add((size_t)rb * n); add(N); add(64);
  const size_t C = (size_t)count, P = p->P;
  size_t desc = 2 * gp_align_up(C * P * sizeof(FeatItem), 256) + 2 * gp_align_up(C * P * sizeof(CovItem), 256) +
                2 * gp_align_up(C * sizeof(GemmProblem), 256) + gp_align_up(C * sizeof(SgbPredWin), 256) +
                2 * gp_align_up(C * sizeof(double*), 256) + 2 * gp_align_up(C * sizeof(int), 256);
  return C * d * sizeof(double) + desc + cholesky_large_batched_workspace_bytes(N, count) + 8192;
}

// mean, var: [count][P][n] (window-major, then source).  Xnew: [count][n].
gp_status gp_sgprb_predict_source(gp_sgprb_plan_t p, const double* params, const double* X, const double* Y,
                                  const double* Xnew, int32_t n, int32_t count, double* mean, double* var, void* workspace,
                                  size_t workspace_bytes) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_predict_source: plan workspace not set");
  if (!params || !X || !Y || !Xnew || !mean || !var || count < 1 || n < 1)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_sgprb_predict_source: bad argument");
  if (!workspace || workspace_bytes < gp_sgprb_predict_source_workspace_bytes(p, count, n) || (((uintptr_t)workspace) & 255))
    return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_predict_source: workspace too small or not 256-byte aligned");
  const int W = count, P = p->P, N = p->N;
  const int64_t ldL = sgb_ld64(N), ld = sgb_ld64(n);
  const int rb = gemm_rowblocks(N, 1);
  GpArena ar(workspace, workspace_bytes);
  const size_t C = (size_t)W;
  // descriptor block
  const size_t b_feat = gp_align_up(C * P * sizeof(FeatItem), 256), b_cov = gp_align_up(C * P * sizeof(CovItem), 256),
               b_prob = gp_align_up(C * sizeof(GemmProblem), 256), b_win = gp_align_up(C * sizeof(SgbPredWin), 256),
               b_ptr = gp_align_up(C * sizeof(double*), 256), b_int = gp_align_up(C * sizeof(int), 256);
  const size_t desc_bytes = 2 * b_feat + 2 * b_cov + 2 * b_prob + b_win + 2 * b_ptr + 2 * b_int;
  char* dd = ar.take<char>(desc_bytes);
  void* chol_ws = ar.take<char>(cholesky_large_batched_workspace_bytes(N, W));
  const size_t featd = gp_align_up(kernel_build_feat_ws_doubles(p->maxm > 0 ? p->maxm : 1, N, n > N ? n : N) * sizeof(double), 256) / sizeof(double);
  std::vector<double*> hL(W), hW(W);
  std::vector<double*> hKx(W), hfeat(W), hs1(W), hdot(W), hV(W), hscal(W);
  for (int w = 0; w < W; w++) {
    hL[w] = ar.take<double>((size_t)N * ldL); hW[w] = ar.take<double>((size_t)N * ldL);
    hKx[w] = ar.take<double>((size_t)N * ld); hfeat[w] = ar.take<double>(featd);
    hs1[w] = ar.take<double>((size_t)rb * n); hdot[w] = ar.take<double>((size_t)rb * n);
    hV[w] = ar.take<double>(N); hscal[w] = ar.take<double>(64);
  }
  if (!ar.ok) return gp_fail(h, GP_ERR_WORKSPACE, "gp_sgprb_predict_source: arena exhausted");
  std::vector<char> hd(desc_bytes, 0);
  size_t off = 0;
  auto region = [&](size_t b) { size_t o = off; off += b; return o; };
  const size_t o_fx = region(b_feat), o_fn = region(b_feat), o_ck = region(b_cov), o_cx = region(b_cov),
               o_pv = region(b_prob), o_pa = region(b_prob), o_win = region(b_win), o_pL = region(b_ptr), o_pW = region(b_ptr),
               o_iM = region(b_int), o_ild = region(b_int);
  FeatItem* fx = (FeatItem*)(hd.data() + o_fx); FeatItem* fn = (FeatItem*)(hd.data() + o_fn);
  CovItem* ck = (CovItem*)(hd.data() + o_ck); CovItem* cx = (CovItem*)(hd.data() + o_cx);
  GemmProblem* pv = (GemmProblem*)(hd.data() + o_pv); GemmProblem* pa = (GemmProblem*)(hd.data() + o_pa);
  SgbPredWin* wins = (SgbPredWin*)(hd.data() + o_win);
  double** pL = (double**)(hd.data() + o_pL); double** pW = (double**)(hd.data() + o_pW);
  int* iM = (int*)(hd.data() + o_iM); int* ild = (int*)(hd.data() + o_ild);
  for (int w = 0; w < W; w++) {
    const double* par = params + (size_t)w * p->nparams;
    const double* Xw = X + (size_t)w * N; const double* Yw = Y + (size_t)w * N; const double* Xn = Xnew + (size_t)w * n;
    for (int i = 0; i < P; i++) {
      DevKern k{p->ktype[i], p->m[i], par + p->off_theta[i]};
      const int mp = sm_mpad(k.m);
      const size_t idx = (size_t)i * W + w;
      double* ft = hfeat[w];         // one table per window, rebuilt for every kernel of the sum (launches are ordered)
      fx[idx] = FeatItem{k, Xw, ft, N, 0};
      fn[idx] = FeatItem{k, Xn, ft + gp_align_up((size_t)2 * mp * N, 32), n, 0};
      cov_item_fill(&ck[idx], k, Xw, N, nullptr, N, hL[w], ldL, i > 0, 0.0, ft);
      cov_item_fill(&cx[idx], k, Xw, N, Xn, n, hKx[w], ld, 0, 0.0, ft);
    }
    { GemmProblem& r = pv[w]; memset(&r, 0, sizeof(r)); r.A = hW[w]; r.lda = ldL; r.M = N; r.v0 = Yw; r.o0 = hV[w]; }
    { GemmProblem& r = pa[w]; memset(&r, 0, sizeof(r));
      r.A = hW[w]; r.lda = ldL; r.B = hKx[w]; r.ldb = ld; r.M = N; r.N = n; r.K = N; r.v0 = hV[w]; r.o0 = hs1[w]; r.o1 = hdot[w]; }
    SgbPredWin& sw = wins[w];
    memset(&sw, 0, sizeof(sw));
    sw.params = par; sw.L = hL[w]; sw.ldL = ldL; sw.scal = hscal[w]; sw.dot = hdot[w]; sw.s1 = hs1[w]; sw.s2 = nullptr;
    sw.kd = hscal[w]; sw.mean = mean + (size_t)w * P * n; sw.var = var + (size_t)w * P * n;
    pL[w] = hL[w]; pW[w] = hW[w]; iM[w] = N; ild[w] = (int)ldL;
  }
  GP_HIP_CHECK(h, hipMemcpyAsync(dd, hd.data(), desc_bytes, hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));       // hd is a stack object
  const FeatItem* dfx = (const FeatItem*)(dd + o_fx); const FeatItem* dfn = (const FeatItem*)(dd + o_fn);
  const CovItem* dck = (const CovItem*)(dd + o_ck); const CovItem* dcx = (const CovItem*)(dd + o_cx);
  const SgbPredWin* dwins = (const SgbPredWin*)(dd + o_win);
  // K = K_sum(X) + s2 I ; L = chol(K) ; W = L^-1 ; V = W y   (sgpr_ss.py:88-90)
  for (int i = 0; i < P; i++) {
    if (gp_kern_is_mercer(p->ktype[i]))
      GP_CHECK(launch_sm_features_items(h, dfx + (size_t)i * W, W, N, sm_mpad(p->m[i]), nullptr, 0));
    GP_CHECK(launch_kernel_build_items(h, p->ktype[i], p->m[i], dck + (size_t)i * W, W, N, N, nullptr, 0));
  }
  hipLaunchKernelGGL(sgb_pred_prep_kernel, dim3(W), dim3(256), 0, h->stream, dwins, N, P, p->d_toff, p->d_ktype, p->d_km);
  GP_HIP_CHECK(h, hipGetLastError());
  if (N > 512) {
    GP_CHECK(launch_cholesky_large_batched(h, hL.data(), hW.data(), W, N, ldL, chol_ws, cholesky_large_batched_workspace_bytes(N, W)));
  } else {
    GP_CHECK(launch_cholesky_batched(h, (double* const*)(dd + o_pL), (const int*)(dd + o_iM), (const int*)(dd + o_ild), W, N, 0));
    GP_CHECK(launch_tri_inverse_batched(h, (const double* const*)(dd + o_pL), (double* const*)(dd + o_pW), (const int*)(dd + o_iM),
                                        (const int*)(dd + o_ild), W));
  }
  GP_CHECK(launch_matvec_batched(h, (const GemmProblem*)(dd + o_pv), W, N, 0));
  for (int i = 0; i < P; i++) {
    // Kx = K_i(X, Xnew); A = W Kx (never stored); mean_i = A^T V; var_i = Kdiag_sum - sum A^2   (sgpr_ss.py:92-103)
    if (gp_kern_is_mercer(p->ktype[i])) {
      GP_CHECK(launch_sm_features_items(h, dfx + (size_t)i * W, W, N, sm_mpad(p->m[i]), nullptr, 0));
      GP_CHECK(launch_sm_features_items(h, dfn + (size_t)i * W, W, n, sm_mpad(p->m[i]), nullptr, 0));
    }
    GP_CHECK(launch_kernel_build_items(h, p->ktype[i], p->m[i], dcx + (size_t)i * W, W, N, n, nullptr, 0));
    GemmFlags f; f.triA = TRI_LOWER; f.big_tiles = 1; f.role = 1; f.timer = GP_TIMER_COND_A;
    f.epilogue = EPI_COLSUMSQ | EPI_COLDOT;
    GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(dd + o_pa), W, N, n, f));
    hipLaunchKernelGGL(sgb_pred_finish_kernel, dim3((n + 255) / 256, W), dim3(256), 0, h->stream, dwins, rb, n, (int64_t)i * n);
    GP_HIP_CHECK(h, hipGetLastError());
  }
  return check_not_pd(h);
}

}  // extern "C"
