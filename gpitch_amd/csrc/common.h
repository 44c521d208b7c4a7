// common.h — shared host/device declarations for libgpitch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/gpitch_abi.h"
#include "switches.h"

#define GP_WAVE 64

// gpflow.transforms.Logistic(a, b) bounds registered on a handle (transform code = 3 + index)
#define GP_MAX_LOGISTIC 8
struct GpLogisticTable { double a[GP_MAX_LOGISTIC]; double b[GP_MAX_LOGISTIC]; };

struct gp_handle_s {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string last_error;
  int32_t not_pd_index = -1;
  int32_t* d_status = nullptr;   // device int[4]: {not_pd_flag, pivot_index, gp_index, spare}
  int num_cus = 256;
  GpLogisticTable logistic = {}; int num_logistic = 0;
  // helper stream for work that can overlap the main stream (the latency-bound Kuu factorisation runs on ~24 CUs
  // while the Kuf builds stream over the rest): created on first use, joined through events
  hipStream_t aux_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr, ev_era = nullptr, ev_kuu = nullptr, ev_diag = nullptr;
  hipStream_t main_stream_saved = nullptr; bool aux_active = false, aux_pending = false;
  // second helper stream ("side"): one more independent piece of a step (the spectral-mixture Kuf-side contractions
  // underneath the second half of the Kuf_bar product); gp_side_begin / _end / _join
  hipStream_t side_stream = nullptr; hipEvent_t ev_side_fork = nullptr, ev_side_join = nullptr;
  hipStream_t side_saved = nullptr; bool side_active = false, side_pending = false;
  // gp_poll_not_pd: pinned landing zone of the status word, event behind its copy
  int32_t* poll_host = nullptr; hipEvent_t ev_poll = nullptr; bool poll_inflight = false;
  // timers
  bool timers_on = false;
  struct TimerRec { hipEvent_t e0, e1; int which; };
  std::vector<TimerRec> pending;
  std::vector<hipEvent_t> event_pool;
  double timer_ms[GP_TIMER_COUNT] = {0};
  int64_t timer_n[GP_TIMER_COUNT] = {0};
};

#define GP_HIP_CHECK(h, expr)                                                                  \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      char _b[512];                                                                            \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
               __LINE__);                                                                      \
      (h)->last_error = _b;                                                                    \
      return GP_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

#define GP_CHECK(expr)                   \
  do {                                   \
    gp_status _s = (expr);               \
    if (_s != GP_OK) return _s;          \
  } while (0)

static inline gp_status gp_fail(gp_handle h, gp_status s, const char* msg) {
  if (h) h->last_error = msg;
  return s;
}

// scoped timer for a kernel class; records HIP events on the handle's stream when enabled
// ---- lean device math for the covariance kernels -------------------------------------------------------------
// The M x N covariance builds and their gradient contraction are bound by float64 VALU issue, not by HBM (about 57
// instructions per entry with the library sqrt / exp).  Both functions below drop the range handling the arguments
// cannot need and stay within 1.5 ulp, far inside the 1e-9 parity bar.
// sqrt(x) for 1e-13 <= x <= 1e300 (a squared distance + 1e-12): v_rsq_f64, one Goldschmidt step and ONE residual
// correction, without the library's input scaling and class checks.  The library's second correction changes nothing
// here: 4.2 M arguments spread over [1e-12, 1e4] all come out correctly rounded after the first (tools/probe_f64_ops.hip
// has the instruction costs: v_rsq_f64 and v_max_f64 are quarter-rate, 17 and 14 cycles per wavefront against 4.75 for
// v_fma_f64).
__device__ __forceinline__ double gp_sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  const double d = fma(-g, g, x);
  return fma(d, h, g);
}
// sqrt(x) and 1/sqrt(x) together (same argument range): the reciprocal costs one more correction step instead of a
// float64 division (~20 instructions) wherever a kernel needs both r and something / r.
__device__ __forceinline__ void gp_sqrt_rsqrt_pos(double x, double& s, double& rinv) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  double r = h + h;
  r = fma(fma(-g, r, 1.0), r, r);
  s = g; rinv = r;
}
// table for gp_exp_neg: tab[j] = 2^(j/64), j < 64 (filled per workgroup into LDS by gp_exp_tab_init)
#define GP_EXP_TAB 64
__device__ __forceinline__ void gp_exp_tab_init(double* tab) {
  if (threadIdx.x < GP_EXP_TAB) tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
}
// exp(x) for x <= 0: x = (64 q + j) ln2/64 + r, |r| <= ln2/128 -> 2^q * tab[j] * P5(r); x < -744 returns 0.
__device__ __forceinline__ double gp_exp_neg(double x, const double* __restrict__ tab) {
  x = fmax(x, -744.0);
  const double n = __builtin_rint(x * 92.33248261689366);             // 64 / ln2
  double r = fma(n, -0x1.62e42fe000000p-7, x);                         // ln2/64, high 29 bits (n * hi is exact)
  r = fma(n, -0x1.f473de6af278fp-36, r);                               // ln2/64, low part
  const int ni = (int)n;
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(tab[ni & 63] * p, ni >> 6);
}

// kernel families (include/gpitch_abi.h gp_kernel_type)
#define GP_KERN_LAST GP_KERN_MERCER_MATERN52SM
static inline __host__ __device__ bool gp_kern_has_partials(int t) { return t >= GP_KERN_MERCER_MATERN12SM && t <= GP_KERN_LAST; }
// feature ("Mercer") form: K = var * env(r) * Phi(x)^T Phi(x'), r = euclid_dist
static inline __host__ __device__ bool gp_kern_is_mercer(int t) { return t == GP_KERN_MERCER_MATERN12SM || t == GP_KERN_MERCER_MATERN52SM; }
// broadcast form: r = |x - x' + 1e-12|, m cosines per entry
static inline __host__ __device__ bool gp_kern_is_broadcast(int t) { return t == GP_KERN_MATERN12SM || t == GP_KERN_MATERN32SM; }
// Kdiag = variance * sum_k energy_k (true) or just variance (false)
static inline __host__ __device__ bool gp_kern_kdiag_energy(int t) { return gp_kern_has_partials(t) && t != GP_KERN_MERCER_MATERN52SM; }

// Fork / join of the handle's helper stream (see gp_handle_s::aux_stream).  Between gp_aux_fork and gp_aux_end the
// launchers (which enqueue on h->stream) target the helper stream; gp_aux_join makes the main stream wait for it.
// gp_aux_fork returns false (and changes nothing) when no helper stream can be had: the work then stays in line.
bool gp_aux_fork(gp_handle h);
// after gp_aux_end and before gp_aux_join: hand the helper stream more work that depends on what the main stream has
// enqueued so far (returns false when there is no pending fork)
bool gp_aux_resume(gp_handle h);
gp_status gp_aux_end(gp_handle h);
gp_status gp_aux_join(gp_handle h);
// the same for the second helper stream: between gp_side_begin (true on success) and gp_side_end the launchers target it;
// the work there starts after everything enqueued on the current stream so far; gp_side_join makes the current stream wait
bool gp_side_begin(gp_handle h);
gp_status gp_side_end(gp_handle h);
gp_status gp_side_join(gp_handle h);

struct GpTimerScope {
  gp_handle h;
  int which;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  GpTimerScope(gp_handle h_, int which_);
  ~GpTimerScope();
};

static inline size_t gp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// simple bump allocator over a caller-provided workspace
struct GpArena {
  char* base = nullptr;
  size_t size = 0, off = 0;
  bool ok = true;
  GpArena(void* p, size_t n) : base((char*)p), size(n) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = gp_align_up(count * sizeof(T), 256);
    if (off + bytes > size) { ok = false; return nullptr; }
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
};

// ---------------------------------------------------------------------------------------------
// device-side kernel descriptor (passed by value in kernel args)
struct DevKern {
  int type;
  int m;
  const double* theta;  // [variance, lengthscales, e_0.., f_0..]
};
static inline DevKern dev_kern(const gp_kernel_desc* k) { return DevKern{k->type, k->num_partials, k->theta}; }

// ---------------------------------------------------------------------------------------------
// launchers implemented across the .hip files (all enqueue on h->stream)

// cov.hip
// f32out != 0: `out` is a float32 matrix (ld in floats) — the strips of a float32 plan; the arithmetic stays float64
gp_status launch_kernel_build(gp_handle h, DevKern k, const double* x1, int n1, const double* x2, int n2,
                              double* out, int64_t ld, int accumulate, double diag_add, double* feat_ws, int feat_ready = 0,
                              int f32out = 0);
size_t kernel_build_feat_ws_doubles(int m, int n1, int n2);
int sm_mpad(int m);  // spectral-mixture partial count padded to a multiple of 4 (feature tables are zero-padded)
gp_status launch_kernel_diag(gp_handle h, DevKern k, int n, double* out, int accumulate);
gp_status launch_sm_features(gp_handle h, DevKern k, const double* x1, int n1, const double* x2, int n2, double* feat_ws);

// chol.hip
gp_status launch_cholesky_batched(gp_handle h, double* const* d_mats, const int* d_M, const int* d_ld, int batch,
                                  int maxM, int pivot_base = 0);
gp_status launch_cholesky_single(gp_handle h, double* A, int M, int64_t ld, int pivot_base = 0);
// one large matrix, blocked over the GEMM kernels: A -> L in place (lower), W = L^-1 (upper part zero); ld even
size_t cholesky_large_workspace_bytes(int N);
size_t cholesky_large_batched_workspace_bytes(int N, int count);
// one M x M matrix, 128 < M <= 1024: factor from one one-workgroup launch, inverse blocked over 128-column panels
// (descriptors prepared once for fixed buffers; run = launches only)
size_t chol_inverse_blocked_workspace_bytes(int M);
gp_status chol_inverse_blocked_prepare(gp_handle h, double* A, double* W, int M, int64_t ld, void* ws, size_t ws_bytes);
gp_status chol_inverse_blocked_run(gp_handle h, int M, int64_t ld, void* ws, size_t ws_bytes);
// host arrays of `count` device pointers: every launch of the blocked factorisation + inverse runs over all matrices
gp_status launch_cholesky_large_batched(gp_handle h, double* const* A, double* const* W, int count, int N, int64_t ld,
                                        void* ws, size_t ws_bytes);
gp_status launch_cholesky_large(gp_handle h, double* A, double* W, int N, int64_t ld, void* ws, size_t ws_bytes);
gp_status launch_tri_inverse_single(gp_handle h, const double* L, double* Linv, int M, int64_t ld);
// Cholesky factor (in place) and its inverse in one launch, one resident workgroup per matrix
gp_status launch_cholesky_inverse_batched(gp_handle h, double* const* d_mats, double* const* d_W, const int* d_M,
                                          const int* d_ld, int batch, int maxM);
gp_status launch_cholesky_inverse_single(gp_handle h, double* A, double* W, int M, int64_t ld);
gp_status launch_zero_upper_blocks_batched(gp_handle h, double* const* d_mats, const int* d_M, const int* d_ld, int batch,
                                           int maxM, int nb);
gp_status check_not_pd(gp_handle h);  // syncs; turns the device flag into GP_ERR_NOT_PD
// chol_cluster.hip: one matrix by a cluster of workgroups (factor in place, W = L^-1 when W is given), one launch; false =
// not a launch the cluster takes (shape, switch, exchange area not allocatable inside a stream capture)
bool cholesky_cluster_takes(int M, int count);
bool launch_cholesky_cluster_single(gp_handle h, double* A, double* W, int M, int64_t ld, int pivot_base, gp_status* st);
bool launch_cholesky_cluster_batched(gp_handle h, double* const* d_mats, double* const* d_W, const int* d_M, const int* d_ld,
                                     int count, int minM, int maxM, gp_status* st);
void cholesky_cluster_release(gp_handle h);

// gemm.hip
struct GemmProblem {
  const double* A; const double* B; double* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  // epilogue / prologue extras (meaning depends on the launch variant)
  const double* v0; const double* v1; const double* v2;
  double* o0; double* o1; double* o2;
  DevKern kern;           // for epilogues that re-evaluate the covariance (hyper-gradient)
  const double* xa; const double* xb;  // kernel inputs (z, x) for those epilogues
  int a_f32;              // small vector kernels (rowdot): A is a float32 strip (lda in floats)
  int pad_;
};
enum GemmTri { TRI_NONE = 0, TRI_LOWER = 1, TRI_UPPER = 2 };
struct GemmFlags {
  int transA = 0, transB = 0;
  int triA = TRI_NONE;   // structure of op(A) (M x K)
  int triB = TRI_NONE;   // structure of op(B) (K x N)
  int triC = TRI_NONE;   // TRI_LOWER: tiles strictly above the diagonal are skipped, upper part zeroed
  double alpha = 1.0, beta = 0.0;
  int big_tiles = 0;     // 128x128 tiles (strip GEMMs) instead of 64x64
  int epilogue = 1;      // bitmask of GemmEpi
  int scale_mode = 0;    // 0 none; 1: opB(B)(k,n) *= v1[n]; 2: opB(B)(k,n) *= v1[k]
  int timer = GP_TIMER_SMALL_GEMM;
  int role = 0;          // 1 cond_A, 2 cond_LTA (needs transA), 3 kuf_bar: dedicated 128x128 instantiations
  int tile_m0 = 0, tile_mcount = 0;   // strip products (big tiles, no split-K): only row-blocks [m0, m0 + mcount) (0 = all)
  const double* aux_x = nullptr;      // role 5 (gemm_strip.hip): the frames x of the batch; aux_ktype: the stationary kernel type
  int aux_ktype = -1;
  int uniform_aligned = 0;            // the caller vouches: every problem has M = maxM (= K structure), N = maxN, 16-byte
                                      // aligned operands with even leading dimensions (gemm_strip.hip's lean form)
  int a32_ok = 0;                     // float32 strips (gemm_wave_f32.hip): every problem's xb = M * M floats of scratch for the M x M operand
  int rows64_ok = 0;                  // roles 1, 2: the caller has sized o0 / o1 for one partial row per 64-ROW tile and asked
                                      // gemm_wave_takes() how many rows the launch will write (gemm_wave.hip)
};
// gemm_wave.hip: the strip products without LDS or barriers, a 64 x 64 tile per wavefront (true = taken, status in *st)
bool launch_gemm_wave(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f, gp_status* st);
bool gemm_wave_takes(int role, int maxM, int maxN, int uniform_aligned);
int gemm_fused_contraction_records(int maxM, int maxN, int ktype);
// gemm_wave_f32.hip: the same for float32 strips
bool launch_gemm_wave_f32(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f, gp_status* st);
bool gemm_wave_f32_takes(int role, int maxM, int maxN, int uniform_aligned);
bool launch_gemm_strip_lean(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f,
                            gp_status* st);
bool gemm_strip_fused_contraction_ok(int maxM, int maxN, int ktype);
bool gemm_f32_fused_contraction_ok(int maxM, int maxN, int ktype);     // gemm_f32.hip: the same for float32 strips
enum GemmEpi {
  EPI_STORE = 1,     // C = alpha*acc + beta*C
  EPI_COLSUMSQ = 2,  // o0[rowblk*N + n] = sum over the tile's rows of (alpha*acc)^2
  EPI_COLDOT = 4     // o1[rowblk*N + n] = sum over the tile's rows of alpha*acc * v0[row]
};
// Batched: d_probs is a device array of `batch` problems; maxM/maxN bound the grid.
gp_status launch_gemm_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN,
                              const GemmFlags& f);
// H = (X * diag(d)) * Y^T over the long dimension with split-K partial slabs + reduction (deterministic).
//   prob.A = X (M x Nlong, lda), prob.B = Y (M x Nlong, ldb), prob.v1 = d (Nlong, when scale_by_k), prob.C = H (M x M, ldc)
//   prob.M = prob.N = M, prob.K = Nlong; prob.o2 = slab workspace (nsplit * M * M doubles).  Only the lower triangle is computed when sym != 0
//   (upper mirrored).
// uniform_aligned != 0: every problem has M = maxM, K = maxNlong, 16-byte aligned rows (gemm_strip.hip's lean form)
gp_status launch_gemm_nt_reduce_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxNlong,
                                        int nsplit, int sym, int scale_by_k, double alpha, int uniform_aligned = 0);
bool launch_gemm_strip_f32_lean(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f,
                                gp_status* st);
bool launch_gemm_strip_nt_lean(gp_handle h, const GemmProblem* d_probs, int batch, int M, int Nlong, int nsplit, int sym,
                               int scale_by_k, gp_status* st);
gp_status launch_tri_inverse_batched(gp_handle h, const double* const* d_L, double* const* d_W, const int* d_M,
                                     const int* d_ld, int batch);
int gemm_nt_nsplit(int M, int Nlong, int batch);
gp_status launch_slab_reduce(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int nsplit, int sym, double alpha);
// gemm_f32.hip: the strip products on float32 strips (roles 1-3: A float64 M x M, B / C float32 strips; nt: float32
// strips, float64 slabs).  Same descriptor struct; ldb / ldc (and lda for nt) count floats.
gp_status launch_gemm_f32_role(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f);
gp_status launch_gemm_f32_nt_reduce_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxNlong,
                                            int nsplit, int sym, int scale_by_k, double alpha, int uniform_aligned = 0);
// leading dimension (in elements) of an M x N strip: even for float64, a multiple of 4 for float32 (16-byte rows)
static inline int64_t gp_strip_ld(int N, bool f32) { return f32 ? (((int64_t)N + 3) & ~(int64_t)3) : (((int64_t)N + 1) & ~(int64_t)1); }
// doubles that hold an M x ld strip of either type
static inline size_t gp_strip_doubles(size_t M, int N, bool f32) {
  const size_t e = M * (size_t)gp_strip_ld(N, f32);
  return f32 ? (e + 1) / 2 : e;
}
int gemm_rowblocks(int M, int big_tiles);

// grouped covariance builds (cov.hip): one launch for all matrices of a kernel family
struct CovItem {
  DevKern k; const double* x1; const double* x2; double* out; const double* f1; const double* f2;
  int64_t ld; double diag_add; int n1, n2, accumulate, vec_ok;
  int f32out, pad_;     // out is float32 (ld in floats)
};
struct FeatItem { DevKern k; const double* x; double* f; int n; int pad; };
// one pass for K = sum of P MercerMatern12sm kernels (cov.hip; true = taken, status in *st)
bool launch_kernel_build_sum(gp_handle h, const DevKern* kernels, double* const* feats, int P, const double* x1, int n1,
                             const double* x2, int n2, double* out, int64_t ld, double diag_add, int f32out, gp_status* st);
void cov_item_fill(CovItem* it, DevKern k, const double* x1, int n1, const double* x2, int n2, double* out, int64_t ld,
                   int accumulate, double diag_add, double* feat_ws, int f32out = 0);
gp_status launch_sm_features_items(gp_handle h, const FeatItem* d_items, int count, int max_n, int mpad,
                                   const double* x_shared, int n_shared);
// engine_strips != 0: the caller vouches that every item writes an engine strip of the shared frames (item.n2 < 0, 256-byte
// aligned output, leading dimension gp_strip_ld) — what the lean Mercer form needs to know on the host
gp_status launch_kernel_build_items(gp_handle h, int type, int m, const CovItem* d_items, int count, int max_n1,
                                    int max_n2, const double* x2_shared, int n2_shared, int engine_strips = 0);
gp_status launch_overlap_merge(gp_handle h, const double* y, int nw, int ws, int64_t ldy, int n, int square, double* out);
// lik.hip
// whitened KL: each item writes GP_KL_BLOCKS partial sums to out[0..GP_KL_BLOCKS)
#define GP_KL_BLOCKS 16
int mpd_lik_blocks(int N);   // block partials (2 doubles each) one launch over N frames leaves
gp_status launch_mpd_lik(gp_handle h, const double* Fmu, const double* Fvar, int64_t f_rs, int64_t f_cs,
                         const double* y, int N, int P, int nlin, const double* noise_var, double scale,
                         double* per_frame, double* partial_sums, int* num_partials_out,
                         double* gFmu, double* gFvar, double* psum = nullptr, const double* gsum = nullptr);
gp_status launch_finish_sum(gp_handle h, const double* partials, int count, int stride, int nsums, double* out,
                            double mul, int accumulate);

// opt.hip
gp_status launch_transform_forward(gp_handle h, const double* free_state, const uint8_t* tcode, int64_t n,
                                   double* params);
gp_status launch_transform_backward(gp_handle h, const double* params, const uint8_t* tcode, int64_t n,
                                    double* free_state);
gp_status launch_adam(gp_handle h, double* free_state, double* params, const double* grad, const uint8_t* tcode,
                      double* m, double* v, int64_t n, int64_t t, double lr, double b1, double b2, double eps);
