// abi.hip — extern "C" entry points of libgpitch_hip.so (see include/gpitch_abi.h).
#include "engine.h"
#include "switches.h"
#include <string.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

// ------------------------------------------------------------------------------------------------
// timers
GpTimerScope::GpTimerScope(gp_handle h_, int which_) : h(h_), which(which_) {
  if (!h || !h->timers_on) return;
  auto get = [&]() {
    hipEvent_t e;
    if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
  };
  e0 = get(); e1 = get();
  if (e0) (void)hipEventRecord(e0, h->stream);
}
GpTimerScope::~GpTimerScope() {
  if (!h || !h->timers_on || !e0 || !e1) return;
  (void)hipEventRecord(e1, h->stream);
  h->pending.push_back({e0, e1, which});
}

bool gp_aux_fork(gp_handle h) {
  if (h->aux_active) return false;
  if (!h->aux_stream) {
    // (aux_priority: the helper stream carries what the step's dependent chains wait for — the Kuu factorisation, the split-K
    // product in front of the M x M chain — beside device-filling strip kernels of the main stream: its workgroups go first)
    hipError_t ce;
    if (gp_switches().aux_priority) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      ce = hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, hi);
    } else {
      ce = hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking);
    }
    if (ce != hipSuccess) { (void)hipGetLastError(); h->aux_stream = nullptr; return false; }
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(h->aux_stream); h->aux_stream = nullptr;
      return false;
    }
  }
  if (hipEventRecord(h->ev_fork, h->stream) != hipSuccess) return false;
  if (hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0) != hipSuccess) return false;
  h->main_stream_saved = h->stream;
  h->stream = h->aux_stream;
  h->aux_active = true;
  return true;
}

bool gp_aux_resume(gp_handle h) {
  if (!h->aux_pending || h->aux_active || !h->aux_stream) return false;
  if (!h->ev_mid && hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming) != hipSuccess) { h->ev_mid = nullptr; return false; }
  if (hipEventRecord(h->ev_mid, h->stream) != hipSuccess) return false;
  if (hipStreamWaitEvent(h->aux_stream, h->ev_mid, 0) != hipSuccess) return false;
  h->main_stream_saved = h->stream;
  h->stream = h->aux_stream;
  h->aux_active = true;
  return true;
}

gp_status gp_aux_end(gp_handle h) {
  if (!h->aux_active) return GP_OK;
  hipError_t e = hipEventRecord(h->ev_join, h->aux_stream);
  h->stream = h->main_stream_saved;
  h->aux_active = false;
  h->aux_pending = true;
  if (e != hipSuccess) return gp_fail(h, GP_ERR_HIP, "hipEventRecord on the helper stream failed");
  return GP_OK;
}

gp_status gp_aux_join(gp_handle h) {
  if (!h->aux_pending) return GP_OK;
  h->aux_pending = false;
  GP_HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  return GP_OK;
}

bool gp_side_begin(gp_handle h) {
  if (h->side_active || h->side_pending) return false;
  if (!h->side_stream) {
    if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess) { h->side_stream = nullptr; return false; }
    if (hipEventCreateWithFlags(&h->ev_side_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_side_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(h->side_stream); h->side_stream = nullptr;
      return false;
    }
  }
  if (hipEventRecord(h->ev_side_fork, h->stream) != hipSuccess) return false;
  if (hipStreamWaitEvent(h->side_stream, h->ev_side_fork, 0) != hipSuccess) return false;
  h->side_saved = h->stream;
  h->stream = h->side_stream;
  h->side_active = true;
  return true;
}

gp_status gp_side_end(gp_handle h) {
  if (!h->side_active) return GP_OK;
  hipError_t e = hipEventRecord(h->ev_side_join, h->side_stream);
  h->stream = h->side_saved;
  h->side_active = false;
  h->side_pending = true;
  if (e != hipSuccess) return gp_fail(h, GP_ERR_HIP, "hipEventRecord on the side stream failed");
  return GP_OK;
}

gp_status gp_side_join(gp_handle h) {
  if (!h->side_pending) return GP_OK;
  h->side_pending = false;
  GP_HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_side_join, 0));
  return GP_OK;
}

static gp_status drain_timers(gp_handle h) {
  for (auto& r : h->pending) {
    GP_HIP_CHECK(h, hipEventSynchronize(r.e1));
    float ms = 0.f;
    GP_HIP_CHECK(h, hipEventElapsedTime(&ms, r.e0, r.e1));
    h->timer_ms[r.which] += ms;
    h->timer_n[r.which] += 1;
    h->event_pool.push_back(r.e0);
    h->event_pool.push_back(r.e1);
  }
  h->pending.clear();
  return GP_OK;
}

extern "C" {

int32_t gp_abi_version(void) { return GPITCH_ABI_VERSION; }

gp_status gp_create(int32_t device_id, void* stream, gp_handle* out) {
  if (!out) return GP_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GP_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= count) return GP_ERR_BAD_ARG;
  gp_handle h = new gp_handle_s();
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess) { delete h; return GP_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete h; return GP_ERR_HIP; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete h; return GP_ERR_NO_DEVICE; }
  h->num_cus = prop.multiProcessorCount;
  // NULL selects HIP's default (null) stream — the one torch and plain HIP code use unless told otherwise,
  // so work enqueued here is ordered with the caller's other device work without extra events.
  h->stream = (hipStream_t)stream;
  h->own_stream = false;
  if (hipMalloc((void**)&h->d_status, 4 * sizeof(int32_t)) != hipSuccess) { delete h; return GP_ERR_HIP; }
  (void)hipMemsetAsync(h->d_status, 0, 4 * sizeof(int32_t), h->stream);
  *out = h;
  return GP_OK;
}

gp_status gp_destroy(gp_handle h) {
  if (!h) return GP_OK;
  (void)hipStreamSynchronize(h->stream);
  for (auto& r : h->pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : h->event_pool) (void)hipEventDestroy(e);
  if (h->poll_host) { (void)hipStreamSynchronize(h->stream); (void)hipHostFree(h->poll_host); (void)hipEventDestroy(h->ev_poll); }
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  if (h->side_stream) { (void)hipStreamSynchronize(h->side_stream); (void)hipStreamDestroy(h->side_stream); }
  if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
  if (h->ev_era) (void)hipEventDestroy(h->ev_era);
  if (h->ev_kuu) (void)hipEventDestroy(h->ev_kuu);
  if (h->ev_diag) (void)hipEventDestroy(h->ev_diag);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  cholesky_cluster_release(h);
  if (h->d_status) (void)hipFree(h->d_status);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return GP_OK;
}

gp_status gp_sync(gp_handle h) {
  if (!h) return GP_ERR_BAD_ARG;
  // the helper stream is normally joined into the handle's stream before an entry point returns; the one exception is
  // work prefetched for a backward pass that was never asked for (gp_pdgp_elbo_begin without _end)
  if (h->aux_stream) GP_HIP_CHECK(h, hipStreamSynchronize(h->aux_stream));
  if (h->side_stream) GP_HIP_CHECK(h, hipStreamSynchronize(h->side_stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return GP_OK;
}

const char* gp_last_error(gp_handle h) { return h ? h->last_error.c_str() : "null handle"; }
int32_t gp_last_not_pd_index(gp_handle h) { return h ? h->not_pd_index : -1; }

gp_status gp_timers_enable(gp_handle h, int32_t on) { if (!h) return GP_ERR_BAD_ARG; h->timers_on = (on != 0); return GP_OK; }
gp_status gp_timers_reset(gp_handle h) {
  if (!h) return GP_ERR_BAD_ARG;
  GP_CHECK(drain_timers(h));
  for (int i = 0; i < GP_TIMER_COUNT; i++) { h->timer_ms[i] = 0; h->timer_n[i] = 0; }
  return GP_OK;
}
gp_status gp_timers_read(gp_handle h, int32_t which, double* total_ms, int64_t* launches) {
  if (!h || which < 0 || which >= GP_TIMER_COUNT) return GP_ERR_BAD_ARG;
  GP_CHECK(drain_timers(h));
  if (total_ms) *total_ms = h->timer_ms[which];
  if (launches) *launches = h->timer_n[which];
  return GP_OK;
}

// ------------------------------------------------------------------------------------------------
// L2 operators
static bool kern_ok(const gp_kernel_desc* k) {
  if (!k || !k->theta) return false;
  if (k->type < GP_KERN_MATERN12 || k->type > GP_KERN_LAST) return false;
  if (gp_kern_has_partials(k->type) && (k->num_partials < 1 || k->num_partials > 32)) return false;
  return true;
}

gp_status gp_kernel_build(gp_handle h, const gp_kernel_desc* kern, const double* x1, int32_t n1, const double* x2,
                          int32_t n2, double* out, int64_t ld, int32_t accumulate) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!kern_ok(kern) || !x1 || !out || n1 < 0) return gp_fail(h, GP_ERR_BAD_ARG, "gp_kernel_build: bad argument");
  if (!x2) n2 = n1;
  if (n2 < 0 || ld < n2) return gp_fail(h, GP_ERR_BAD_ARG, "gp_kernel_build: bad n2/ld");
  if (n1 == 0 || n2 == 0) return GP_OK;
  DevKern k = dev_kern(kern);
  double* feat = nullptr;
  if (gp_kern_is_mercer(k.type)) {
    // one-shot operator: the feature scratch is a transient allocation (the plans carry their own)
    size_t nd = kernel_build_feat_ws_doubles(k.m, n1, x2 ? n2 : n1);
    GP_HIP_CHECK(h, hipMallocAsync((void**)&feat, nd * sizeof(double), h->stream));
  }
  gp_status s = launch_kernel_build(h, k, x1, n1, x2, n2, out, ld, accumulate, 0.0, feat);
  if (feat) GP_HIP_CHECK(h, hipFreeAsync(feat, h->stream));
  return s;
}

// float32 output of the same build (inputs and arithmetic float64, one rounding at the store): the Kuf strips of the
// float32 configurations
gp_status gp_kernel_build_f32(gp_handle h, const gp_kernel_desc* kern, const double* x1, int32_t n1, const double* x2,
                              int32_t n2, float* out, int64_t ld, int32_t accumulate) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!kern_ok(kern) || !x1 || !out || n1 < 0) return gp_fail(h, GP_ERR_BAD_ARG, "gp_kernel_build_f32: bad argument");
  if (!x2) n2 = n1;
  if (n2 < 0 || ld < n2) return gp_fail(h, GP_ERR_BAD_ARG, "gp_kernel_build_f32: bad n2/ld");
  if (n1 == 0 || n2 == 0) return GP_OK;
  DevKern k = dev_kern(kern);
  double* feat = nullptr;
  if (gp_kern_is_mercer(k.type)) {
    size_t nd = kernel_build_feat_ws_doubles(k.m, n1, x2 ? n2 : n1);
    GP_HIP_CHECK(h, hipMallocAsync((void**)&feat, nd * sizeof(double), h->stream));
  }
  gp_status s = launch_kernel_build(h, k, x1, n1, x2, n2, reinterpret_cast<double*>(out), ld, accumulate, 0.0, feat, 0, 1);
  if (feat) GP_HIP_CHECK(h, hipFreeAsync(feat, h->stream));
  return s;
}

gp_status gp_kernel_diag(gp_handle h, const gp_kernel_desc* kern, int32_t n, double* out, int32_t accumulate) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!kern_ok(kern) || !out || n < 0) return gp_fail(h, GP_ERR_BAD_ARG, "gp_kernel_diag: bad argument");
  return launch_kernel_diag(h, dev_kern(kern), n, out, accumulate);
}

size_t gp_chol_workspace_bytes(int32_t M) {
  if (M <= 0) return 256;
  return gp_align_up((size_t)M * M * sizeof(double), 256) + gp_align_up(kernel_build_feat_ws_doubles(32, M, M) * sizeof(double), 256);
}

gp_status gp_kuu_cholesky(gp_handle h, const gp_kernel_desc* kern, const double* z, int32_t M, double jitter, double* L,
                          double* Linv, void* workspace, size_t workspace_bytes) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!kern_ok(kern) || !z || M <= 0) return gp_fail(h, GP_ERR_BAD_ARG, "gp_kuu_cholesky: bad argument");
  GpArena ar(workspace, workspace_bytes);
  double* Lbuf = L ? L : ar.take<double>((size_t)M * M);
  DevKern k = dev_kern(kern);
  double* feat = gp_kern_is_mercer(k.type) ? ar.take<double>(kernel_build_feat_ws_doubles(k.m, M, M)) : nullptr;
  if (!ar.ok || !Lbuf) return gp_fail(h, GP_ERR_WORKSPACE, "gp_kuu_cholesky: workspace too small");
  GP_CHECK(launch_kernel_build(h, k, z, M, nullptr, M, Lbuf, M, 0, jitter, feat));
  if (Linv) GP_CHECK(launch_cholesky_inverse_single(h, Lbuf, Linv, M, M));   // one launch (on an LDS copy for M <= 64)
  else GP_CHECK(launch_cholesky_single(h, Lbuf, M, M));
  return check_not_pd(h);
}

gp_status gp_cholesky_inplace(gp_handle h, double* A, int32_t M, int64_t ld) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!A || M <= 0 || ld < M) return gp_fail(h, GP_ERR_BAD_ARG, "gp_cholesky_inplace: bad argument");
  GP_CHECK(launch_cholesky_single(h, A, M, ld));
  return check_not_pd(h);
}

size_t gp_conditional_workspace_bytes(int32_t N, int32_t M) {
  if (N <= 0 || M <= 0) return 256;
  return cond_task_workspace_doubles(M, N, 32, false) * sizeof(double) + cond_batch_desc_bytes(1) + 4096;
}

static gp_status conditional_diag_impl(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N,
                                       const double* z, int32_t M, const double* q_mu, const double* q_sqrt,
                                       int32_t whiten, double jitter, double* fmean, double* fvar, void* workspace,
                                       size_t workspace_bytes, bool f32, const char* who) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!kern_ok(kern) || !xnew || !z || !q_mu || !fmean || !fvar || N < 0 || M <= 0)
    return gp_fail(h, GP_ERR_BAD_ARG, who);
  if (N == 0) return GP_OK;
  GpArena ar(workspace, workspace_bytes);
  CondBatch cb;
  cb.f32 = f32;
  cb.tasks.resize(1);
  CondTask& t = cb.tasks[0];
  t.kern = dev_kern(kern); t.z = z; t.M = M; t.q_mu = q_mu; t.q_sqrt = q_sqrt; t.fmean = fmean; t.fvar = fvar;
  cb.desc_bytes = cond_batch_desc_bytes(1);
  cb.d_desc = ar.take<char>(cb.desc_bytes);
  if (!cond_task_carve(ar, t, N, whiten != 0, f32) || !cb.d_desc)
    return gp_fail(h, GP_ERR_WORKSPACE, "gp_conditional_diag: workspace too small");
  cb.N = N;
  gp_status st = cond_batch_upload(h, cb, whiten != 0, jitter);
  if (st == GP_OK) st = cond_batch_run(h, cb, xnew, N, whiten != 0, jitter);
  // cb.h_desc (a member of this stack object) is the source of an asynchronous descriptor copy: the stream is drained
  // on EVERY path before it goes out of scope (check_not_pd synchronises)
  const gp_status pd = check_not_pd(h);
  return st != GP_OK ? st : pd;
}

gp_status gp_conditional_diag(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N, const double* z,
                              int32_t M, const double* q_mu, const double* q_sqrt, int32_t whiten, double jitter,
                              double* fmean, double* fvar, void* workspace, size_t workspace_bytes) {
  return conditional_diag_impl(h, kern, xnew, N, z, M, q_mu, q_sqrt, whiten, jitter, fmean, fvar, workspace,
                               workspace_bytes, false, "gp_conditional_diag: bad argument");
}

// the same conditional with the M x N strips (Kuf, A) in float32 and the two strip products on the float32 matrix
// path; Kuu, its factor, W, and every reduction stay float64 (whitened form only)
gp_status gp_conditional_diag_f32(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N, const double* z,
                                  int32_t M, const double* q_mu, const double* q_sqrt, double jitter, double* fmean,
                                  double* fvar, void* workspace, size_t workspace_bytes) {
  return conditional_diag_impl(h, kern, xnew, N, z, M, q_mu, q_sqrt, 1, jitter, fmean, fvar, workspace, workspace_bytes,
                               true, "gp_conditional_diag_f32: bad argument");
}

gp_status gp_conditional_diag_f32w(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N, const double* z,
                                   int32_t M, const double* q_mu, const double* q_sqrt, int32_t whiten, double jitter,
                                   double* fmean, double* fvar, void* workspace, size_t workspace_bytes) {
  return conditional_diag_impl(h, kern, xnew, N, z, M, q_mu, q_sqrt, whiten, jitter, fmean, fvar, workspace, workspace_bytes,
                               true, "gp_conditional_diag_f32w: bad argument");
}

// full_cov = True of the same operator (GPflow 0.5 conditionals.conditional): the N x N posterior covariance
//   K(xnew, xnew) - A^T A + (Lq^T A')^T (Lq^T A'),   A = Lm^-1 Kuf,  A' = A (whitened) or Lm^-T A (unwhitened).
// Never used by the reference's own callers (its N is a window of frames: N^2 values); here for API parity.
size_t gp_conditional_full_workspace_bytes(int32_t N, int32_t M) {
  if (N <= 0 || M <= 0) return 256;
  return gp_conditional_workspace_bytes(N, M) + gp_align_up((size_t)N * sizeof(double), 256) +
         gp_align_up(3 * sizeof(GemmProblem), 256) + 512;
}

gp_status gp_conditional_full(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N, const double* z,
                              int32_t M, const double* q_mu, const double* q_sqrt, int32_t whiten, double jitter,
                              double* fmean, double* fcov, void* workspace, size_t workspace_bytes) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!kern_ok(kern) || !xnew || !z || !q_mu || !fmean || !fcov || N < 0 || M <= 0)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_conditional_full: bad argument");
  if (N == 0) return GP_OK;
  if (!workspace || workspace_bytes < gp_conditional_full_workspace_bytes(N, M) || (((uintptr_t)workspace) & 255))
    return gp_fail(h, GP_ERR_WORKSPACE, "gp_conditional_full: workspace too small or not 256-byte aligned");
  GpArena ar(workspace, workspace_bytes);
  double* fvar_diag = ar.take<double>(N);
  GemmProblem* d_prob = ar.take<GemmProblem>(3);
  CondBatch cb;
  cb.tasks.resize(1);
  CondTask& t = cb.tasks[0];
  t.kern = dev_kern(kern); t.z = z; t.M = M; t.q_mu = q_mu; t.q_sqrt = q_sqrt; t.fmean = fmean; t.fvar = fvar_diag;
  cb.desc_bytes = cond_batch_desc_bytes(1);
  cb.d_desc = ar.take<char>(cb.desc_bytes);
  if (!cond_task_carve(ar, t, N, whiten != 0, false) || !cb.d_desc || !ar.ok)
    return gp_fail(h, GP_ERR_WORKSPACE, "gp_conditional_full: workspace too small");
  cb.N = N;
  gp_status st = cond_batch_upload(h, cb, whiten != 0, jitter);
  if (st == GP_OK) st = cond_batch_run(h, cb, xnew, N, whiten != 0, jitter);     // mean, and A (A') left in the workspace
  const int64_t ld = gp_strip_ld(N, false);
  GemmProblem hp[3];
  memset(hp, 0, sizeof(hp));
  const double* Aeff = whiten ? t.A : t.A2;
  { GemmProblem& r = hp[0]; r.A = t.A; r.lda = ld; r.B = t.A; r.ldb = ld; r.C = fcov; r.ldc = N; r.M = N; r.N = N; r.K = M; }
  { GemmProblem& r = hp[1]; r.A = q_sqrt; r.lda = M; r.B = Aeff; r.ldb = ld; r.C = t.Kuf; r.ldc = ld; r.M = M; r.N = N; r.K = M; }
  { GemmProblem& r = hp[2]; r.A = t.Kuf; r.lda = ld; r.B = t.Kuf; r.ldb = ld; r.C = fcov; r.ldc = N; r.M = N; r.N = N; r.K = M; }
  if (st == GP_OK) {
    hipError_t e = hipMemcpyAsync(d_prob, hp, sizeof(hp), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) st = gp_fail(h, GP_ERR_HIP, hipGetErrorString(e));
  }
  if (st == GP_OK) st = launch_kernel_build(h, t.kern, xnew, N, nullptr, N, fcov, N, 0, 0.0, t.feat, 0, 0);
  if (st == GP_OK) { GemmFlags f; f.transA = 1; f.alpha = -1.0; f.beta = 1.0; st = launch_gemm_batched(h, d_prob + 0, 1, N, N, f); }
  if (st == GP_OK && q_sqrt) {
    GemmFlags f; f.transA = 1; f.triA = TRI_UPPER;          // tril(q_sqrt)^T, as matrix_band_part has it
    st = launch_gemm_batched(h, d_prob + 1, 1, M, N, f);
    if (st == GP_OK) { f = GemmFlags(); f.transA = 1; f.beta = 1.0; st = launch_gemm_batched(h, d_prob + 2, 1, N, N, f); }
  }
  // cb.h_desc and hp are sources of asynchronous copies: the stream is drained on every path (check_not_pd synchronises)
  const gp_status pd = check_not_pd(h);
  return st != GP_OK ? st : pd;
}

gp_status gp_mpd_varexp(gp_handle h, const double* Fmu, const double* Fvar, const double* y, int32_t N, int32_t P,
                        int32_t nlin, const double* noise_var, double* per_frame, double* sum_host) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!Fmu || !Fvar || !y || !noise_var || N < 0 || P < 1 || nlin < 0 || nlin > 2)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_mpd_varexp: bad argument");
  if (N == 0) { if (sum_host) *sum_host = 0.0; return GP_OK; }
  int blocks = mpd_lik_blocks(N);
  double* partial = nullptr;
  if (sum_host) GP_HIP_CHECK(h, hipMallocAsync((void**)&partial, (2 * (size_t)blocks + 2) * sizeof(double), h->stream));
  int nb = 0;
  gp_status s = launch_mpd_lik(h, Fmu, Fvar, (int64_t)2 * P, 1, y, N, P, nlin, noise_var, 1.0, per_frame, partial, &nb,
                               nullptr, nullptr);
  if (s == GP_OK && sum_host) {
    double* res = partial + 2 * (size_t)blocks;
    s = launch_finish_sum(h, partial, nb, 2, 1, res, 1.0, 0);
    if (s == GP_OK) {
      GP_HIP_CHECK(h, hipMemcpyAsync(sum_host, res, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
  }
  if (partial) GP_HIP_CHECK(h, hipFreeAsync(partial, h->stream));
  return s;
}

size_t gp_gauss_kl_workspace_bytes(int32_t M, int32_t with_kernel) {
  if (M <= 0 || !with_kernel) return 4096;
  const size_t mm = gp_align_up((size_t)M * M * sizeof(double), 256);
  return 4096 + 2 * mm + gp_align_up((size_t)gemm_rowblocks(M, 0) * M * sizeof(double), 256) + gp_chol_workspace_bytes(M);
}

// shared body: kern != NULL builds K = kern.K(z) + jitter I; Kmat != NULL takes the caller's K (M x M, ld = M)
static gp_status gauss_kl_impl(gp_handle h, const double* q_mu, const double* q_sqrt, int32_t M,
                               const gp_kernel_desc* kern, const double* z, double jitter, const double* Kmat,
                               double* out_host, void* workspace, size_t workspace_bytes, const char* who) {
  GpArena ar(workspace, workspace_bytes);
  const size_t item_bytes = kl_item_bytes() > klu_item_bytes() ? kl_item_bytes() : klu_item_bytes();
  char* d_item = ar.take<char>(item_bytes);
  GemmProblem* d_prob = ar.take<GemmProblem>(1);
  double* d_out = ar.take<double>(GP_KL_BLOCKS + 4);
  double* d_res = d_out;
  if (!ar.ok) return gp_fail(h, GP_ERR_WORKSPACE, who);
  std::vector<char> item(item_bytes);
  GemmProblem r;
  if (kern || Kmat) {
    // L = chol(K), W = L^-1, trace term from the column sums of (W Lq)^2
    double* L = ar.take<double>((size_t)M * M);
    double* W = ar.take<double>((size_t)M * M);
    const int nrb = gemm_rowblocks(M, 0);
    double* tr = ar.take<double>((size_t)nrb * M);
    void* cw = ar.take<char>(gp_chol_workspace_bytes(M));
    if (!ar.ok) return gp_fail(h, GP_ERR_WORKSPACE, who);
    if (kern) {
      GP_CHECK(gp_kuu_cholesky(h, kern, z, M, jitter, L, W, cw, gp_chol_workspace_bytes(M)));
    } else {
      GP_HIP_CHECK(h, hipMemcpyAsync(L, Kmat, (size_t)M * M * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      GP_CHECK(launch_cholesky_single(h, L, M, M));
      GP_CHECK(launch_tri_inverse_single(h, L, W, M, M));
    }
    memset(&r, 0, sizeof(r));
    r.A = W; r.lda = M; r.B = q_sqrt; r.ldb = M; r.M = M; r.N = M; r.K = M; r.ldc = M; r.o0 = tr;
    GP_HIP_CHECK(h, hipMemcpyAsync(d_prob, &r, sizeof(r), hipMemcpyHostToDevice, h->stream));
    GemmFlags f;
    f.triA = TRI_LOWER; f.triB = TRI_LOWER; f.epilogue = EPI_COLSUMSQ;
    GP_CHECK(launch_gemm_batched(h, d_prob, 1, M, M, f));
    klu_item_fill(item.data(), q_mu, q_sqrt, L, W, tr, nrb, M, d_out);
    GP_HIP_CHECK(h, hipMemcpyAsync(d_item, item.data(), item.size(), hipMemcpyHostToDevice, h->stream));
    GP_CHECK(launch_kl_unwhite(h, d_item, 1));
  } else {
    kl_item_fill(item.data(), q_mu, q_sqrt, M, d_out, nullptr, nullptr);
    GP_HIP_CHECK(h, hipMemcpyAsync(d_item, item.data(), item.size(), hipMemcpyHostToDevice, h->stream));
    GP_CHECK(launch_kl_white(h, d_item, 1));
    d_res = d_out + GP_KL_BLOCKS;
    GP_CHECK(launch_finish_sum(h, d_out, GP_KL_BLOCKS, 1, 1, d_res, 1.0, 0));
  }
  GP_HIP_CHECK(h, hipMemcpyAsync(out_host, d_res, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // also keeps the host-side `r` / `item` alive long enough
  return check_not_pd(h);
}

gp_status gp_gauss_kl(gp_handle h, const double* q_mu, const double* q_sqrt, int32_t M, const gp_kernel_desc* kern,
                      const double* z, double jitter, double* out_host, void* workspace, size_t workspace_bytes) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!q_mu || !q_sqrt || M < 1 || !out_host || (kern && (!kern_ok(kern) || !z)))
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_gauss_kl: bad argument");
  return gauss_kl_impl(h, q_mu, q_sqrt, M, kern, z, jitter, nullptr, out_host, workspace, workspace_bytes,
                       "gp_gauss_kl: workspace too small (gp_gauss_kl_workspace_bytes)");
}

gp_status gp_gauss_kl_matrix(gp_handle h, const double* q_mu, const double* q_sqrt, int32_t M, const double* K,
                             double* out_host, void* workspace, size_t workspace_bytes) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!q_mu || !q_sqrt || M < 1 || !out_host || !K) return gp_fail(h, GP_ERR_BAD_ARG, "gp_gauss_kl_matrix: bad argument");
  return gauss_kl_impl(h, q_mu, q_sqrt, M, nullptr, nullptr, 0.0, K, out_host, workspace, workspace_bytes,
                       "gp_gauss_kl_matrix: workspace too small (gp_gauss_kl_workspace_bytes)");
}

gp_status gp_overlap_merge(gp_handle h, const double* windows, int32_t num_windows, int32_t ws, int64_t ld, int32_t n,
                           int32_t square, double* out) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!windows || !out || num_windows < 2 || ws < 3 || (ws & 1) == 0 || ld < ws)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_overlap_merge: need >= 2 windows of odd length ws >= 3, ld >= ws");
  const int ll = (ws - 1) / 2;
  if (n != ll * (num_windows + 1) + 1) return gp_fail(h, GP_ERR_BAD_ARG, "gp_overlap_merge: n must be (ws-1)/2 * (num_windows+1) + 1");
  return launch_overlap_merge(h, windows, num_windows, ws, ld, n, square, out);
}

gp_status gp_transform_register_logistic(gp_handle h, double a, double b, uint8_t* code_out) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!code_out || !(b > a)) return gp_fail(h, GP_ERR_BAD_ARG, "gp_transform_register_logistic: need b > a");
  for (int i = 0; i < h->num_logistic; i++)
    if (h->logistic.a[i] == a && h->logistic.b[i] == b) { *code_out = (uint8_t)(3 + i); return GP_OK; }
  if (h->num_logistic >= GP_MAX_LOGISTIC)
    return gp_fail(h, GP_ERR_UNSUPPORTED, "gp_transform_register_logistic: table full (8 distinct (a, b) pairs per handle)");
  h->logistic.a[h->num_logistic] = a; h->logistic.b[h->num_logistic] = b;
  *code_out = (uint8_t)(3 + h->num_logistic++);
  return GP_OK;
}

gp_status gp_transform_forward(gp_handle h, const double* fs, const uint8_t* tcode, int64_t n, double* params) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!fs || !tcode || !params || n < 0) return gp_fail(h, GP_ERR_BAD_ARG, "gp_transform_forward: bad argument");
  return launch_transform_forward(h, fs, tcode, n, params);
}
gp_status gp_transform_backward(gp_handle h, const double* params, const uint8_t* tcode, int64_t n, double* fs) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!fs || !tcode || !params || n < 0) return gp_fail(h, GP_ERR_BAD_ARG, "gp_transform_backward: bad argument");
  return launch_transform_backward(h, params, tcode, n, fs);
}
// Blocking form: drains the handle's streams, then GP_ERR_NOT_PD (pivot index via gp_last_not_pd_index) if a Cholesky has
// failed since the flag was last cleared; clears the flag.
gp_status gp_check_not_pd(gp_handle h) {
  if (!h) return GP_ERR_BAD_ARG;
  GP_CHECK(gp_sync(h));
  h->poll_inflight = false;
  if (h->poll_host) h->poll_host[0] = 0;
  return check_not_pd(h);
}

// Non-blocking look at the not-positive-definite flag: copies the device status word into pinned host memory behind
// everything enqueued so far and reports what the PREVIOUS poll's copy (complete by now or not yet) showed.  *flag = 1
// once a poll has seen a failed Cholesky; gp_check_not_pd / any host-scalar call then returns GP_ERR_NOT_PD as usual.
gp_status gp_poll_not_pd(gp_handle h, int32_t* flag) {
  if (!h || !flag) return GP_ERR_BAD_ARG;
  *flag = 0;
  if (!h->poll_host) {
    GP_HIP_CHECK(h, hipHostMalloc((void**)&h->poll_host, 4 * sizeof(int32_t), hipHostMallocDefault));
    memset(h->poll_host, 0, 4 * sizeof(int32_t));
    GP_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_poll, hipEventDisableTiming));
    h->poll_inflight = false;
  }
  if (h->poll_inflight) {
    const hipError_t q = hipEventQuery(h->ev_poll);
    if (q == hipSuccess) { h->poll_inflight = false; if (h->poll_host[0] != 0) { *flag = 1; return GP_OK; } }
    else if (q != hipErrorNotReady) { GP_HIP_CHECK(h, q); }
    else return GP_OK;          // the previous copy has not landed yet: nothing new to say, nothing new to enqueue
  }
  GP_HIP_CHECK(h, hipMemcpyAsync(h->poll_host, h->d_status, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  GP_HIP_CHECK(h, hipEventRecord(h->ev_poll, h->stream));
  h->poll_inflight = true;
  return GP_OK;
}

// Asynchronous hand-over of the not-positive-definite status word: behind everything enqueued so far on the handle's
// stream, copy {flag, pivot, matrix index, spare} into `host_status4` (pinned host memory of the caller) and clear the
// device word — so the NEXT evaluation on this stream reports its own failures only.  The caller reads host_status4 once
// an event recorded after this call has completed.  Used by the window-batched fit: one status snapshot per evaluation.
gp_status gp_take_not_pd(gp_handle h, int32_t* host_status4) {
  if (!h || !host_status4) return GP_ERR_BAD_ARG;
  GP_HIP_CHECK(h, hipMemcpyAsync(host_status4, h->d_status, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  GP_HIP_CHECK(h, hipMemsetAsync(h->d_status, 0, 4 * sizeof(int32_t), h->stream));
  return GP_OK;
}

gp_status gp_adam_step(gp_handle h, double* fs, double* params, const double* grad, const uint8_t* tcode, double* m,
                       double* v, int64_t n, int64_t t, double lr, double beta1, double beta2, double eps) {
  if (!h) return GP_ERR_BAD_ARG;
  if (!fs || !params || !grad || !tcode || !m || !v || n < 0 || t < 1)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_adam_step: bad argument");
  return launch_adam(h, fs, params, grad, tcode, m, v, n, t, lr, beta1, beta2, eps);
}

}  // extern "C"

// ---- run-time switches (switches.h): GPITCH_AMD_SWITCHES="name=value,..." read once --------------------------------------
const GpSwitches& gp_switches() {
  static const GpSwitches sw = []() {
    GpSwitches w;
    const char* e = getenv("GPITCH_AMD_SWITCHES");
    if (!e) return w;
    struct { const char* name; int* slot; } table[] = {
        {"strip_wave", &w.strip_wave}, {"strip_wave_f32", &w.strip_wave_f32}, {"strip_wave_roles", &w.strip_wave_roles}, {"strip_lean", &w.strip_lean},
        {"hyper_fuse", &w.hyper_fuse}, {"kufbar_split", &w.kufbar_split}, {"cond_a_early", &w.cond_a_early},
        {"blocked_256", &w.blocked_256}, {"cov_sum", &w.cov_sum}, {"hyper_sum", &w.hyper_sum}, {"chol_cluster", &w.chol_cluster}, {"aux_priority", &w.aux_priority}, {"gemm_tile32", &w.gemm_tile32}, {"nt_cover", &w.nt_cover}};
    std::string all(e);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos);
      if (end == std::string::npos) end = all.size();
      const std::string item = all.substr(pos, end - pos);
      pos = end + 1;
      const size_t eq = item.find('=');
      if (item.empty()) continue;
      bool known = false;
      if (eq != std::string::npos)
        for (auto& t : table)
          if (item.compare(0, eq, t.name) == 0 && strlen(t.name) == eq) { *t.slot = atoi(item.c_str() + eq + 1); known = true; }
      if (!known) fprintf(stderr, "libgpitch_hip: GPITCH_AMD_SWITCHES: unknown switch '%s' ignored (gpitch_amd/csrc/switches.h)\n", item.c_str());
    }
    return w;
  }();
  return sw;
}
