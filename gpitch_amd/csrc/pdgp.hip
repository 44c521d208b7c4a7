// pdgp.hip — the Pdgp model plan: ELBO forward / backward and predictions behind the C-ABI.
// Mirrors gpitch/pdgp.py:48-208 (Pdgp.build_prior_kl :113-131, build_likelihood :133-170,
// predict_act / predict_com / predict_act_n_com :172-208).
#include "pdgp_plan.h"
#include <string.h>


static size_t pdgp_bwd_doubles(const gp_pdgp_plan_s* p) {
  size_t d = 0;
  auto add = [&](size_t c) { d += gp_align_up(c * sizeof(double), 256) / sizeof(double); };
  for (int g = 0; g < p->G; g++) {
    const size_t M = p->gps[g].M;
    for (int i = 0; i < 6; i++) add(M * M);
    add(gp_strip_doubles(M, p->maxN, p->gps[g].f32 != 0));
    if (p->gps[g].f32) add((M * M + 1) / 2);
    add(M); add(M); add(M); add((size_t)65 * M);
    const size_t ns = hyper_num_sums(p->gps[g].m);
    const size_t colblocks = (p->maxN + 255) / 256 + 1;
    add(ns * hyper_kuf_records(p->maxN, (int)M));
    add(ns * hyper_kuf_records((int)M, (int)M));
    add(colblocks * M + ((M + 255) / 256 + 1) * M);
  }
  add(p->G + 8);
  if (!p->whiten) {
    for (int g = 0; g < p->G; g++) { const size_t M = p->gps[g].M; add(2 * (M + M * M) + 8); }
    add((size_t)p->G * GP_KL_BLOCKS);
  }
  return d;
}

extern "C" {

static gp_status pdgp_create_impl(gp_handle h, const gp_pdgp_config* cfg, const int32_t* gp_index, int32_t count,
                                  gp_pdgp_plan* out) {
  if (!h || !out) return GP_ERR_BAD_ARG;
  *out = nullptr;
  if (!cfg || cfg->num_sources < 1 || cfg->max_batch < 1 || !cfg->M_act || !cfg->M_com || !cfg->kern_type_act ||
      !cfg->kern_type_com || !cfg->partials_act || !cfg->partials_com || cfg->nlin < 0 || cfg->nlin > 2)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_create: bad config");
  if (gp_index) {
    if (count < 1 || count > 2 * cfg->num_sources) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_create_subset: bad count");
    for (int l = 0; l < count; l++)
      if (gp_index[l] < 0 || gp_index[l] >= 2 * cfg->num_sources || (l > 0 && gp_index[l] <= gp_index[l - 1]))
        return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_create_subset: gp_index must be strictly increasing in [0, 2 P)");
  }
  gp_pdgp_plan p = new gp_pdgp_plan_s();
  p->h = h; p->P = cfg->num_sources; p->G = gp_index ? count : 2 * p->P; p->whiten = cfg->whiten ? 1 : 0; p->nlin = cfg->nlin;
  p->maxN = cfg->max_batch; p->jitter = cfg->jitter;
  p->subset = gp_index != nullptr;
  if (gp_index) p->grow.assign(gp_index, gp_index + count);
  p->gps.resize(p->G);
  p->n64 = p->G;
  int64_t off = 1;  // [0] = noise variance
  for (int l = 0; l < p->G; l++) {
    PdgpGP& q = p->gps[l];
    const int g = gp_index ? gp_index[l] : l;      // row of the whole model's [g_0..g_{P-1}, f_0..f_{P-1}]
    const bool act = g < p->P;
    const int i = act ? g : g - p->P;
    q.M = act ? cfg->M_act[i] : cfg->M_com[i];
    q.ktype = act ? cfg->kern_type_act[i] : cfg->kern_type_com[i];
    q.m = act ? cfg->partials_act[i] : cfg->partials_com[i];
    const bool sm = gp_kern_has_partials(q.ktype);
    if (q.M < 1 || q.ktype < 0 || q.ktype > GP_KERN_LAST || (sm && (q.m < 1 || q.m > 32)) || (!sm && q.m != 0)) {
      delete p;
      return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_create: bad per-GP config");
    }
    q.off_theta = off; off += GP_THETA_LEN(q.m);
    q.off_z = off; off += q.M;
    q.off_qmu = off; off += q.M;
    // keep q_sqrt 16-byte aligned inside the vector so it can be a vectorised GEMM operand
    if (off & 1) off += 1;
    q.off_qsqrt = off; off += (int64_t)q.M * q.M;
    if (q.M > p->maxM) p->maxM = q.M;
    if (q.m > p->maxm) p->maxm = q.m;
  }
  p->nparams = off;
  *out = p;
  return GP_OK;
}

gp_status gp_pdgp_create(gp_handle h, const gp_pdgp_config* cfg, gp_pdgp_plan* out) {
  return pdgp_create_impl(h, cfg, nullptr, 0, out);
}

gp_status gp_pdgp_create_subset(gp_handle h, const gp_pdgp_config* cfg, const int32_t* gp_index, int32_t count,
                                gp_pdgp_plan* out) {
  if (!gp_index) return h ? gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_create_subset: gp_index is NULL") : GP_ERR_BAD_ARG;
  return pdgp_create_impl(h, cfg, gp_index, count, out);
}

gp_status gp_pdgp_destroy(gp_pdgp_plan p) {
  if (p && p->h && p->h->aux_stream) (void)hipStreamSynchronize(p->h->aux_stream);   // nothing of this plan still in flight
  if (p && p->h && p->h->side_stream) (void)hipStreamSynchronize(p->h->side_stream);
  delete p;
  return GP_OK;
}

gp_status gp_pdgp_set_grad_needs(gp_pdgp_plan p, int32_t g, int32_t need_theta, int32_t need_z) {
  if (!p || g < 0 || g >= p->G) return GP_ERR_BAD_ARG;
  if (p->gps[g].need_theta != (need_theta != 0) || p->gps[g].need_z != (need_z != 0)) {
    p->gps[g].need_theta = need_theta != 0;
    p->gps[g].need_z = need_z != 0;
    p->last_params = nullptr;   // force the backward descriptors to be rebuilt
  }
  return GP_OK;
}
gp_status gp_pdgp_set_precision(gp_pdgp_plan p, int32_t bits) {
  if (!p || (bits != 32 && bits != 64)) return GP_ERR_BAD_ARG;
  if (p->ws) return gp_fail(p->h, GP_ERR_BAD_ARG, "gp_pdgp_set_precision: call it before gp_pdgp_set_workspace");
  p->f32 = (bits == 32);
  p->cb.f32 = (bits == 32);
  for (auto& q : p->gps) q.f32 = p->f32;
  p->n64 = p->f32 ? 0 : p->G;
  return GP_OK;
}
// Per-GP strip precision: bits[g] in {32, 64} for the plan's latent GPs in engine order [g_0..g_{P-1}, f_0..f_{P-1}] (a subset
// plan: its own rows, same order); float64 GPs must precede float32 ones — the transcription model's use is "activation
// GPs (Matern-3/2 on a 16-kHz grid: cond(Kuu) ~ 1e9) float64, component GPs float32".
gp_status gp_pdgp_set_gp_precision(gp_pdgp_plan p, const int32_t* bits, int32_t count) {
  if (!p || !bits || count != p->G) return GP_ERR_BAD_ARG;
  if (p->ws) return gp_fail(p->h, GP_ERR_BAD_ARG, "gp_pdgp_set_gp_precision: call it before gp_pdgp_set_workspace");
  int n64 = 0;
  for (int g = 0; g < count; g++) {
    if (bits[g] != 32 && bits[g] != 64) return gp_fail(p->h, GP_ERR_BAD_ARG, "gp_pdgp_set_gp_precision: bits must be 32 or 64");
    if (bits[g] == 64) { if (n64 != g) return gp_fail(p->h, GP_ERR_UNSUPPORTED, "gp_pdgp_set_gp_precision: float64 latent GPs must precede float32 ones"); n64 = g + 1; }
  }
  for (int g = 0; g < count; g++) p->gps[g].f32 = (bits[g] == 32);
  p->n64 = n64;
  p->f32 = (n64 == 0);
  p->cb.f32 = (n64 == 0);
  return GP_OK;
}
gp_status gp_pdgp_set_overlap(gp_pdgp_plan p, int32_t level) {
  if (!p || level < 0 || level > 2) return GP_ERR_BAD_ARG;
  p->overlap = level;
  p->cb.overlap = (level >= 1);
  return GP_OK;
}
int64_t gp_pdgp_num_params(gp_pdgp_plan p) { return p ? p->nparams : 0; }

gp_status gp_pdgp_layout(gp_pdgp_plan p, int32_t g, int64_t* off_theta, int64_t* off_z, int64_t* off_qmu,
                         int64_t* off_qsqrt) {
  if (!p || g < 0 || g >= p->G) return GP_ERR_BAD_ARG;
  if (off_theta) *off_theta = p->gps[g].off_theta;
  if (off_z) *off_z = p->gps[g].off_z;
  if (off_qmu) *off_qmu = p->gps[g].off_qmu;
  if (off_qsqrt) *off_qsqrt = p->gps[g].off_qsqrt;
  return GP_OK;
}

static size_t pdgp_misc_bytes(const gp_pdgp_plan_s* p) {
  return 2 * pdgp_kl_region_bytes(p->G) + 24 * gp_align_up(p->G * sizeof(GemmProblem), 256) +
         gp_align_up(p->G * hyper_finish_item_bytes(), 256) + gp_align_up(2 * p->G * sizeof(HyperItem), 256);
}

size_t gp_pdgp_workspace_bytes(gp_pdgp_plan p) {
  if (!p) return 0;
  size_t d = 0;
  for (int g = 0; g < p->G; g++) d += cond_task_workspace_doubles(p->gps[g].M, p->maxN, p->gps[g].m, p->whiten != 0, p->gps[g].f32 != 0);
  auto addd = [&](size_t c) { d += gp_align_up(c * sizeof(double), 256) / sizeof(double); };
  for (int i = 0; i < 4; i++) addd((size_t)p->G * p->maxN);
  if (p->subset) for (int i = 0; i < 2; i++) addd((size_t)2 * p->P * p->maxN);
  addd((size_t)p->G * GP_KL_BLOCKS);
  addd(2 * (size_t)mpd_lik_blocks(p->maxN) + 8);
  if (!p->whiten)
    for (int g = 0; g < p->G; g++) addd((size_t)gemm_rowblocks(p->gps[g].M, 0) * p->gps[g].M);
  {
    d += pdgp_bwd_doubles(p);
    int ns = gemm_nt_nsplit(p->maxM, p->maxN, p->G);
    if (ns < 2) ns = 2;
    size_t slab = 0;
    for (int g = 0; g < p->G; g++) slab += gp_align_up((size_t)ns * p->gps[g].M * p->gps[g].M * sizeof(double), 256) / sizeof(double);
    d += slab;
  }
  return d * sizeof(double) + cond_batch_desc_bytes(p->G) + pdgp_misc_bytes(p) + 8192;
}

gp_status gp_pdgp_set_workspace(gp_pdgp_plan p, void* workspace, size_t bytes) {
  if (!p) return GP_ERR_BAD_ARG;
  if (!workspace || bytes < gp_pdgp_workspace_bytes(p) || (((uintptr_t)workspace) & 255))
    return gp_fail(p->h, GP_ERR_WORKSPACE, "gp_pdgp_set_workspace: workspace too small or not 256-byte aligned");
  p->ws = workspace; p->ws_bytes = bytes;
  GpArena ar(workspace, bytes);
  p->cb.tasks.assign(p->G, CondTask());
  p->cb.desc_bytes = cond_batch_desc_bytes(p->G);
  p->cb.d_desc = ar.take<char>(p->cb.desc_bytes);
  p->misc_bytes = pdgp_misc_bytes(p);
  p->d_misc = ar.take<char>(p->misc_bytes);
  p->fmean = ar.take<double>((size_t)p->G * p->maxN);
  p->fvar = ar.take<double>((size_t)p->G * p->maxN);
  p->gFmu = ar.take<double>((size_t)p->G * p->maxN);
  p->gFvar = ar.take<double>((size_t)p->G * p->maxN);
  if (p->subset) {
    p->gF_full_mu = ar.take<double>((size_t)2 * p->P * p->maxN);
    p->gF_full_var = ar.take<double>((size_t)2 * p->P * p->maxN);
  }
  p->kl = ar.take<double>((size_t)p->G * GP_KL_BLOCKS);
  p->lik_partials = ar.take<double>(2 * (size_t)mpd_lik_blocks(p->maxN) + 8);
  for (int g = 0; g < p->G; g++) {
    CondTask& t = p->cb.tasks[g];
    t.M = p->gps[g].M;
    t.kern = DevKern{p->gps[g].ktype, p->gps[g].m, nullptr};
    t.f32 = p->gps[g].f32 != 0;
    if (!cond_task_carve(ar, t, p->maxN, p->whiten != 0, t.f32)) return gp_fail(p->h, GP_ERR_WORKSPACE, "workspace carve failed");
  }
  p->bw.assign(p->G, BwdBufs());
  p->tr_part.assign(p->G, nullptr);
  if (!p->whiten)
    for (int g = 0; g < p->G; g++) p->tr_part[g] = ar.take<double>((size_t)gemm_rowblocks(p->gps[g].M, 0) * p->gps[g].M);
  {
    p->nsplit = gemm_nt_nsplit(p->maxM, p->maxN, p->G);
    if (p->nsplit < 2) p->nsplit = 2;
    for (int g = 0; g < p->G; g++) {
      const size_t M = p->gps[g].M;
      BwdBufs& b = p->bw[g];
      b.H = ar.take<double>(M * M); b.E = ar.take<double>(M * M); b.T1 = ar.take<double>(M * M);
      b.T2 = ar.take<double>(M * M); b.Wbar = ar.take<double>(M * M); b.R = ar.take<double>(M * M);
      b.G = ar.take<double>(gp_strip_doubles(M, p->maxN, p->gps[g].f32 != 0));
      b.R32 = p->gps[g].f32 ? ar.take<double>((M * M + 1) / 2) : nullptr;
      b.u = ar.take<double>(M); b.Lu = ar.take<double>(M); b.alpha = ar.take<double>(M);
      b.upart = ar.take<double>((size_t)p->nsplit * M);
      const size_t ns = hyper_num_sums(p->gps[g].m);
      const size_t colblocks = (p->maxN + 255) / 256 + 1;
      b.hyp_part = ar.take<double>(ns * hyper_kuf_records(p->maxN, (int)M));
      b.hyp_part_uu = ar.take<double>(ns * hyper_kuf_records((int)M, (int)M));
      b.gz_part = ar.take<double>(colblocks * M + ((M + 255) / 256 + 1) * M);
    }
    {  // sum_n gv per GP, contiguous so one launch fills all of them
      double* gvs = ar.take<double>(p->G + 8);
      for (int g = 0; g < p->G; g++) p->bw[g].gvsum = gvs + g;
    }
    size_t slab_total = 0;
    for (int g = 0; g < p->G; g++) slab_total += gp_align_up((size_t)p->nsplit * p->gps[g].M * p->gps[g].M * sizeof(double), 256) / sizeof(double);
    p->slabs = ar.take<double>(slab_total);
    if (!p->whiten) {
      auto ev = [](size_t c) { return (c + 1) & ~(size_t)1; };   // keep every piece 16-byte aligned
      size_t tot = 0;
      for (int g = 0; g < p->G; g++) { const size_t M = p->gps[g].M; tot += 2 * (ev(M) + ev(M * M)); }
      p->qw_block = ar.take<double>(tot); p->qw_doubles = tot;
      double* c = p->qw_block;
      for (int g = 0; g < p->G && c; g++) {
        const size_t M = p->gps[g].M;
        BwdBufs& b = p->bw[g];
        b.qmu_w = c; c += ev(M); b.Lq_w = c; c += ev(M * M); b.g_qmu_w = c; c += ev(M); b.g_Lq_w = c; c += ev(M * M);
      }
      p->kl_dummy = ar.take<double>((size_t)p->G * GP_KL_BLOCKS);
    }
  }
  if (!ar.ok) return gp_fail(p->h, GP_ERR_WORKSPACE, "gp_pdgp_set_workspace: arena exhausted");
  p->last_params = nullptr; p->last_n = -1;
  return GP_OK;
}

}  // extern "C"

gp_status pdgp_backward(gp_pdgp_plan p, const double* params, const double* x, int n, double* grad);  // bwd.hip
gp_status pdgp_prefetch_backward(gp_pdgp_plan p, int n, bool* kl_done);
gp_status pdgp_upload_bwd(gp_pdgp_plan p, const double* params, const double* x, int n, double* grad);

// (re)bind the parameter vector / batch to the device descriptors
static gp_status pdgp_bind(gp_pdgp_plan p, const double* params, const double* x, int n, double* grad,
                           double* fmean, double* fvar) {
  gp_handle h = p->h;
  // (the batch pointers x / y are not part of any descriptor: a fresh minibatch tensor every step must not
  //  force a re-upload)
  const bool same = (p->last_params == params && p->last_n == n && p->last_grad == grad &&
                     p->cb.tasks[0].fmean == fmean);
  if (same && p->cb.uploaded) return GP_OK;
  for (int g = 0; g < p->G; g++) {
    CondTask& t = p->cb.tasks[g];
    const PdgpGP& q = p->gps[g];
    t.kern.theta = params + q.off_theta;
    t.z = params + q.off_z;
    t.q_mu = params + q.off_qmu;
    t.q_sqrt = params + q.off_qsqrt;
    t.fmean = fmean + (size_t)g * n;
    t.fvar = fvar + (size_t)g * n;
  }
  p->cb.N = n;
  GP_CHECK(cond_batch_upload(h, p->cb, p->whiten != 0, p->jitter));
  // KL items
  p->h_misc.assign(p->misc_bytes, 0);
  p->off_kl_items = 0;
  for (int g = 0; g < p->G; g++) {
    const PdgpGP& q = p->gps[g];
    const CondTask& t = p->cb.tasks[g];
    if (p->whiten) {
      kl_item_fill(p->h_misc.data() + p->off_kl_items + g * kl_item_bytes(), params + q.off_qmu, params + q.off_qsqrt,
                   q.M, p->kl + (size_t)g * GP_KL_BLOCKS, grad ? grad + q.off_qmu : nullptr, grad ? grad + q.off_qsqrt : nullptr);
    } else {
      klu_item_fill(p->h_misc.data() + p->off_kl_items + g * klu_item_bytes(), params + q.off_qmu, params + q.off_qsqrt,
                    t.L, t.W, p->tr_part[g], gemm_rowblocks(q.M, 0), q.M, p->kl + (size_t)g * GP_KL_BLOCKS);
      // trace term: column sums of squares of W Lq (its own descriptor slot, after the backward ones)
      GemmProblem& r = *(GemmProblem*)(p->h_misc.data() + pdgp_kltr_offset(p->G) + g * sizeof(GemmProblem));
      memset(&r, 0, sizeof(r));
      r.A = t.W; r.lda = q.M; r.B = params + q.off_qsqrt; r.ldb = q.M; r.M = q.M; r.N = q.M; r.K = q.M; r.ldc = q.M;
      r.o0 = p->tr_part[g];
    }
  }
  if (grad) GP_CHECK(pdgp_upload_bwd(p, params, x, n, grad));
  GP_HIP_CHECK(h, hipMemcpyAsync(p->d_misc, p->h_misc.data(), p->misc_bytes, hipMemcpyHostToDevice, h->stream));
  p->last_params = params; p->last_x = x; p->last_n = n; p->last_grad = grad;
  return GP_OK;
}

extern "C" {

// Shared body of gp_pdgp_elbo (xchg == NULL, one call) and of the two-stage pitch-sharded form:
//   stage 1 (begin): conditionals, per-frame partial sums [A | B | D] and sum of the local KL terms -> xchg[0..3n]
//   stage 2 (end)  : likelihood + gradients from the rank-summed xchg, local backward pass.
static gp_status pdgp_forward(gp_pdgp_plan p, const double* params, const double* x, const double* y, int n,
                              double* grad, double* xchg, double* fmean = nullptr, double* fvar = nullptr) {
  gp_handle h = p->h;
  p->factor_valid = false;   // an optimiser step normally follows: predictions must re-factorise
  GP_CHECK(pdgp_bind(p, params, x, n, grad, fmean ? fmean : p->fmean, fvar ? fvar : p->fvar));
  if (grad) GP_HIP_CHECK(h, hipMemsetAsync(grad, 0, (size_t)p->nparams * sizeof(double), h->stream));
  GP_CHECK(cond_batch_run(h, p->cb, x, n, p->whiten != 0, p->jitter));
  bool kl_done = false;   // the whitened KL kernel went to the helper stream with the backward prefetch
  if (grad) GP_CHECK(pdgp_prefetch_backward(p, n, &kl_done));
  if (kl_done) GP_HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_era, 0));   // behind the forward strips: no stall
  if (xchg) {
    GP_CHECK(launch_mpd_lik(h, p->fmean, p->fvar, 1, n, y, n, p->P, p->nlin, params, 1.0, nullptr, nullptr, nullptr,
                            nullptr, nullptr, xchg, nullptr));
  }
  if (p->whiten) {
    if (!kl_done) GP_CHECK(launch_kl_white(h, p->d_misc + p->off_kl_items, p->G));
  } else {
    // gauss_kl(q_mu, q_sqrt, K = Kuu + jitter I) (pdgp.py:123-129): L and W of the conditional are reused
    // (one value per GP: the other partial-sum slots of the whitened layout stay zero)
    GP_HIP_CHECK(h, hipMemsetAsync(p->kl, 0, (size_t)p->G * GP_KL_BLOCKS * sizeof(double), h->stream));
    GemmFlags f;
    f.triA = TRI_LOWER; f.triB = TRI_LOWER; f.epilogue = EPI_COLSUMSQ;
    GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(p->d_misc + pdgp_kltr_offset(p->G)), p->G, p->maxM, p->maxM, f));
    GP_CHECK(launch_kl_unwhite(h, p->d_misc + p->off_kl_items, p->G));
  }
  if (xchg) GP_CHECK(launch_finish_sum(h, p->kl, p->G * GP_KL_BLOCKS, 1, 1, xchg + 3 * (size_t)n, 1.0, 0));
  return GP_OK;
}

// GP-sharded plan: rows grow[l] of the whole model's gradient arrays -> this plan's compact [G][n] arrays
struct RowGather { int rows[64]; int count; };
__global__ void __launch_bounds__(256) rows_gather_kernel(const double* __restrict__ src_mu, const double* __restrict__ src_var,
                                                          double* __restrict__ dst_mu, double* __restrict__ dst_var,
                                                          RowGather rg, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
  if (i >= n) return;
  const size_t s = (size_t)rg.rows[l] * n + i, d = (size_t)l * n + i;
  dst_mu[d] = src_mu[s];
  dst_var[d] = src_var[s];
}

// full_mu / full_var / kl_total != NULL: the GP-sharded second stage (moments of ALL 2 P latent GPs, [2 P][n], and the
// whole model's KL sum as one device double)
static gp_status pdgp_finish(gp_pdgp_plan p, const double* params, const double* x, const double* y, int n,
                             double num_data, const double* xchg, double* elbo_dev, double* elbo_host, double* grad,
                             const double* full_mu = nullptr, const double* full_var = nullptr,
                             const double* kl_total = nullptr) {
  gp_handle h = p->h;
  int nb = 0;
  const double scale = num_data / (double)n;
  if (full_mu) {
    GP_CHECK(launch_mpd_lik(h, full_mu, full_var, 1, n, y, n, p->P, p->nlin, params, scale, nullptr, p->lik_partials, &nb,
                            grad ? p->gF_full_mu : nullptr, grad ? p->gF_full_var : nullptr, nullptr, nullptr));
    if (grad) {
      for (int l0 = 0; l0 < p->G; l0 += 64) {
        RowGather rg;
        rg.count = p->G - l0 < 64 ? p->G - l0 : 64;
        for (int l = 0; l < rg.count; l++) rg.rows[l] = p->grow[l0 + l];
        hipLaunchKernelGGL(rows_gather_kernel, dim3((n + 255) / 256, rg.count), dim3(256), 0, h->stream, p->gF_full_mu,
                           p->gF_full_var, p->gFmu + (size_t)l0 * n, p->gFvar + (size_t)l0 * n, rg, n);
        GP_HIP_CHECK(h, hipGetLastError());
      }
    }
  } else {
    GP_CHECK(launch_mpd_lik(h, p->fmean, p->fvar, 1, n, y, n, p->P, p->nlin, params, scale, nullptr, p->lik_partials, &nb,
                            grad ? p->gFmu : nullptr, grad ? p->gFvar : nullptr, nullptr, xchg));
  }
  const double* kl_one = kl_total ? kl_total : (xchg ? xchg + 3 * (size_t)n : nullptr);
  if (grad) {     // (pdgp_backward launches it)
    p->fin.lik_partials = p->lik_partials; p->fin.nb = nb; p->fin.elbo = elbo_dev; p->fin.g_noise = grad; p->fin.pending = true;
    if (kl_one) { p->fin.kl = kl_one; p->fin.nkl = 1; } else { p->fin.kl = p->kl; p->fin.nkl = p->G * GP_KL_BLOCKS; }
    GP_CHECK(pdgp_backward(p, params, x, n, grad));
    if (p->fin.pending) return gp_fail(h, GP_ERR_HIP, "pdgp_backward left the ELBO reduction behind");
  } else if (kl_one) GP_CHECK(launch_elbo_finish(h, p->lik_partials, nb, kl_one, 1, elbo_dev, nullptr));
  else GP_CHECK(launch_elbo_finish(h, p->lik_partials, nb, p->kl, p->G * GP_KL_BLOCKS, elbo_dev, nullptr));
  if (elbo_host) {
    GP_HIP_CHECK(h, hipMemcpyAsync(elbo_host, elbo_dev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GP_CHECK(check_not_pd(h));
  }
  return GP_OK;
}

gp_status gp_pdgp_elbo(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                       double num_data, double* elbo_dev, double* elbo_host, double* grad) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_pdgp_elbo: workspace not set");
  if (!params || !x || !y || !elbo_dev || n < 1 || n > p->maxN) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_elbo: bad argument");
  if (p->subset) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_elbo: a GP-sharded plan is evaluated with gp_pdgp_cond_begin / _end");
  GP_CHECK(pdgp_forward(p, params, x, y, n, grad, nullptr));
  return pdgp_finish(p, params, x, y, n, num_data, nullptr, elbo_dev, elbo_host, grad);
}

gp_status gp_pdgp_elbo_begin(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                             double* grad, double* exchange) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_pdgp_elbo_begin: workspace not set");
  if (!params || !x || !y || !exchange || n < 1 || n > p->maxN) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_elbo_begin: bad argument");
  if (p->subset) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_elbo_begin: a GP-sharded plan is evaluated with gp_pdgp_cond_begin / _end");
  p->staged_n = 0;
  GP_CHECK(pdgp_forward(p, params, x, y, n, grad, exchange));
  p->staged_n = n; p->staged_grad = grad; p->staged_params = params;
  return GP_OK;
}

/* GP-sharded evaluation (SURVEY section 8e option 2): see include/gpitch_abi.h */
gp_status gp_pdgp_cond_begin(gp_pdgp_plan p, const double* params, const double* x, int32_t n, double* grad,
                             double* fmean_local, double* fvar_local, double* kl_local_sum) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_pdgp_cond_begin: workspace not set");
  if (!p->subset) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_cond_begin: not a gp_pdgp_create_subset plan");
  if (!params || !x || !fmean_local || !fvar_local || !kl_local_sum || n < 1 || n > p->maxN)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_cond_begin: bad argument");
  p->staged_n = 0;
  GP_CHECK(pdgp_forward(p, params, x, nullptr, n, grad, nullptr, fmean_local, fvar_local));
  GP_CHECK(launch_finish_sum(h, p->kl, p->G * GP_KL_BLOCKS, 1, 1, kl_local_sum, 1.0, 0));
  p->staged_n = n; p->staged_grad = grad; p->staged_params = params;
  return GP_OK;
}

gp_status gp_pdgp_cond_end(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                           double num_data, const double* fmean_full, const double* fvar_full, const double* kl_total,
                           double* elbo_dev, double* elbo_host, double* grad) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->subset) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_cond_end: not a gp_pdgp_create_subset plan");
  if (!params || !x || !y || !fmean_full || !fvar_full || !kl_total || !elbo_dev)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_cond_end: bad argument");
  if (p->staged_n != n || p->staged_grad != grad || p->staged_params != params)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_cond_end: no matching gp_pdgp_cond_begin (same params, n and grad required)");
  p->staged_n = 0;
  return pdgp_finish(p, params, x, y, n, num_data, nullptr, elbo_dev, elbo_host, grad, fmean_full, fvar_full, kl_total);
}

gp_status gp_pdgp_elbo_end(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                           double num_data, const double* exchange, double* elbo_dev, double* elbo_host, double* grad) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!params || !x || !y || !exchange || !elbo_dev) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_elbo_end: bad argument");
  if (p->staged_n != n || p->staged_grad != grad || p->staged_params != params)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_elbo_end: no matching gp_pdgp_elbo_begin (same params, n and grad required)");
  p->staged_n = 0;
  return pdgp_finish(p, params, x, y, n, num_data, exchange, elbo_dev, elbo_host, grad);
}

static gp_status pdgp_predict_impl(gp_pdgp_plan p, const double* params, const double* xnew, int32_t n, double* fmean,
                                   double* fvar, double* mean_source, bool reuse_factor) {
  if (!p) return GP_ERR_BAD_ARG;
  gp_handle h = p->h;
  if (!p->ws) return gp_fail(h, GP_ERR_WORKSPACE, "gp_pdgp_predict: workspace not set");
  if (!params || !xnew || !fmean || !fvar || n < 1 || n > p->maxN) return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_predict: bad argument");
  if (p->subset && mean_source)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_predict: a GP-sharded plan holds single latent GPs, not (activation, component) pairs: mean_source must be NULL");
  if (reuse_factor && !p->factor_valid)
    return gp_fail(h, GP_ERR_BAD_ARG, "gp_pdgp_predict_reuse: no factorisation to reuse (call gp_pdgp_predict first)");
  GP_CHECK(pdgp_bind(p, params, xnew, n, nullptr, fmean, fvar));
  GP_CHECK(cond_batch_run(h, p->cb, xnew, n, p->whiten != 0, p->jitter, reuse_factor));
  if (mean_source) GP_CHECK(launch_mean_source(h, fmean, p->P, n, p->nlin, mean_source));
  gp_status st = check_not_pd(h);
  p->factor_valid = (st == GP_OK);
  return st;
}

gp_status gp_pdgp_predict(gp_pdgp_plan p, const double* params, const double* xnew, int32_t n, double* fmean,
                          double* fvar, double* mean_source) {
  return pdgp_predict_impl(p, params, xnew, n, fmean, fvar, mean_source, false);
}

gp_status gp_pdgp_predict_reuse(gp_pdgp_plan p, const double* params, const double* xnew, int32_t n, double* fmean,
                                double* fvar, double* mean_source) {
  return pdgp_predict_impl(p, params, xnew, n, fmean, fvar, mean_source, true);
}

}  // extern "C"
