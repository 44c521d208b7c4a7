// engine.hip — host orchestration of a batch of sparse-GP conditionals (see engine.h).
// Follows GPflow-0.5 `conditionals.conditional` as called from gpitch/pdgp.py:147-155:
//   Kmm = K(z)+jitter I; Lm = chol; A = Lm^-1 Kmn; fvar = Kdiag - sum A^2; [A = Lm^-T A]; fmean = A^T f;
//   LTA = tril(q_sqrt)^T A; fvar += sum LTA^2.
#include "engine.h"
#include <string.h>

#define CB_NB 128          // panel width of the blocked Kuu factorisation
#define CB_MAX_PANELS 8    // M <= 1024

size_t cond_task_workspace_doubles(int M, int N, int m, bool whiten, bool f32) {
  const size_t strip = gp_strip_doubles((size_t)M, N, f32);
  const int rb = (M + 63) / 64;      // partial-sum rows: one per 64-row tile (gemm_wave.hip; the 128-row forms use half of them)
  size_t d = 0;
  auto add = [&](size_t c) { d += gp_align_up(c * sizeof(double), 256) / sizeof(double); };
  add((size_t)M * M); add((size_t)M * M);          // L, W
  add((size_t)CB_NB * M);                          // block-row scratch of the blocked inverse
  add(strip); add(strip);                          // Kuf, A
  if (!whiten) add(strip);                         // A2
  if (m > 0) { add(kernel_build_feat_ws_doubles(m, M, N)); add(kernel_build_feat_ws_doubles(m, M, M)); }
  add((size_t)rb * N); add((size_t)rb * N); add((size_t)rb * N);  // s1, s2, dot
  if (f32) { add(((size_t)M * M + 1) / 2); add(((size_t)M * M + 1) / 2); }   // float32 copies of W, tril(Lq)^T
  return d;
}

bool cond_task_carve(GpArena& ar, CondTask& t, int N, bool whiten, bool f32) {
  const size_t strip = gp_strip_doubles((size_t)t.M, N, f32);
  const int rb = (t.M + 63) / 64;
  t.L = ar.take<double>((size_t)t.M * t.M);
  t.W = ar.take<double>((size_t)t.M * t.M);
  t.Tblk = ar.take<double>((size_t)CB_NB * t.M);
  t.Kuf = ar.take<double>(strip);
  t.A = ar.take<double>(strip);
  t.A2 = whiten ? nullptr : ar.take<double>(strip);
  const bool mercer = (t.kern.m > 0 && gp_kern_is_mercer(t.kern.type));
  t.feat = mercer ? ar.take<double>(kernel_build_feat_ws_doubles(t.kern.m, t.M, N)) : nullptr;
  // the Kuu build has its own feature table: it runs concurrently with the Kuf build (cond_batch_run)
  t.feat_uu = mercer ? ar.take<double>(kernel_build_feat_ws_doubles(t.kern.m, t.M, t.M)) : nullptr;
  t.s1 = ar.take<double>((size_t)rb * N);
  t.s2 = ar.take<double>((size_t)rb * N);
  t.dot = ar.take<double>((size_t)rb * N);
  t.W32 = f32 ? ar.take<double>(((size_t)t.M * t.M + 1) / 2) : nullptr;
  t.Lq32 = f32 ? ar.take<double>(((size_t)t.M * t.M + 1) / 2) : nullptr;
  return ar.ok;
}

size_t cond_batch_desc_bytes(int count) {
  size_t b = 0;
  b += gp_align_up(count * sizeof(double*), 256) * 2;  // chol ptrs, W ptrs
  b += gp_align_up(count * sizeof(int), 256) * 2;      // Ms, lds
  b += gp_align_up(count * sizeof(GemmProblem), 256) * 3;
  b += gp_align_up(count * cond_finish_item_bytes(), 256);
  // grouped covariance builds: Kuu + Kuf items, and z (x2) / x feature items
  b += 2 * gp_align_up(count * sizeof(CovItem), 256) + 3 * gp_align_up(count * sizeof(FeatItem), 256);
  // blocked Cholesky + inverse (up to CB_MAX_PANELS 128-column panels): per panel 2 pointer arrays, 1 size array,
  // 4 GEMM problem arrays
  b += CB_MAX_PANELS * (2 * gp_align_up(count * sizeof(double*), 256) + gp_align_up(count * sizeof(int), 256) +
                        4 * gp_align_up(count * sizeof(GemmProblem), 256));
  // ... and the diagonal blocks of all panels as one batch (2 pointer arrays, sizes, leading dimensions)
  b += 2 * gp_align_up((size_t)CB_MAX_PANELS * count * sizeof(double*), 256) +
       2 * gp_align_up((size_t)CB_MAX_PANELS * count * sizeof(int), 256);
  return b;
}

gp_status cond_batch_upload(gp_handle h, CondBatch& cb, bool whiten, double jitter) {
  const int G = (int)cb.tasks.size();
  const int N = cb.N;
  // strip precision per task: a uniform batch (cb.f32) or float64 tasks first, float32 tasks behind them
  cb.n64 = 0;
  for (int g = 0; g < G; g++) {
    CondTask& t = cb.tasks[g];
    if (cb.f32) t.f32 = true;
    if (!t.f32) { if (cb.n64 != g) return gp_fail(h, GP_ERR_UNSUPPORTED, "per-GP precision: float64 latent GPs must precede float32 ones"); cb.n64 = g + 1; }
  }
  size_t need = cond_batch_desc_bytes(G);
  if (cb.desc_bytes < need || !cb.d_desc) return gp_fail(h, GP_ERR_WORKSPACE, "descriptor workspace too small");
  cb.h_desc.assign(need, 0);
  size_t off = 0;
  auto region = [&](size_t bytes) { size_t o = off; off += gp_align_up(bytes, 256); return o; };
  cb.off_chol_ptrs = region(G * sizeof(double*));
  cb.off_w_ptrs = region(G * sizeof(double*));
  cb.off_Ms = region(G * sizeof(int));
  cb.off_lds = region(G * sizeof(int));
  cb.off_f1 = region(G * sizeof(GemmProblem));
  cb.off_f1u = region(G * sizeof(GemmProblem));
  cb.off_f2 = region(G * sizeof(GemmProblem));
  cb.off_finish = region(G * cond_finish_item_bytes());
  double** cp = (double**)(cb.h_desc.data() + cb.off_chol_ptrs);
  double** wp = (double**)(cb.h_desc.data() + cb.off_w_ptrs);
  int* Ms = (int*)(cb.h_desc.data() + cb.off_Ms);
  int* lds = (int*)(cb.h_desc.data() + cb.off_lds);
  GemmProblem* f1 = (GemmProblem*)(cb.h_desc.data() + cb.off_f1);
  GemmProblem* f1u = (GemmProblem*)(cb.h_desc.data() + cb.off_f1u);
  GemmProblem* f2 = (GemmProblem*)(cb.h_desc.data() + cb.off_f2);
  char* fin = cb.h_desc.data() + cb.off_finish;
  cb.maxM = 0;
  for (int g = 0; g < G; g++) if (cb.tasks[g].M > cb.maxM) cb.maxM = cb.tasks[g].M;
  // which strip products of the float64 tasks take the wave form (one partial row per 64-row tile instead of per 128)
  const int uni = cond_batch_uniform(cb, N);
  cb.wave_a = gemm_wave_takes(1, cb.maxM, N, uni);
  cb.wave_lta = whiten && gemm_wave_takes(2, cb.maxM, N, uni);
  cb.wave_a32 = gemm_wave_f32_takes(1, cb.maxM, N, uni);
  cb.wave_lta32 = whiten && gemm_wave_f32_takes(2, cb.maxM, N, uni);
  for (int g = 0; g < G; g++) {
    const CondTask& t = cb.tasks[g];
    const int64_t ldN = gp_strip_ld(N, t.f32);
    cp[g] = t.L; wp[g] = t.W; Ms[g] = t.M; lds[g] = t.M;
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.A = t.W; p.lda = t.M; p.B = t.Kuf; p.ldb = ldN; p.C = t.A; p.ldc = ldN;
    p.M = t.M; p.N = N; p.K = t.M;
    p.v0 = t.q_mu; p.o0 = t.s1; p.o1 = t.dot; p.xb = t.W32;
    f1[g] = p;
    memset(&p, 0, sizeof(p));
    p.A = t.W; p.lda = t.M; p.B = t.A; p.ldb = ldN; p.C = t.A2; p.ldc = ldN;
    p.M = t.M; p.N = N; p.K = t.M;
    p.v0 = t.q_mu; p.o1 = t.dot;
    f1u[g] = p;
    memset(&p, 0, sizeof(p));
    p.A = t.q_sqrt; p.lda = t.M; p.B = whiten ? t.A : t.A2; p.ldb = ldN; p.C = nullptr; p.ldc = ldN;
    p.M = t.M; p.N = N; p.K = t.M;
    p.o0 = t.s2; p.xb = t.Lq32;
    f2[g] = p;
    const int rb = gemm_rowblocks(t.M, 1), rb64 = t.M / 64;
    const int rb_a = (t.f32 ? cb.wave_a32 : cb.wave_a) ? rb64 : rb, rb_lta = (t.f32 ? cb.wave_lta32 : cb.wave_lta) ? rb64 : rb;
    cond_finish_fill(fin + g * cond_finish_item_bytes(), t.s1, rb_a, t.s2, t.q_sqrt ? rb_lta : 0, t.dot,
                     whiten ? rb_a : rb, t.kern, t.fmean, t.fvar);
  }
  // Grouped covariance builds: the latent GPs are sorted into kernel families (type, padded partial count); each
  // family's Kuu (and Kuf) matrices are built by ONE launch (48 + 36 launches per step at P = 12 otherwise, which
  // left the device waiting for the host at the start of every step).
  {
    cb.groups.clear();
    std::vector<int> order;
    for (int g = 0; g < G; g++) {
      const CondTask& t = cb.tasks[g];
      const int key_m = gp_kern_has_partials(t.kern.type) ? t.kern.m : 0;
      int gi = -1;
      for (size_t q = 0; q < cb.groups.size(); q++)
        if (cb.groups[q].type == t.kern.type && cb.groups[q].m == key_m && cb.groups[q].f32 == t.f32) gi = (int)q;
      if (gi < 0) { CondBatch::Group ng; ng.type = t.kern.type; ng.m = key_m; ng.f32 = t.f32; cb.groups.push_back(ng); gi = (int)cb.groups.size() - 1; }
      cb.groups[gi].members.push_back(g);
    }
    cb.off_cov_uu = region(G * sizeof(CovItem));
    cb.off_cov_uf = region(G * sizeof(CovItem));
    cb.off_feat_zuu = region(G * sizeof(FeatItem));
    cb.off_feat_zuf = region(G * sizeof(FeatItem));
    cb.off_feat_x = region(G * sizeof(FeatItem));
    CovItem* uu = (CovItem*)(cb.h_desc.data() + cb.off_cov_uu);
    CovItem* uf = (CovItem*)(cb.h_desc.data() + cb.off_cov_uf);
    FeatItem* fzuu = (FeatItem*)(cb.h_desc.data() + cb.off_feat_zuu);
    FeatItem* fzuf = (FeatItem*)(cb.h_desc.data() + cb.off_feat_zuf);
    FeatItem* fx = (FeatItem*)(cb.h_desc.data() + cb.off_feat_x);
    int pos = 0;
    for (auto& gr : cb.groups) {
      gr.first = pos; gr.maxM = 0;
      for (int g : gr.members) {
        const CondTask& t = cb.tasks[g];
        if (t.M > gr.maxM) gr.maxM = t.M;
        cov_item_fill(&uu[pos], t.kern, t.z, t.M, nullptr, t.M, t.L, t.M, 0, jitter, t.feat_uu);
        cov_item_fill(&uf[pos], t.kern, t.z, t.M, nullptr, -1, t.Kuf, gp_strip_ld(N, t.f32), 0, 0.0, t.feat, t.f32 ? 1 : 0);
        fzuu[pos] = FeatItem{t.kern, t.z, t.feat_uu, t.M, 0};
        fzuf[pos] = FeatItem{t.kern, t.z, t.feat, t.M, 0};
        const int mp = sm_mpad(t.kern.m);
        fx[pos] = FeatItem{t.kern, nullptr, t.feat ? t.feat + gp_align_up((size_t)2 * mp * t.M, 32) : nullptr, -1, 0};
        pos++;
      }
    }
  }
  // Blocked factorisation of the Kuu batch (M > 256): 128-column panels; the diagonal blocks go to the one-workgroup
  // kernels (batched over the GPs), the O(M^3) panel solve / trailing update / block-row inverse to the batched GEMMs.
  // One workgroup per GP for the whole 512 x 512 factor + inverse took 1.8 ms of a 31 ms step.
  cb.nblk = (cb.maxM + CB_NB - 1) / CB_NB;
  // (two panels already pay off for long batches: the factor comes from one resident launch, the first row-block of
  // A = W Kuf starts from W's diagonal blocks — cond_batch_run — and the fused factor + inverse kernel was the head's
  // critical path: 0.46 ms at M = 256)
  const bool blk256 = gp_switches().blocked_256 != 0;
  cb.blocked = (cb.maxM > 256 || (blk256 && cb.maxM > 128 && cb.N >= 4096)) && cb.nblk <= CB_MAX_PANELS;
  if (cb.blocked) {
    for (int k = 0; k < cb.nblk; k++) {
      cb.off_blk_mats[k] = region(G * sizeof(double*));
      cb.off_blk_w[k] = region(G * sizeof(double*));
      cb.off_blk_M[k] = region(G * sizeof(int));
      for (int q = 0; q < 4; q++) cb.off_blk_gemm[k][q] = region(G * sizeof(GemmProblem));
      double** bm = (double**)(cb.h_desc.data() + cb.off_blk_mats[k]);
      double** bw = (double**)(cb.h_desc.data() + cb.off_blk_w[k]);
      int* bM = (int*)(cb.h_desc.data() + cb.off_blk_M[k]);
      GemmProblem* gp[4];
      for (int q = 0; q < 4; q++) gp[q] = (GemmProblem*)(cb.h_desc.data() + cb.off_blk_gemm[k][q]);
      const int c0 = k * CB_NB;
      for (int g = 0; g < G; g++) {
        const CondTask& t = cb.tasks[g];
        const int64_t ld = t.M;
        const int nb = (c0 >= t.M) ? 0 : ((t.M - c0 < CB_NB) ? t.M - c0 : CB_NB);
        const int r0 = c0 + nb, mrem = (nb > 0) ? t.M - r0 : 0;
        const int64_t dk = (nb > 0) ? (int64_t)c0 * ld + c0 : 0;
        bm[g] = t.L + dk; bw[g] = t.W + dk; bM[g] = nb;
        for (int q = 0; q < 4; q++) memset(&gp[q][g], 0, sizeof(GemmProblem));
        if (nb == 0) continue;
        { GemmProblem& r = gp[0][g];   // panel, in place: L[r0:, c0:c0+nb] <- L[r0:, c0:c0+nb] W_kk^T  (one 128-wide tile column)
          r.A = t.L + (int64_t)r0 * ld + c0; r.lda = ld; r.B = t.W + dk; r.ldb = ld; r.C = t.L + (int64_t)r0 * ld + c0; r.ldc = ld;
          r.M = mrem; r.N = nb; r.K = nb; }
        { GemmProblem& r = gp[1][g];   // trailing: L[r0:, r0:] -= P P^T (lower)
          r.A = t.L + (int64_t)r0 * ld + c0; r.lda = ld; r.B = r.A; r.ldb = ld; r.C = t.L + (int64_t)r0 * ld + r0; r.ldc = ld;
          r.M = mrem; r.N = mrem; r.K = nb; }
        { GemmProblem& r = gp[2][g];   // T = L[c0:c0+nb, :c0] W[:c0, :c0]   (128 x M scratch)
          r.A = t.L + (int64_t)c0 * ld; r.lda = ld; r.B = t.W; r.ldb = ld; r.C = t.Tblk; r.ldc = ld; r.M = nb; r.N = c0; r.K = c0; }
        { GemmProblem& r = gp[3][g];   // W[c0:c0+nb, :c0] = -W_kk T
          r.A = t.W + dk; r.lda = ld; r.B = t.Tblk; r.ldb = ld; r.C = t.W + (int64_t)c0 * ld; r.ldc = ld; r.M = nb; r.N = c0; r.K = nb; }
      }
    }
    // every panel's diagonal block of every GP as ONE batch (inverted in one launch after a whole-matrix factorisation)
    cb.off_diag_mats = region((size_t)cb.nblk * G * sizeof(double*));
    cb.off_diag_w = region((size_t)cb.nblk * G * sizeof(double*));
    cb.off_diag_M = region((size_t)cb.nblk * G * sizeof(int));
    cb.off_diag_ld = region((size_t)cb.nblk * G * sizeof(int));
    {
      double** dm = (double**)(cb.h_desc.data() + cb.off_diag_mats);
      double** dw = (double**)(cb.h_desc.data() + cb.off_diag_w);
      int* dM = (int*)(cb.h_desc.data() + cb.off_diag_M);
      int* dl = (int*)(cb.h_desc.data() + cb.off_diag_ld);
      for (int k = 0; k < cb.nblk; k++)
        for (int g = 0; g < G; g++) {
          const CondTask& t = cb.tasks[g];
          const int c0 = k * CB_NB;
          const int nb = (c0 >= t.M) ? 0 : ((t.M - c0 < CB_NB) ? t.M - c0 : CB_NB);
          const int64_t dk = (nb > 0) ? (int64_t)c0 * t.M + c0 : 0;
          dm[k * G + g] = t.L + dk; dw[k * G + g] = t.W + dk; dM[k * G + g] = nb; dl[k * G + g] = t.M;
        }
    }
    if (off > need) return gp_fail(h, GP_ERR_WORKSPACE, "descriptor workspace too small (blocked factorisation)");
  }
  GP_HIP_CHECK(h, hipMemcpyAsync(cb.d_desc, cb.h_desc.data(), need, hipMemcpyHostToDevice, h->stream));
  cb.uploaded = true;
  return GP_OK;
}

// blocks of W below the block diagonal, block row by block row: W[k, :c0] = -W_kk (L[k, :c0] W[:c0, :c0])
static gp_status cond_batch_block_row_inverse(gp_handle h, CondBatch& cb) {
  const int G = (int)cb.tasks.size();
  for (int k = 1; k < cb.nblk; k++) {
    const int c0 = k * CB_NB;
    GemmFlags f;
    f.triB = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(cb.d_desc + cb.off_blk_gemm[k][2]), G, CB_NB, c0, f));
    f = GemmFlags();
    f.triA = TRI_LOWER; f.alpha = -1.0;
    GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(cb.d_desc + cb.off_blk_gemm[k][3]), G, CB_NB, c0, f));
  }
  return GP_OK;
}

// Kuu_g -> L_g (lower, in place) and W_g = L_g^-1 for the whole batch.
// `resident`: device-filling kernels will run beside the factorisation (the Kuf strip builds of a long batch).  The
// panel-blocked path is the faster one on an otherwise idle device, but each of its ~25 launches asks again for
// whole CUs (512 threads x 256 VGPRs per workgroup) and gets them only when those kernels have drained.  So the
// factor comes from ONE launch, one workgroup per matrix, started before the strip builds and resident beside them;
// the inverse (by then the builds are ending) goes the blocked way: all diagonal 128-blocks in one launch, the
// blocks below them as batched GEMMs.
// the Kuu batch is one the workgroup-cluster factorisation takes (sizes whole tiles, few enough workgroups)
static bool cond_batch_cluster_shape(const CondBatch& cb, int* min_m) {
  int minM = cb.maxM;
  bool whole = true;
  for (const CondTask& t : cb.tasks) { minM = t.M < minM ? t.M : minM; whole = whole && (t.M % 32) == 0; }
  *min_m = minM;
  const int G = (int)cb.tasks.size();
  return whole && cholesky_cluster_takes(minM, G) && cholesky_cluster_takes(cb.maxM, G);
}

static gp_status cond_batch_factorize(gp_handle h, CondBatch& cb, bool resident) {
  const int G = (int)cb.tasks.size();
  {   // a few Kuu-sized matrices (one or two pitches: a pitch-sharded rank, BASELINE configs[1]; cfg3's 24 of 256 rows): a workgroup
      // cluster each, factor and inverse from one launch (chol_cluster.hip); the one-workgroup forms below keep G CUs busy for 0.7 ms
    int minM = cb.maxM;
    const bool whole = cond_batch_cluster_shape(cb, &minM);
    gp_status st = GP_OK;
    if (whole && launch_cholesky_cluster_batched(h, (double* const*)(cb.d_desc + cb.off_chol_ptrs), (double* const*)(cb.d_desc + cb.off_w_ptrs),
                                                 (const int*)(cb.d_desc + cb.off_Ms), (const int*)(cb.d_desc + cb.off_lds), G, minM,
                                                 cb.maxM, &st))
      return st;
  }
  if (!cb.blocked) {
    return launch_cholesky_inverse_batched(h, (double* const*)(cb.d_desc + cb.off_chol_ptrs),
                                           (double* const*)(cb.d_desc + cb.off_w_ptrs),
                                           (const int*)(cb.d_desc + cb.off_Ms), (const int*)(cb.d_desc + cb.off_lds), G,
                                           cb.maxM);
  }
  if (resident && cb.maxM <= 512) {
    GP_CHECK(launch_cholesky_batched(h, (double* const*)(cb.d_desc + cb.off_chol_ptrs),
                                     (const int*)(cb.d_desc + cb.off_Ms), (const int*)(cb.d_desc + cb.off_lds), G,
                                     cb.maxM));
    GP_CHECK(launch_zero_upper_blocks_batched(h, (double* const*)(cb.d_desc + cb.off_w_ptrs),
                                              (const int*)(cb.d_desc + cb.off_Ms), (const int*)(cb.d_desc + cb.off_lds), G,
                                              cb.maxM, CB_NB));
    GP_CHECK(launch_tri_inverse_batched(h, (const double* const*)(cb.d_desc + cb.off_diag_mats),
                                        (double* const*)(cb.d_desc + cb.off_diag_w),
                                        (const int*)(cb.d_desc + cb.off_diag_M), (const int*)(cb.d_desc + cb.off_diag_ld),
                                        cb.nblk * G));
    // the first row-block of A = W Kuf needs only W's first diagonal block: cond_batch_run starts it from here, underneath
    // the six dependent launches of the block-row inverse (0.2 ms of a few workgroups each)
    if (h->aux_active) {
      if (!h->ev_diag && hipEventCreateWithFlags(&h->ev_diag, hipEventDisableTiming) != hipSuccess) h->ev_diag = nullptr;
      if (h->ev_diag && hipEventRecord(h->ev_diag, h->stream) == hipSuccess) cb.diag_ready = true;
    }
    return cond_batch_block_row_inverse(h, cb);
  }
  const int* lds = (const int*)(cb.d_desc + cb.off_lds);
  // zero above the block diagonal of every W in one launch (the diagonal blocks are written whole, the blocks
  // below by the block-row inverse)
  GP_CHECK(launch_zero_upper_blocks_batched(h, (double* const*)(cb.d_desc + cb.off_w_ptrs),
                                            (const int*)(cb.d_desc + cb.off_Ms), lds, G, cb.maxM, CB_NB));
  for (int k = 0; k < cb.nblk; k++) {
    double* const* mats = (double* const*)(cb.d_desc + cb.off_blk_mats[k]);
    double* const* ws = (double* const*)(cb.d_desc + cb.off_blk_w[k]);
    const int* bM = (const int*)(cb.d_desc + cb.off_blk_M[k]);
    const int c0 = k * CB_NB, rem = cb.maxM - c0 - CB_NB;
    GP_CHECK(launch_cholesky_batched(h, mats, bM, lds, G, CB_NB, c0));
    GP_CHECK(launch_tri_inverse_batched(h, (const double* const*)mats, ws, bM, lds, G));
    if (rem > 0) {
      GemmFlags f;
      f.transB = 1; f.triB = TRI_UPPER; f.big_tiles = 1;   // 128-wide tiles: each workgroup reads exactly the rows it rewrites
      GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(cb.d_desc + cb.off_blk_gemm[k][0]), G, rem, CB_NB, f));
      f = GemmFlags();
      f.transB = 1; f.triC = TRI_LOWER; f.alpha = -1.0; f.beta = 1.0;
      GP_CHECK(launch_gemm_batched(h, (const GemmProblem*)(cb.d_desc + cb.off_blk_gemm[k][1]), G, rem, rem, f));
    }
  }
  GP_CHECK(launch_zero_upper_blocks_batched(h, (double* const*)(cb.d_desc + cb.off_chol_ptrs),
                                            (const int*)(cb.d_desc + cb.off_Ms), lds, G, cb.maxM, CB_NB));
  return cond_batch_block_row_inverse(h, cb);
}

// every task the same M (so every strip problem is maxM x N), q_sqrt 16-byte aligned: what gemm_strip.hip's form assumes
// (W, Kuf, A come out of the 256-byte-aligned arena with even leading dimensions)
int cond_batch_uniform(const CondBatch& cb, int N) {
  if (cb.tasks.empty() || (N & 1)) return 0;
  for (const CondTask& t : cb.tasks)
    if (t.M != cb.maxM || (t.q_sqrt && (((uintptr_t)t.q_sqrt) & 15))) return 0;
  return 1;
}

gp_status cond_batch_run(gp_handle h, CondBatch& cb, const double* x, int N, bool whiten, double jitter,
                         bool reuse_factor) {
  const int G = (int)cb.tasks.size();
  if (G == 0 || N <= 0) return GP_OK;
  if (!cb.uploaded || cb.N != N) return gp_fail(h, GP_ERR_BAD_ARG, "conditional batch descriptors not uploaded");
  // 1-2. Kuu + jitter I, its Cholesky factor and inverse — skipped when the caller vouches that L and W already hold
  //    the factor of the current parameters (repeated predictions at new inputs, pdgp.py:17-44 predict_windowed).
  //    The factorisation is latency-bound on ~G CUs, so it runs on the handle's helper stream while the main stream
  //    builds the Kuf strips (HBM / VALU-bound on all CUs); they meet again before A = W Kuf.
  // The host feeds both queues alternately (a Kuu build for the helper stream, a Kuf build for the main stream):
  // issuing all of one stream's launches first left the other queue waiting for the host for ~0.6 ms.
  const bool resident = (N >= 4096);   // long batch: the strip builds fill the device (independent of the overlap level)
  const bool forked = !reuse_factor && (N >= 4096) && cb.overlap && gp_aux_fork(h);
  // one launch per kernel family for the Kuu builds (helper stream when forked) ...
  auto build_kuu = [&]() -> gp_status {
    for (const auto& gr : cb.groups) {
      const int cnt = (int)gr.members.size();
      if (gp_kern_is_mercer(gr.type))
        GP_CHECK(launch_sm_features_items(h, (const FeatItem*)(cb.d_desc + cb.off_feat_zuu) + gr.first, cnt, gr.maxM,
                                          sm_mpad(gr.m), nullptr, 0));
      GP_CHECK(launch_kernel_build_items(h, gr.type, gr.m, (const CovItem*)(cb.d_desc + cb.off_cov_uu) + gr.first, cnt,
                                         gr.maxM, gr.maxM, nullptr, 0));
    }
    return GP_OK;
  };
  // ... and for the Kuf strips (main stream): every item shares the frames x
  auto build_kuf = [&]() -> gp_status {
    for (const auto& gr : cb.groups) {
      const int cnt = (int)gr.members.size();
      if (gp_kern_is_mercer(gr.type)) {
        const int mp = sm_mpad(gr.m);
        GP_CHECK(launch_sm_features_items(h, (const FeatItem*)(cb.d_desc + cb.off_feat_zuf) + gr.first, cnt, gr.maxM, mp,
                                          nullptr, 0));
        GP_CHECK(launch_sm_features_items(h, (const FeatItem*)(cb.d_desc + cb.off_feat_x) + gr.first, cnt, N, mp, x, N));
      }
      GP_CHECK(launch_kernel_build_items(h, gr.type, gr.m, (const CovItem*)(cb.d_desc + cb.off_cov_uf) + gr.first, cnt,
                                         gr.maxM, N, x, N, gr.f32 ? 2 : 1));     // 2: the strips are float32
    }
    return GP_OK;
  };
  // a strip product over the batch: the float64 tasks [0, n64) on the float64 kernels, the float32 tasks behind them on the
  // float32 ones (f32flags: the flags of the float32 launch where they differ)
  auto strips = [&](size_t off, const GemmFlags& f, const GemmFlags* f32flags = nullptr) -> gp_status {
    const GemmProblem* d = (const GemmProblem*)(cb.d_desc + off);
    const int n64 = (cb.n64 < 0) ? (cb.f32 ? 0 : G) : cb.n64;
    if (n64 > 0) GP_CHECK(launch_gemm_batched(h, d, n64, cb.maxM, N, f));
    if (n64 < G) {
      GemmFlags g32 = f32flags ? *f32flags : f;
      // (float32 tasks: their own wave form — scratch for the float32 copy of the M x M operand is in xb; partial rows per 64-row tile)
      g32.a32_ok = 1;
      g32.rows64_ok = (g32.role == 1) ? (cb.wave_a32 ? 1 : 0) : (g32.role == 2 && !f32flags) ? (cb.wave_lta32 ? 1 : 0) : 0;
      GP_CHECK(launch_gemm_f32_role(h, d + n64, G - n64, cb.maxM, N, g32));
    }
    return GP_OK;
  };
  gp_status st = GP_OK;
  cb.diag_ready = false;
  if (!reuse_factor) {
    st = build_kuu();
    // The Kuf builds (main stream) start only once the Kuu builds are through, i.e. together with the factorisation
    // launch: its workgroups must be on their CUs before the strip builds fill the device.
    // (not with the cluster factorisation: its launch is short and early builds pay more than its workgroups' wait costs —
    // cfg3 3.82 -> 3.76 ms; with the one-workgroup kernels the wait stays: headline 19.4 against 19.4 - 19.7 without it)
    int min_m_unused = 0;
    if (forked && resident && st == GP_OK && !cond_batch_cluster_shape(cb, &min_m_unused)) {
      if (!h->ev_kuu && hipEventCreateWithFlags(&h->ev_kuu, hipEventDisableTiming) != hipSuccess) h->ev_kuu = nullptr;
      if (h->ev_kuu && hipEventRecord(h->ev_kuu, h->stream) == hipSuccess)
        (void)hipStreamWaitEvent(h->main_stream_saved, h->ev_kuu, 0);
    }
    if (st == GP_OK) st = cond_batch_factorize(h, cb, resident);
  }
  if (forked) { gp_status s2 = gp_aux_end(h); if (st == GP_OK) st = s2; }
  GP_CHECK(st);
  // 3. Kuf
  GP_CHECK(build_kuf());
  // 4. A = W Kuf (+ column reductions).  When the factorisation left an event behind the diagonal blocks of W, the first
  // row-block goes ahead of the join (it reads W[0:128, 0:128] only) and the rest follows it.
  {
    GemmFlags f;
    f.triA = TRI_LOWER; f.big_tiles = 1; f.timer = GP_TIMER_COND_A; f.role = 1;
    f.epilogue = EPI_STORE | EPI_COLSUMSQ | (whiten ? EPI_COLDOT : 0);
    f.uniform_aligned = cond_batch_uniform(cb, N);
    f.rows64_ok = cb.wave_a ? 1 : 0;
    const bool early_ok = gp_switches().cond_a_early != 0;
    const bool early = early_ok && forked && cb.diag_ready && cb.maxM > 128 &&
                       hipStreamWaitEvent(h->stream, h->ev_diag, 0) == hipSuccess;
    auto cond_a = [&](int m0, int mcount) -> gp_status {
      GemmFlags g = f;
      g.tile_m0 = m0; g.tile_mcount = mcount;
      return strips(cb.off_f1, g);
    };
    if (early) {
      gp_status s1 = cond_a(0, 1);
      gp_status s2 = gp_aux_join(h);
      GP_CHECK(s1); GP_CHECK(s2);
      GP_CHECK(cond_a(1, 0));
    } else {
      GP_CHECK(gp_aux_join(h));
      GP_CHECK(cond_a(0, 0));
    }
  }
  if (!whiten) {
    GemmFlags f;
    f.transA = 1; f.triA = TRI_UPPER; f.big_tiles = 1; f.timer = GP_TIMER_COND_A;
    f.epilogue = EPI_STORE | EPI_COLDOT;
    // float32 strips: the same product shape as Lq^T A (op(A) = W^T upper, read row-wise), stored: role 2 of gemm_f32.hip
    GemmFlags f3 = f; f3.role = 2;
    GP_CHECK(strips(cb.off_f1u, f, &f3));
  }
  // 5. LTA = tril(q_sqrt)^T A  (only its column sums of squares are needed)
  bool any_qsqrt = false;
  for (int g = 0; g < G; g++) any_qsqrt |= (cb.tasks[g].q_sqrt != nullptr);
  if (any_qsqrt) {
    for (int g = 0; g < G; g++)
      if (!cb.tasks[g].q_sqrt) return gp_fail(h, GP_ERR_UNSUPPORTED, "mixed null/non-null q_sqrt in one batch");
    GemmFlags f;
    f.transA = 1; f.triA = TRI_UPPER; f.big_tiles = 1; f.timer = GP_TIMER_COND_LTA; f.role = 2;
    f.epilogue = EPI_COLSUMSQ;
    f.uniform_aligned = whiten ? cond_batch_uniform(cb, N) : 0;
    f.rows64_ok = cb.wave_lta ? 1 : 0;
    GP_CHECK(strips(cb.off_f2, f));
  }
  // 6. fmean / fvar
  GP_CHECK(launch_cond_finish(h, cb.d_desc + cb.off_finish, G, N));
  return GP_OK;
}
