// lik.hip — modulated-GP likelihood: Gauss-Hermite variational expectations, their gradients,
// the whitened KL term and the small vector kernels that stitch the conditional together (gfx950).
//
// Replaces  MpdLik.variational_expectations  gpitch/likelihoods.py:422-447
//           hermgauss1d                      gpitch/likelihoods.py:33-45   (20-point Gauss-Hermite)
//           log_lik_exp                      gpitch/likelihoods.py:47-68
//           gauss_kl (whitened)              gpitch/pdgp.py:120-121 (GPflow 0.5 kullback_leiblers)
//           nonlinearities                   gpitch/methods.py:216-233
// Sixteen lanes per audio frame (one source each), reads of the 4P conditional moments,
// block partial sums written to a scratch vector and finished in a fixed order (reproducible ELBO).
#include "common.h"

#define LIK_THREADS 256
#define GH_POINTS 20

// numpy.polynomial.hermite.hermgauss(20) as exact float64 hex literals (data of the reference's
// algorithm: gpflow.quadrature.hermgauss at likelihoods.py:35)
__constant__ double c_gh_x[GH_POINTS] = {
    -0x1.58cc7ca59b160p+2, -0x1.26a2bbb67f55ep+2, -0x1.f8ee072f5de17p+1, -0x1.ac867f9b566b1p+1,
    -0x1.64f798cfeaf13p+1, -0x1.20a2fcf426dedp+1, -0x1.bd10ceb867454p+0, -0x1.3bec6b39e4f51p+0,
    -0x1.7996281385f71p-1, -0x1.f67530743d203p-3, 0x1.f67530743d203p-3, 0x1.7996281385f71p-1,
    0x1.3bec6b39e4f51p+0, 0x1.bd10ceb867454p+0, 0x1.20a2fcf426dedp+1, 0x1.64f798cfeaf13p+1,
    0x1.ac867f9b566b1p+1, 0x1.f8ee072f5de17p+1, 0x1.26a2bbb67f55ep+2, 0x1.58cc7ca59b160p+2};
// hermgauss weights / sqrt(pi) (likelihoods.py:37)
__constant__ double c_gh_w[GH_POINTS] = {
    0x1.1b3b45ae1f142p-43, 0x1.10e7d83542f1ap-32, 0x1.072c77c84087ep-24, 0x1.276bdd4d669f2p-18,
    0x1.0e2b15190024dp-13, 0x1.dfc024629beb1p-10, 0x1.caae5f0667278p-7, 0x1.f7dc3610551aep-5,
    0x1.4b3dfdef813b4p-3, 0x1.0b0d563a28706p-2, 0x1.0b0d563a28706p-2, 0x1.4b3dfdef813b4p-3,
    0x1.f7dc3610551aep-5, 0x1.caae5f0667278p-7, 0x1.dfc024629beb1p-10, 0x1.0e2b15190024dp-13,
    0x1.276bdd4d669f2p-18, 0x1.072c77c84087ep-24, 0x1.10e7d83542f1ap-32, 0x1.1b3b45ae1f142p-43};

__device__ __forceinline__ void nlin_eval(int nlin, double x, double& s, double& ds) {
  const double PI = 3.141592653589793;
  if (nlin == GP_NLIN_LOGISTIC) {          // methods.py:216-218
    s = 1.0 / (1.0 + exp(-2.0 * (x - PI)));
    ds = 2.0 * s * (1.0 - s);
  } else if (nlin == GP_NLIN_SOFTPLUS) {   // methods.py:220-222 (naive form, as the reference)
    s = log(exp(x) + 1.0);
    ds = 1.0 / (1.0 + exp(-x));
  } else {                                 // methods.py:232-233
    double d = x - PI;
    s = exp(-2.0 * d * d);
    ds = -4.0 * d * s;
  }
}

struct Quad {
  double E1, E2, dE1m, dE1s, dE2m, dE2s;
};

__device__ __forceinline__ Quad gh_quad(int nlin, double mg, double vg, bool want_grad) {
  Quad q = {0, 0, 0, 0, 0, 0};
  const double sd = sqrt(2.0 * vg);
#pragma unroll 4
  for (int hh = 0; hh < GH_POINTS; hh++) {
    const double xh = c_gh_x[hh], wh = c_gh_w[hh];
    double s, ds;
    nlin_eval(nlin, xh * sd + mg, s, ds);
    q.E1 = fma(s, wh, q.E1);
    q.E2 = fma(s * s, wh, q.E2);
    if (want_grad) {
      const double t1 = wh * ds, t2 = 2.0 * wh * s * ds;
      q.dE1m += t1;
      q.dE1s = fma(t1, xh, q.dE1s);
      q.dE2m += t2;
      q.dE2s = fma(t2, xh, q.dE2s);
    }
  }
  return q;
}

// Sixteen lanes per frame, one source (pitch) per lane (sources beyond 16 go round again): the 20-point
// quadrature of a source — all the exp() work — is per (frame, source), while a frame's sums over its sources are
// a handful of multiply-adds.  Each lane parks its a_i = E1_i m_f_i, E2_i and v_f_i + m_f_i^2 in LDS, then EVERY
// lane of the frame replays the reference's sequential accumulation over i = 0..P-1 (same order, same fused
// operations as a one-thread-per-frame loop), so the result does not depend on how sources are dealt to lanes.
// Fmu/Fvar element (n, c) at [n * rs + c * cs]; columns [g_0..g_{P-1}, f_0..f_{P-1}].
// partial[2*blk] = sum varexp * scale ; partial[2*blk+1] = sum d(varexp*scale)/d noise_var
#define LIK_LANES 16
#define LIK_FRAMES (LIK_THREADS / LIK_LANES)
int mpd_lik_blocks(int N) { return (N + LIK_FRAMES - 1) / LIK_FRAMES; }

__global__ void __launch_bounds__(LIK_THREADS) mpd_lik_kernel(const double* __restrict__ Fmu, const double* __restrict__ Fvar,
                                                              int64_t rs, int64_t cs, const double* __restrict__ y, int N,
                                                              int P, int nlin, const double* __restrict__ noise_var,
                                                              double scale, double* __restrict__ per_frame,
                                                              double* __restrict__ partial, double* __restrict__ gFmu,
                                                              double* __restrict__ gFvar, double* __restrict__ psum,
                                                              const double* __restrict__ gsum) {
  // Pitch-sharded operation (the P sources here are one rank's share of a larger model):
  //   psum != NULL : write this rank's per-frame partial sums  A = sum a_i, B = sum E2_i(v_f+m_f^2), D = sum a_i^2
  //                  to psum[0..N), psum[N..2N), psum[2N..3N) and stop;
  //   gsum != NULL : the same three vectors summed over all ranks; the cross term becomes C = A^2 - D
  //                  (likelihoods.py:56-65 builds the same quantity as an explicit pair sum).
  extern __shared__ double lik_sm[];                       // [LIK_FRAMES][3][P]: a_i | E2_i | v_f_i + m_f_i^2
  const int fl = threadIdx.x / LIK_LANES, l = threadIdx.x % LIK_LANES;
  const int n = blockIdx.x * LIK_FRAMES + fl;
  double* sa = lik_sm + (size_t)fl * 3 * P;
  double* se = sa + P;
  double* sc = se + P;
  const bool live = (n < N);
  const bool need_pass1 = (!gsum || psum);
  const bool need_grad = (gFmu != nullptr) && !psum;
  const double s2 = noise_var[0];
  Quad q0 = {0, 0, 0, 0, 0, 0};      // this lane's first source, kept for the gradient pass
  bool cached = false;
  if (live && need_pass1) {
    for (int i = l; i < P; i += LIK_LANES) {
      const double mg = Fmu[n * rs + i * cs], vg = Fvar[n * rs + i * cs];
      const double mf = Fmu[n * rs + (i + P) * cs], vf = Fvar[n * rs + (i + P) * cs];
      const bool keep = need_grad && (i == l);
      Quad q = gh_quad(nlin, mg, vg, keep);
      if (keep) { q0 = q; cached = true; }
      sa[i] = q.E1 * mf;
      se[i] = q.E2;
      sc[i] = vf + mf * mf;
    }
  }
  __syncthreads();
  double ve = 0.0, dnoise = 0.0;
  if (live) {
    const double Y = y[n];
    // A, B, C (C as the reference's pair sum 2*sum_{i<j} a_i a_j, accumulated with a running prefix)
    double A = 0.0, B = 0.0, Cpair = 0.0, D = 0.0;
    if (need_pass1) {
      for (int i = 0; i < P; i++) {
        const double a = sa[i];
        Cpair = fma(a, A, Cpair);  // a_i * sum_{j<i} a_j
        D = fma(a, a, D);
        A += a;
        B = fma(se[i], sc[i], B);
      }
    }
    double C = 2.0 * Cpair;
    if (psum && l == 0) {
      psum[n] = A; psum[(int64_t)N + n] = B; psum[2 * (int64_t)N + n] = D;
    }
    if (gsum) {
      A = gsum[n]; B = gsum[(int64_t)N + n];
      C = A * A - gsum[2 * (int64_t)N + n];
    }
    const double resid = Y * Y - 2.0 * Y * A + B + C;
    const double LOG2PI = 1.8378770664093453;
    const double v = -0.5 * ((1.0 / s2) * resid + LOG2PI + log(s2));
    if (per_frame && l == 0) per_frame[n] = v;
    if (l == 0) ve = v * scale;
    if (need_grad) {
      if (l == 0) dnoise = scale * (0.5 * resid / (s2 * s2) - 0.5 / s2);
      const double qf = -0.5 * scale / s2;
      for (int i = l; i < P; i += LIK_LANES) {
        const double mg = Fmu[n * rs + i * cs], vg = Fvar[n * rs + i * cs];
        const double mf = Fmu[n * rs + (i + P) * cs], vf = Fvar[n * rs + (i + P) * cs];
        const Quad q = (cached && i == l) ? q0 : gh_quad(nlin, mg, vg, true);
        const double a = q.E1 * mf;
        const double da = qf * (-2.0 * Y + 2.0 * (A - a));  // d/d a_i
        const double dE1 = da * mf;
        const double dE2 = qf * (vf + mf * mf);
        const double sd = sqrt(2.0 * vg);
        const double inv_sd = sd > 0.0 ? 1.0 / sd : 0.0;
        gFmu[n * rs + i * cs] = dE1 * q.dE1m + dE2 * q.dE2m;
        gFvar[n * rs + i * cs] = (dE1 * q.dE1s + dE2 * q.dE2s) * inv_sd;
        gFmu[n * rs + (i + P) * cs] = da * q.E1 + qf * q.E2 * 2.0 * mf;
        gFvar[n * rs + (i + P) * cs] = qf * q.E2;
      }
    }
  }
  // block reduction (fixed tree)
  __shared__ double red[2][LIK_THREADS / 64];
  for (int o = 32; o > 0; o >>= 1) { ve += __shfl_down(ve, o, 64); dnoise += __shfl_down(dnoise, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ve; red[1][threadIdx.x >> 6] = dnoise; }
  __syncthreads();
  if (threadIdx.x == 0 && partial) {
    double a = 0, b = 0;
    for (int w = 0; w < LIK_THREADS / 64; w++) { a += red[0][w]; b += red[1][w]; }
    partial[2 * blockIdx.x] = a;
    partial[2 * blockIdx.x + 1] = b;
  }
}

gp_status launch_mpd_lik(gp_handle h, const double* Fmu, const double* Fvar, int64_t f_rs, int64_t f_cs,
                         const double* y, int N, int P, int nlin, const double* noise_var, double scale,
                         double* per_frame, double* partial_sums, int* num_partials_out, double* gFmu,
                         double* gFvar, double* psum, const double* gsum) {
  if (N <= 0) { if (num_partials_out) *num_partials_out = 0; return GP_OK; }
  GpTimerScope ts(h, GP_TIMER_LIK);
  const int blocks = mpd_lik_blocks(N);
  const size_t sh = (size_t)LIK_FRAMES * 3 * P * sizeof(double);
  if (sh > 48 * 1024) return gp_fail(h, GP_ERR_UNSUPPORTED, "too many sources for the likelihood kernel's LDS staging");
  hipLaunchKernelGGL(mpd_lik_kernel, dim3(blocks), dim3(LIK_THREADS), sh, h->stream, Fmu, Fvar, f_rs, f_cs, y, N, P,
                     nlin, noise_var, scale, per_frame, partial_sums, gFmu, gFvar, psum, gsum);
  GP_HIP_CHECK(h, hipGetLastError());
  if (num_partials_out) *num_partials_out = blocks;
  return GP_OK;
}

// out[s] (+)= mul * sum_{c < count} partials[c * stride + s], s < nsums; one block, fixed order
__global__ void __launch_bounds__(256) finish_sum_kernel(const double* __restrict__ partials, int count, int stride,
                                                         int nsums, double* __restrict__ out, double mul,
                                                         int accumulate) {
  __shared__ double red[256];
  for (int s = 0; s < nsums; s++) {
    double a = 0.0;
    for (int c = threadIdx.x; c < count; c += 256) a += partials[(int64_t)c * stride + s];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[s] = (accumulate ? out[s] : 0.0) + mul * red[0];
    __syncthreads();
  }
}

gp_status launch_finish_sum(gp_handle h, const double* partials, int count, int stride, int nsums, double* out,
                            double mul, int accumulate) {
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, h->stream, partials, count, stride, nsums, out, mul,
                     accumulate);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// ---------------------------------------------------------------------------------------------
// conditional finish: fmean[n] = sum_rb dot[rb][n]; fvar[n] = kdiag - sum_rb s1[rb][n] + sum_rb s2[rb][n]
struct CondFinish {
  const double* s1; const double* s2; const double* dot;  // partials [rowblocks][N]
  int rb1, rb2, rbdot;                                    // number of row-block partials in each
  DevKern kern;
  double* fmean; double* fvar;                            // N each
};

__global__ void __launch_bounds__(256) cond_finish_kernel(const CondFinish* __restrict__ items, int N) {
  const CondFinish it = items[blockIdx.y];
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const double* th = it.kern.theta;
  double kd = th[0];
  if (gp_kern_kdiag_energy(it.kern.type)) {
    double s = th[2];
    for (int p = 1; p < it.kern.m; p++) s += th[2 + p];
    kd = kd * s;
  }
  double a = 0.0, b = 0.0, d = 0.0;
  for (int r = 0; r < it.rb1; r++) a += it.s1[(int64_t)r * N + n];
  for (int r = 0; r < it.rb2; r++) b += it.s2[(int64_t)r * N + n];
  for (int r = 0; r < it.rbdot; r++) d += it.dot[(int64_t)r * N + n];
  it.fmean[n] = d;
  it.fvar[n] = (kd - a) + b;   // same association as the reference: (Kdiag - sum A^2) + sum LTA^2
}

gp_status launch_cond_finish(gp_handle h, const void* d_items, int count, int N) {
  if (count <= 0 || N <= 0) return GP_OK;
  hipLaunchKernelGGL(cond_finish_kernel, dim3((N + 255) / 256, count), dim3(256), 0, h->stream,
                     (const CondFinish*)d_items, N);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
size_t cond_finish_item_bytes() { return sizeof(CondFinish); }
void cond_finish_fill(void* host_item, const double* s1, int rb1, const double* s2, int rb2, const double* dot,
                      int rbdot, DevKern k, double* fmean, double* fvar) {
  CondFinish* c = (CondFinish*)host_item;
  c->s1 = s1; c->s2 = s2; c->dot = dot; c->rb1 = rb1; c->rb2 = rb2; c->rbdot = rbdot; c->kern = k;
  c->fmean = fmean; c->fvar = fvar;
}

// ---------------------------------------------------------------------------------------------
// whitened KL per GP:  0.5*(|mu|^2 - M - sum log Lq_ii^2 + |tril(Lq)|_F^2)   (GPflow gauss_kl, K=None)
// and, when g_mu/g_L != NULL, ACCUMULATES  -dKL  into the ELBO gradient vector.
struct KlItem {
  const double* q_mu; const double* q_sqrt; int M;
  double* out;       // kl value as GP_KL_BLOCKS partial sums (their total is the KL)
  double* g_mu; double* g_sqrt;
};

// grid (items, GP_KL_BLOCKS): block b of an item takes a contiguous band of rows, so that the M^2-element sweep
// (and its read-modify-write of the gradient) is not serialised behind one workgroup's memory latency.
__global__ void __launch_bounds__(256) kl_white_kernel(const KlItem* __restrict__ items) {
  const KlItem it = items[blockIdx.x];
  const int M = it.M, nb = gridDim.y, b = blockIdx.y;
  const int rows_per = (M + nb - 1) / nb;
  const int r0 = b * rows_per, r1 = min(M, r0 + rows_per);
  double acc = 0.0;
  for (int i = r0 + (int)threadIdx.x; i < r1; i += 256) {
    const double mu = it.q_mu[i];
    const double d = it.q_sqrt[(int64_t)i * M + i];
    acc += mu * mu - log(d * d);
    if (it.g_mu) it.g_mu[i] -= mu;
  }
  for (int i = r0; i < r1; i++) {
    const double* row = it.q_sqrt + (int64_t)i * M;
    double* grow = it.g_sqrt ? it.g_sqrt + (int64_t)i * M : nullptr;
    for (int j = threadIdx.x; j <= i; j += 256) {
      const double l = row[j];
      acc = fma(l, l, acc);
      if (grow) grow[j] -= (i == j) ? (l - 1.0 / l) : l;
    }
  }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    it.out[b] = 0.5 * (s - (b == 0 ? (double)M : 0.0));
  }
}

gp_status launch_kl_white(gp_handle h, const void* d_items, int count) {
  if (count <= 0) return GP_OK;
  hipLaunchKernelGGL(kl_white_kernel, dim3(count, GP_KL_BLOCKS), dim3(256), 0, h->stream, (const KlItem*)d_items);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
size_t kl_item_bytes() { return sizeof(KlItem); }
void kl_item_fill(void* host_item, const double* q_mu, const double* q_sqrt, int M, double* out, double* g_mu,
                  double* g_sqrt) {
  KlItem* k = (KlItem*)host_item;
  k->q_mu = q_mu; k->q_sqrt = q_sqrt; k->M = M; k->out = out; k->g_mu = g_mu; k->g_sqrt = g_sqrt;
}

// unwhitened KL per GP (GPflow gauss_kl with K = Kuu + jitter I, pdgp.py:126-129), given L = chol(K), W = L^-1
// and the column sums of squares of W Lq (trace term) as row-block partials [nrb][M]:
//   0.5*(|W mu|^2 - M - sum log Lq_ii^2 + |W Lq|_F^2 + sum log L_ii^2)
struct KlUItem {
  const double* q_mu; const double* q_sqrt; const double* L; const double* W; const double* tr_part;
  int nrb; int M; double* out;
};

__global__ void __launch_bounds__(1024) kl_unwhite_kernel(const KlUItem* __restrict__ items) {
  const KlUItem it = items[blockIdx.x];
  const int M = it.M;
  double acc = 0.0;
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    double al = 0.0;
    for (int k = 0; k <= i; k++) al = fma(it.W[(int64_t)i * M + k], it.q_mu[k], al);
    const double dq = it.q_sqrt[(int64_t)i * M + i], dl = it.L[(int64_t)i * M + i];
    acc += al * al - log(dq * dq) + log(dl * dl);
    for (int r = 0; r < it.nrb; r++) acc += it.tr_part[(int64_t)r * M + i];
  }
  __shared__ double red[16];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w];
    it.out[0] = 0.5 * (s - (double)M);
  }
}

gp_status launch_kl_unwhite(gp_handle h, const void* d_items, int count) {
  if (count <= 0) return GP_OK;
  hipLaunchKernelGGL(kl_unwhite_kernel, dim3(count), dim3(1024), 0, h->stream, (const KlUItem*)d_items);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
size_t klu_item_bytes() { return sizeof(KlUItem); }
void klu_item_fill(void* host_item, const double* q_mu, const double* q_sqrt, const double* L, const double* W,
                   const double* tr_part, int nrb, int M, double* out) {
  KlUItem* k = (KlUItem*)host_item;
  k->q_mu = q_mu; k->q_sqrt = q_sqrt; k->L = L; k->W = W; k->tr_part = tr_part; k->nrb = nrb; k->M = M; k->out = out;
}

// elbo = sum(lik partials) - sum(kl);  grad_noise = sum(noise partials)
__global__ void __launch_bounds__(256) elbo_finish_kernel(const double* __restrict__ lik_partials, int nblocks,
                                                          const double* __restrict__ kl, int nkl,
                                                          double* __restrict__ elbo, double* __restrict__ g_noise) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int c = threadIdx.x; c < nblocks; c += 256) { a += lik_partials[2 * c]; b += lik_partials[2 * c + 1]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double k = 0.0;
    for (int g = 0; g < nkl; g++) k += kl[g];
    elbo[0] = red[0][0] - k;
    elbo[1] = k;   // sum of the KL terms (Pdgp.build_prior_kl)
    if (g_noise) g_noise[0] = red[1][0];
  }
}

gp_status launch_elbo_finish(gp_handle h, const double* lik_partials, int nblocks, const double* kl, int nkl,
                             double* elbo, double* g_noise) {
  hipLaunchKernelGGL(elbo_finish_kernel, dim3(1), dim3(256), 0, h->stream, lik_partials, nblocks, kl, nkl, elbo,
                     g_noise);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// mean_source[i][n] = nlin(mean_act_i[n]) * mean_com_i[n]   (pdgp.py:207)
__global__ void __launch_bounds__(256) mean_source_kernel(const double* __restrict__ fmean, int P, int n, int nlin,
                                                          double* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= n) return;
  double s, ds;
  nlin_eval(nlin, fmean[(int64_t)i * n + j], s, ds);
  out[(int64_t)i * n + j] = s * fmean[(int64_t)(i + P) * n + j];
}

gp_status launch_mean_source(gp_handle h, const double* fmean, int P, int n, int nlin, double* out) {
  if (n <= 0 || P <= 0) return GP_OK;
  hipLaunchKernelGGL(mean_source_kernel, dim3((n + 255) / 256, P), dim3(256), 0, h->stream, fmean, P, n, nlin, out);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
