// bwd.hip — hand-written reverse pass of the whitened Pdgp ELBO (what TF autodiff supplies to
// gpflow Model.optimize in the reference, gpitch/pdgp.py:133-170 via demo-modgp.py:44-45).
//
// With A = W Kuf (W = L^-1), B = Lq^T A, fmean = A^T mu, fvar = kdiag - colsum(A^2) + colsum(B^2) and
// upstream gm = dELBO/dfmean, gv = dELBO/dfvar (N-vectors per latent GP), D = diag(2 gv):
//   H     = A D A^T                      (split-K product over the frames, symmetric)
//   u     = A gm                          -> d/d mu
//   d/dLq = tril(H Lq)
//   E     = Lq Lq^T - I
//   Wbar  = tril(E H L^T + mu (L u)^T)    (uses Kuf = L A, so no second pass over the frames)
//   Kuf_bar = R (A D) + alpha gm^T,  R = W^T E,  alpha = W^T mu      (one dense strip GEMM)
//   Lbar  = -tril(W^T Wbar W^T);  P = Phi(L^T Lbar);  S = W^T P W;  Kuu_bar = (S + S^T)/2
// and the hyper-parameter / inducing-input gradients are contractions of Kuf_bar and Kuu_bar with the
// analytic kernel derivatives, evaluated on the fly.
#include "pdgp_plan.h"
#include <string.h>
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// small batched vector / matrix kernels (problem fields reused; see each kernel)

// o0[i] = sum_n A[i][n] * v0[n];  if o1: o1[i] += that.   grid (M, batch)
__global__ void __launch_bounds__(256) rowdot_kernel(const GemmProblem* __restrict__ probs) {
  const GemmProblem p = probs[blockIdx.y];
  const int i = blockIdx.x;
  if (i >= p.M) return;
  double acc = 0.0;
  if (p.a_f32) {
    const float* row = reinterpret_cast<const float*>(p.A) + (int64_t)i * p.lda;
    for (int n = threadIdx.x; n < p.N; n += 256) acc = fma((double)row[n], p.v0[n], acc);
  } else {
    const double* row = p.A + (int64_t)i * p.lda;
    for (int n = threadIdx.x; n < p.N; n += 256) acc = fma(row[n], p.v0[n], acc);
  }
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = (red[0] + red[1]) + (red[2] + red[3]);
    p.o0[i] = s;
    if (p.o1) p.o1[i] += s;
  }
}

// C[i][i] -= 1
__global__ void __launch_bounds__(256) sub_identity_kernel(const GemmProblem* __restrict__ probs) {
  const GemmProblem p = probs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < p.M) p.C[(int64_t)i * p.ldc + i] -= 1.0;
}

// C = Phi(C): lower triangle kept, diagonal halved, strictly-upper zeroed
__global__ void __launch_bounds__(256) phi_kernel(const GemmProblem* __restrict__ probs) {
  const GemmProblem p = probs[blockIdx.y];
  const int64_t total = (int64_t)p.M * p.M;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / p.M), j = (int)(idx % p.M);
    double* c = p.C + (int64_t)i * p.ldc + j;
    if (j > i) *c = 0.0;
    else if (j == i) *c *= 0.5;
  }
}

// C[i][j] += v0[i] * v1[j] for j <= i
__global__ void __launch_bounds__(256) rank1_tril_kernel(const GemmProblem* __restrict__ probs) {
  const GemmProblem p = probs[blockIdx.y];
  const int64_t total = (int64_t)p.M * p.M;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / p.M), j = (int)(idx % p.M);
    if (j <= i) p.C[(int64_t)i * p.ldc + j] += p.v0[i] * p.v1[j];
  }
}

// o0 = op(A) v0, A is M x M (lda); trans: o0[j] = sum_i A[i][j] v0[i].   grid (M, batch), one block per output
__global__ void __launch_bounds__(64) matvec_kernel(const GemmProblem* __restrict__ probs, int trans) {
  const GemmProblem p = probs[blockIdx.y];
  const int r = blockIdx.x;
  if (r >= p.M) return;
  double acc = 0.0;
  if (!trans) for (int k = threadIdx.x; k < p.M; k += 64) acc = fma(p.A[(int64_t)r * p.lda + k], p.v0[k], acc);
  else        for (int k = threadIdx.x; k < p.M; k += 64) acc = fma(p.A[(int64_t)k * p.lda + r], p.v0[k], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if (threadIdx.x == 0) p.o0[r] = acc;
}

// out[g] = sum_n v[g * stride + n]   (one block per row; fixed reduction order)
__global__ void __launch_bounds__(256) batched_sum_kernel(const double* __restrict__ v, int64_t stride, int n,
                                                          double* __restrict__ out) {
  const double* row = v + (int64_t)blockIdx.x * stride;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += row[i];
  __shared__ double red[4];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

static int ew_grid(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 512 ? 512 : (b < 1 ? 1 : b)); }

gp_status launch_rowdot_batched(gp_handle h, const GemmProblem* d, int batch, int maxM) {
  hipLaunchKernelGGL(rowdot_kernel, dim3(maxM, batch), dim3(256), 0, h->stream, d);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
gp_status launch_sub_identity_batched(gp_handle h, const GemmProblem* d, int batch, int maxM) {
  hipLaunchKernelGGL(sub_identity_kernel, dim3((maxM + 255) / 256, batch), dim3(256), 0, h->stream, d);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
gp_status launch_phi_batched(gp_handle h, const GemmProblem* d, int batch, int maxM) {
  hipLaunchKernelGGL(phi_kernel, dim3(ew_grid((int64_t)maxM * maxM), batch), dim3(256), 0, h->stream, d);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
gp_status launch_rank1_tril_batched(gp_handle h, const GemmProblem* d, int batch, int maxM) {
  hipLaunchKernelGGL(rank1_tril_kernel, dim3(ew_grid((int64_t)maxM * maxM), batch), dim3(256), 0, h->stream, d);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
gp_status launch_matvec_batched(gp_handle h, const GemmProblem* d, int batch, int maxM, int trans) {
  hipLaunchKernelGGL(matvec_kernel, dim3(maxM, batch), dim3(64), 0, h->stream, d, trans);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// ---------------------------------------------------------------------------------------------
// hyper-parameter contraction:  sums[s] = sum_{i,j} Gw[i][j] * dK[i][j]/dtheta_s
//   Gw[i][j] = G[i*ldg + j] (+ alpha[i]*gm[j])            (Kuf side, symmetric = 0)
//   Gw[i][j] = (G[i][j] + G[j][i]) / 2                     (Kuu side, symmetric = 1, x2 == x1)
// layout of the sums: [d_variance, d_lengthscales, d_energy_0.., d_frequency_0..]  (2 + 2m)
// gz_part[colblock][i] = sum_{j in block} Gw[i][j] * dK[i][j]/dx1_i
#define HY_THREADS 256
#define HY_ROWS 32
// element `idx` of a strip that is float64 or (g32, a float32 plan) float32
__device__ __forceinline__ double hy_ld(const double* __restrict__ p, int64_t idx, int g32) {
  return g32 ? (double)reinterpret_cast<const float*>(p)[idx] : p[idx];
}
typedef const double __attribute__((address_space(1))) * hy_gcptr;
typedef const float __attribute__((address_space(1))) * hy_gcfptr;
__device__ __forceinline__ double hy_ldg(hy_gcptr p, int64_t idx, int g32) {
  return g32 ? (double)((hy_gcfptr)p)[idx] : p[idx];
}

int hyper_num_sums(int m) { return 2 + 2 * m; }
// rows of x1 per workgroup: HY_ROWS for the strips; small contractions (window-sized problems, the Kuu side) get
// more, smaller workgroups (a 64 x 2001 contraction is otherwise 16 workgroups on a 256-CU device)
#define HY_ROWS_SMALL 8
static int hy_rows_for(int n1, int n2) { return ((int64_t)n1 * n2 >= (1 << 20)) ? HY_ROWS : HY_ROWS_SMALL; }
// partial records one contraction of an M x N problem can leave (any N' <= N: the plan is sized for its largest batch)
size_t hyper_kuf_records(int N, int M) {
  const size_t generic = ((size_t)(N + HY_THREADS - 1) / HY_THREADS + 1) * ((size_t)(M + HY_ROWS - 1) / HY_ROWS + 1);
  const size_t mfma = (size_t)(M + 63) / 64 * 32 + 1;  // hyper_sm_rows_kernel: groups of four row tiles x <= 32 column segments
  const size_t fused = ((size_t)(M + 63) / 64) * ((size_t)(N + 63) / 64);   // the Kuf_bar product's epilogue: one per 64 x 64 tile
  // small form: only while M * N' < 2^20
  const int64_t nsmall = (M > 0) ? (((int64_t)1 << 20) + M - 1) / M : 0;
  const int64_t ncap = nsmall < N ? nsmall : N;
  const size_t small = ((size_t)(ncap + HY_THREADS - 1) / HY_THREADS + 1) * ((size_t)(M + HY_ROWS_SMALL - 1) / HY_ROWS_SMALL + 1);
  size_t r = generic > mfma ? generic : mfma;
  if (fused > r) r = fused;
  return r > small ? r : small;
}

// MPAD = spectral-mixture partial count padded to a multiple of 4 (feature tables are zero-padded, so the
// inner loop carries no guards); SM = Mercer Matern-1/2 SM kernel; GZ = also produce the inducing-input gradient.
// KT >= 0: the (stationary) kernel type as a compile-time constant, so the row loop carries no type switch.
template <int MPAD, bool SM, bool GZ, int KT = -1>
__global__ void __launch_bounds__(HY_THREADS) hyper_contract_kernel(DevKern k, const double* __restrict__ x1, int n1,
                                                                    const double* __restrict__ x2, int n2,
                                                                    const double* __restrict__ G, int64_t ldg,
                                                                    const double* __restrict__ alpha,
                                                                    const double* __restrict__ gm, int symmetric,
                                                                    const double* __restrict__ f1,
                                                                    const double* __restrict__ f2,
                                                                    double* __restrict__ partials,
                                                                    double* __restrict__ gz_part, int wg_rows, int g32,
                                                                    const HyperItem* __restrict__ items) {
  if (items) {     // one launch for many contractions (window-batched SGPR plans): blockIdx.z = item
    const HyperItem it = items[blockIdx.z];
    k = it.k; x1 = it.x1; n1 = it.n1; G = it.G; ldg = it.ldg; alpha = it.alpha; gm = it.gm;
    if (it.x2) { x2 = it.x2; n2 = it.n2; }      // null: the launch's shared x2 / n2 (the frames of the batch)
    symmetric = it.symmetric; f1 = it.f1; f2 = it.f2; partials = it.partials; gz_part = it.gz; g32 = it.g32;
  }
  extern __shared__ double smem[];  // [HY_ROWS][2*MPAD] row features | omega[MPAD] | reduction scratch
  const double* th = k.theta;
  const double var = th[0], ls = th[1];
  const int m = k.m;
  const int j = blockIdx.x * HY_THREADS + threadIdx.x;
  const int i0 = blockIdx.y * wg_rows;             // wg_rows <= HY_ROWS rows of x1 per workgroup
  const int iend = min(i0 + wg_rows, n1);
  double* fzs = smem;
  __shared__ double etab[GP_EXP_TAB];  // 2^(j/64) for gp_exp_neg
  gp_exp_tab_init(etab);
  __shared__ double row_a[HY_ROWS];   // x1[i] / lengthscale, once per row
  if (threadIdx.x < HY_ROWS) row_a[threadIdx.x] = (i0 + (int)threadIdx.x < n1) ? x1[i0 + threadIdx.x] / ls : 0.0;
  double* omega = smem + (SM ? HY_ROWS * 2 * MPAD : 0);
  double* red = omega + (SM ? MPAD : 0);  // [4 waves][max(2+2m, HY_ROWS)]
  if (SM) {
    for (int t = threadIdx.x; t < HY_ROWS * 2 * MPAD; t += HY_THREADS) {
      int q = t / HY_ROWS, ii = t % HY_ROWS;
      fzs[ii * 2 * MPAD + q] = (ii < wg_rows && i0 + ii < n1) ? f1[(size_t)q * n1 + i0 + ii] : 0.0;
    }
    if ((int)threadIdx.x < MPAD) omega[threadIdx.x] = ((int)threadIdx.x < m) ? 6.283185307179586 * th[2 + m + threadIdx.x] : 0.0;
  }
  __syncthreads();
  const bool live = (j < n2);
  const int jc = live ? j : n2 - 1;
  const double xb = x2[jc];
  const double b = xb / ls, bb = __dmul_rn(b, b);
  const double gmj = (gm && live) ? gm[jc] : 0.0;
  double fxc[MPAD], fxs[MPAD];  // column features (static indices only)
  double acc_e[MPAD], acc_f[MPAD];
#pragma unroll
  for (int q = 0; q < MPAD; q++) {
    fxc[q] = SM ? f2[(size_t)q * n2 + jc] : 0.0;
    fxs[q] = SM ? f2[(size_t)(q + MPAD) * n2 + jc] : 0.0;
    acc_e[q] = 0.0; acc_f[q] = 0.0;
  }
  double acc_v = 0.0, acc_l = 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double inv_ls = 1.0 / ls, inv_ls2 = inv_ls * inv_ls;   // hoisted: f64 division is ~25 instructions

  for (int i = i0; i < iend; i++) {
    const double xa = x1[i];
    double w = 0.0;
    if (live) {
      w = hy_ld(G, (int64_t)i * ldg + j, g32);
      if (symmetric) w = 0.5 * (w + hy_ld(G, (int64_t)j * ldg + i, g32));
      if (alpha) w = fma(alpha[i], gmj, w);
    }
    const double a = row_a[i - i0], aa = __dmul_rn(a, a);
    const double r2 = __dadd_rn(__dadd_rn(-2.0 * __dmul_rn(a, b), aa), bb);
    const double d = xa - xb;
    double dz = 0.0;  // w * dK/dx1
    const int ktype = (KT >= 0) ? KT : k.type;
    if (!SM && ktype == GP_KERN_RBF) {
      const double e = gp_exp_neg(-0.5 * r2, etab);
      acc_v = fma(w, e, acc_v);
      acc_l = fma(w, var * e * r2 * inv_ls, acc_l);
      if (GZ) dz = -w * var * e * d * inv_ls2;
    } else {
      double r, rinv;   // sqrt and reciprocal sqrt from one rsq: no float64 division in the row loop
      gp_sqrt_rsqrt_pos(__dadd_rn(r2, 1e-12), r, rinv);
      if (SM) {
        // envelope phi(r) and phi'(r): Matern-1/2 (MercerMatern12sm) or Matern-5/2 (Matern52 * MercerCosMix)
        double E, dE;
        if (k.type == GP_KERN_MERCER_MATERN12SM) { E = gp_exp_neg(-r, etab); dE = -E; }
        else {
          const double s5 = 2.23606797749979, e5 = gp_exp_neg(-s5 * r, etab);
          E = (1.0 + s5 * r + (5.0 / 3.0) * r * r) * e5; dE = -(5.0 / 3.0) * r * (1.0 + s5 * r) * e5;
        }
        const double wvE = w * var * E, wvD = w * var * dE;
        const double wd = wvE * d;
        const double* fz = &fzs[(i - i0) * 2 * MPAD];
        double S = 0.0, Ssin = 0.0;
#pragma unroll
        for (int q = 0; q < MPAD; q++) {
          const double zc = fz[q], zs = fz[q + MPAD];
          const double cc = fma(zc, fxc[q], zs * fxs[q]);   // e_q cos(w_q d)
          const double ss = fma(zs, fxc[q], -zc * fxs[q]);  // e_q sin(w_q d)
          S += cc;
          acc_e[q] = fma(wvE, cc, acc_e[q]);
          acc_f[q] = fma(wd, ss, acc_f[q]);
          if (GZ) Ssin = fma(ss, omega[q], Ssin);
        }
        acc_v = fma(w * E, S, acc_v);
        acc_l = fma(-wvD * S, r2 * rinv * inv_ls, acc_l);
        if (GZ) dz = wvD * S * d * inv_ls2 * rinv - wvE * Ssin;
      } else {
        double phi, dphi;  // K = var * phi(r), dphi = phi'(r)
        if (ktype == GP_KERN_MATERN12) { phi = gp_exp_neg(-r, etab); dphi = -phi; }
        else if (ktype == GP_KERN_MATERN32) {
          const double s3 = 1.7320508075688772, e = gp_exp_neg(-s3 * r, etab);
          phi = (1.0 + s3 * r) * e; dphi = -3.0 * r * e;
        } else {
          const double s5 = 2.23606797749979, e = gp_exp_neg(-s5 * r, etab);
          phi = (1.0 + s5 * r + (5.0 / 3.0) * r * r) * e; dphi = -(5.0 / 3.0) * r * (1.0 + s5 * r) * e;
        }
        acc_v = fma(w, phi, acc_v);
        acc_l = fma(w * var * dphi, -r2 * rinv * inv_ls, acc_l);
        if (GZ) dz = w * var * dphi * d * inv_ls2 * rinv;
      }
    }
    if (GZ) {
      for (int o = 32; o > 0; o >>= 1) dz += __shfl_down(dz, o, 64);
      if (lane == 0) red[wave * HY_ROWS + (i - i0)] = dz;
    }
  }
  if (GZ) {
    __syncthreads();
    if ((int)threadIdx.x < wg_rows && i0 + (int)threadIdx.x < n1) {
      const int t = threadIdx.x;
      double s = (red[0 * HY_ROWS + t] + red[1 * HY_ROWS + t]) + (red[2 * HY_ROWS + t] + red[3 * HY_ROWS + t]);
      gz_part[(int64_t)blockIdx.x * n1 + i0 + t] = symmetric ? 2.0 * s : s;
    }
    __syncthreads();
  }
  // block-reduce the (2 + 2m) sums; the energy sums carry a factor e_q (cc = e_q cos), the frequency sums
  // a factor -1/(2 pi) relative to d/df_q
  const int ns = 2 + 2 * m;
  auto wred = [&](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; };
  double rv = wred(acc_v), rl = wred(acc_l);
  if (lane == 0) { red[wave * ns + 0] = rv; red[wave * ns + 1] = rl; }
  if (SM) {
#pragma unroll
    for (int q = 0; q < MPAD; q++) {
      double re = wred(acc_e[q]), rf = wred(acc_f[q]);
      if (lane == 0 && q < m) {
        red[wave * ns + 2 + q] = re / th[2 + q];
        red[wave * ns + 2 + m + q] = -6.283185307179586 * rf;
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < ns) {
    const int t = threadIdx.x;
    const double s = (red[0 * ns + t] + red[1 * ns + t]) + (red[2 * ns + t] + red[3 * ns + t]);
    partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ns + t] = s;
  }
}


// Kuf-side contraction of ONE weight matrix G with the derivatives of P MercerMatern12sm kernels of a sum (SGPRSS,
// sgpr_ss.py:42-43: every kernel of the Add sees the same Kuf_bar) in one pass: G, alpha and gm are read once instead of P
// times (at N = 65536, M = 512, P = 5 the per-kernel launches read the 268-MB strip five times: 0.72 ms of a 4.8-ms
// evaluation).  Per kernel the arithmetic and the partial-record layout are hyper_contract_kernel<4, true, false>'s.
struct HySumArgs { int P; int pad_; DevKern k[8]; const double* f1[8]; const double* f2[8]; double* partials[8]; };
template <int P>
__global__ void __launch_bounds__(HY_THREADS) hyper_contract_sum_kernel(HySumArgs a, const double* __restrict__ x1, int n1,
                                                                        const double* __restrict__ x2, int n2,
                                                                        const double* __restrict__ G, int64_t ldg,
                                                                        const double* __restrict__ alpha,
                                                                        const double* __restrict__ gm, int wg_rows, int g32) {
  constexpr int MPAD = 4;
  __shared__ double fzs[P][HY_ROWS * 2 * MPAD];
  __shared__ double row_a[P][HY_ROWS];
  __shared__ double red[4 * (2 + 2 * MPAD)];
  __shared__ double etab[GP_EXP_TAB];
  gp_exp_tab_init(etab);
  const int j = blockIdx.x * HY_THREADS + threadIdx.x;
  const int i0 = blockIdx.y * wg_rows;
  const int iend = min(i0 + wg_rows, n1);
  double var[P], inv_ls[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    const double* th = a.k[p].theta;
    var[p] = th[0]; inv_ls[p] = 1.0 / th[1];
    if (threadIdx.x < HY_ROWS) row_a[p][threadIdx.x] = (i0 + (int)threadIdx.x < n1) ? x1[i0 + threadIdx.x] / th[1] : 0.0;
    for (int t = threadIdx.x; t < HY_ROWS * 2 * MPAD; t += HY_THREADS) {
      const int q = t / HY_ROWS, ii = t % HY_ROWS;
      fzs[p][ii * 2 * MPAD + q] = (ii < wg_rows && i0 + ii < n1) ? a.f1[p][(size_t)q * n1 + i0 + ii] : 0.0;
    }
  }
  __syncthreads();
  const bool live = (j < n2);
  const int jc = live ? j : n2 - 1;
  const double xb = x2[jc];
  const double gmj = (gm && live) ? gm[jc] : 0.0;
  double b[P], bb[P], fxc[P][MPAD], fxs[P][MPAD], acc_v[P], acc_l[P], acc_e[P][MPAD], acc_f[P][MPAD];
#pragma unroll
  for (int p = 0; p < P; p++) {
    b[p] = xb / a.k[p].theta[1]; bb[p] = __dmul_rn(b[p], b[p]);
    acc_v[p] = 0.0; acc_l[p] = 0.0;
#pragma unroll
    for (int q = 0; q < MPAD; q++) {
      fxc[p][q] = a.f2[p][(size_t)q * n2 + jc];
      fxs[p][q] = a.f2[p][(size_t)(q + MPAD) * n2 + jc];
      acc_e[p][q] = 0.0; acc_f[p][q] = 0.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = i0; i < iend; i++) {
    double w = 0.0;
    if (live) {
      w = hy_ld(G, (int64_t)i * ldg + j, g32);
      if (alpha) w = fma(alpha[i], gmj, w);
    }
    const double d = x1[i] - xb;
#pragma unroll
    for (int p = 0; p < P; p++) {
      const double av = row_a[p][i - i0], aa = __dmul_rn(av, av);
      const double r2 = __dadd_rn(__dadd_rn(-2.0 * __dmul_rn(av, b[p]), aa), bb[p]);
      double r, rinv;
      gp_sqrt_rsqrt_pos(__dadd_rn(r2, 1e-12), r, rinv);
      const double E = gp_exp_neg(-r, etab);
      const double wvE = w * var[p] * E, wvD = -wvE;
      const double wd = wvE * d;
      const double* fz = &fzs[p][(i - i0) * 2 * MPAD];
      double S = 0.0;
#pragma unroll
      for (int q = 0; q < MPAD; q++) {
        const double zc = fz[q], zs = fz[q + MPAD];
        const double cc = fma(zc, fxc[p][q], zs * fxs[p][q]);   // e_q cos(w_q d)
        const double ss = fma(zs, fxc[p][q], -zc * fxs[p][q]);  // e_q sin(w_q d)
        S += cc;
        acc_e[p][q] = fma(wvE, cc, acc_e[p][q]);
        acc_f[p][q] = fma(wd, ss, acc_f[p][q]);
      }
      acc_v[p] = fma(w * E, S, acc_v[p]);
      acc_l[p] = fma(-wvD * S, r2 * rinv * inv_ls[p], acc_l[p]);
    }
  }
  auto wred = [&](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; };
#pragma unroll
  for (int p = 0; p < P; p++) {
    const double* th = a.k[p].theta;
    const int m = a.k[p].m, ns = 2 + 2 * m;
    const double rv = wred(acc_v[p]), rl = wred(acc_l[p]);
    __syncthreads();      // red is reused kernel after kernel
    if (lane == 0) { red[wave * ns + 0] = rv; red[wave * ns + 1] = rl; }
#pragma unroll
    for (int q = 0; q < MPAD; q++) {
      const double re = wred(acc_e[p][q]), rf = wred(acc_f[p][q]);
      if (lane == 0 && q < m) {
        red[wave * ns + 2 + q] = re / th[2 + q];
        red[wave * ns + 2 + m + q] = -6.283185307179586 * rf;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < ns) {
      const int t = threadIdx.x;
      const double sum = (red[0 * ns + t] + red[1 * ns + t]) + (red[2 * ns + t] + red[3 * ns + t]);
      a.partials[p][((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ns + t] = sum;
    }
  }
}

// true = taken.  P MercerMatern12sm kernels with at most four partials each, Kuf side (no inducing-input gradient);
// partials[p] = kernel p's partial records (hyper_kuf_records(n2, n1) x (2 + 2 m_p) doubles); *nparts = records left each.
bool launch_hyper_contract_sum(gp_handle h, const DevKern* kernels, double* const* feats, double* const* partials, int P,
                               const double* x1, int n1, const double* x2, int n2, const double* G, int64_t ldg,
                               const double* alpha, const double* gm, int g32, int* nparts, gp_status* st) {
  const bool enabled = gp_switches().hyper_sum != 0;
  if (!enabled || P < 2 || P > 6 || n1 <= 0 || n2 <= 0) return false;
  for (int p = 0; p < P; p++)
    if (kernels[p].type != GP_KERN_MERCER_MATERN12SM || kernels[p].m < 1 || kernels[p].m > 4 || !feats[p] || !partials[p]) return false;
  GpTimerScope ts(h, GP_TIMER_HYPER);
  HySumArgs a;
  a.P = P; a.pad_ = 0;
  const size_t f2off = gp_align_up((size_t)2 * 4 * n1, 32);
  for (int p = 0; p < 8; p++) {
    const int q = p < P ? p : 0;
    a.k[p] = kernels[q]; a.f1[p] = feats[q]; a.f2[p] = (x2 == x1) ? feats[q] : feats[q] + f2off; a.partials[p] = partials[q];
  }
  const int wg_rows = hy_rows_for(n1, n2);
  dim3 grid((n2 + HY_THREADS - 1) / HY_THREADS, (n1 + wg_rows - 1) / wg_rows);
#define HYS(P_) hipLaunchKernelGGL((hyper_contract_sum_kernel<P_>), grid, dim3(HY_THREADS), 0, h->stream, a, x1, n1, x2, n2, G, ldg, \
                                   alpha, gm, wg_rows, g32)
  switch (P) {
    case 2: HYS(2); break;
    case 3: HYS(3); break;
    case 4: HYS(4); break;
    case 5: HYS(5); break;
    default: HYS(6); break;
  }
#undef HYS
  hipError_t e = hipGetLastError();
  *st = (e == hipSuccess) ? GP_OK : gp_fail(h, GP_ERR_HIP, hipGetErrorString(e));
  if (nparts) *nparts = grid.x * grid.y;
  return true;
}

// Kuf side of a Mercer spectral-mixture kernel when the covariance values K themselves are still in memory (the
// forward pass's Kuf strip) and no inducing-input gradient is wanted.  Two identities remove the per-entry cosine
// sums:   sum_q e_q cos(w_q d) = K / (var phi(r))   (the variance / lengthscale terms become sums of w K ...), and
//         sum_j c_ij (zc_iq fxc_jq + zs_iq fxs_jq) = zc_iq sum_j c_ij fxc_jq + zs_iq sum_j c_ij fxs_jq
// (the row features leave the frame loop).  With c_ij = w_ij var phi(r_ij) and d_ij = c_ij (z_i - x_j) the per-partial
// sums are two small GEMMs over the frames,
//     TE[i][f] = sum_j c_ij F[f][j] ,   TD[i][f] = sum_j d_ij F[f][j]        (F = column features, 2m wide)
// followed by a dot with the row features Zf[i][f].  The float64 MFMA has the same peak as the float64 VALU, so this
// is not about flops: it takes the 4 x 2m multiply-adds per entry off the vector pipe.
// Row-streaming layout: a WAVEFRONT owns 16 rows (lane = row, as the MFMA A operand wants it) and streams along the
// frames, so
//  - the row-side quantities (z_i, z_i / l, alpha_i) are per-lane constants: no LDS, no barrier in the loop;
//  - G and Kuf are read as 64-byte pieces per row (16 bytes per lane, 4 lanes per row);
//  - A = c or d is computed by the very lane that must supply it, B = the column features;
//  - the feature index is interleaved (f even: cos of partial f / 2, odd: its sin), so the "partner" feature the
//    frequency gradient needs (sin <-> cos) is the neighbouring lane: one DPP swap on the final tile.
// The 16 columns of a block are dealt to the four k-lanes of a row in 16-byte pairs (columns 8 s + 2 kq + u): any
// consistent assignment of columns to MFMA k-slots gives the same sums, and this one makes every load a 64-byte row
// piece.  One partial record per (row tile, column segment).
// (Round 1's form gave a workgroup 64 columns and walked the rows in 32-row chunks restaged through LDS: 2.42 vs 2.34 ms
// for all contractions of a step run alone, the same step time — the step is bound by the float64 units)
#define HYR_COLSEG_MIN 512
#define HYR_MAX_SEGS 32
#define HYR_CF_MAX 4096        // columns of a segment whose envelope column factors fit the LDS table
#define HYR_SEP 1.0            // separable envelope: at least this many lengthscales between an entry's row and column
__device__ __forceinline__ double hyr_swap1(double v) {      // value of the neighbouring lane (lane ^ 1)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);    // quad_perm [1, 0, 3, 2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int NT, bool M52, bool G32>      // feature tiles (16 wide), Matern-5/2 envelope (else Matern-1/2), float32 strips
__global__ void __launch_bounds__(256, 2) hyper_sm_rows_kernel(DevKern k, const double* __restrict__ x1, int n1,
                                                               const double* __restrict__ x2, int n2,
                                                               const double* __restrict__ G, int64_t ldg,
                                                               const double* __restrict__ alpha,
                                                               const double* __restrict__ gm,
                                                               const double* __restrict__ Kuf, int64_t ldk,
                                                               const double* __restrict__ f1,
                                                               const double* __restrict__ f2,
                                                               double* __restrict__ partials, int g32, int col_seg,
                                                               const HyperItem* __restrict__ items) {
  if (items) {     // one launch for a whole kernel family: blockIdx.y = item (latent GP)
    const HyperItem it = items[blockIdx.y];
    k = it.k; x1 = it.x1; n1 = it.n1; G = it.G; ldg = it.ldg; alpha = it.alpha; gm = it.gm;
    if (it.x2) { x2 = it.x2; n2 = it.n2; }
    Kuf = it.kvals; ldk = it.ldk; f1 = it.f1; f2 = it.f2; partials = it.partials;
  }
  typedef double d4 __attribute__((ext_vector_type(4)));
  typedef double d2v __attribute__((ext_vector_type(2)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  typedef const d2v __attribute__((address_space(1))) * pd2v;
  typedef const f2v __attribute__((address_space(1))) * pf2v;
  const int mpad = ((k.m + 3) / 4) * 4;        // = sm_mpad(m): the feature tables hold 2 mpad rows (cos rows, sin rows)
  const int NF = 2 * mpad;                     // features: phi = 2 q (cos of partial q), 2 q + 1 (sin); NF <= 16 NT
  const hy_gcptr gG = (hy_gcptr)G, gK = (hy_gcptr)Kuf, gx1 = (hy_gcptr)x1, gx2 = (hy_gcptr)x2, galpha = (hy_gcptr)alpha,
                 ggm = (hy_gcptr)gm, gf1 = (hy_gcptr)f1, gf2 = (hy_gcptr)f2, th = (hy_gcptr)k.theta;
  __shared__ double etab[GP_EXP_TAB];
  __shared__ double red[4][2 + 2 * NT * 16];
  gp_exp_tab_init(etab);
  __syncthreads();
  const double var = th[0], ls = th[1];
  const int m = k.m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  // a workgroup = four adjacent row tiles walking the SAME columns: the column features each wavefront streams are in
  // the L1 / L2 the other three have just filled (one tile per workgroup, columns in quarters: 1.4x the algorithmic HBM
  // fetch from feature re-reads alone)
  if (blockIdx.x * 64 >= n1) return;           // (uniform per workgroup: the grid is sized for the largest item)
  const int i0 = (blockIdx.x * 4 + wave) * 16;
  const bool tile_on = (i0 < n1);              // a whole wavefront past the last row: no columns, joins the final sum
  const int row = i0 + lc;
  const bool rowok = (row < n1);
  const int rowc = rowok ? row : n1 - 1;
  const double zi = gx1[rowc], a = zi / ls, aa = __dmul_rn(a, a), m2a = -2.0 * a, al = galpha[rowc];
  const double inv_ls = 1.0 / ls;
  // table row of feature phi = 16 t + lc (B operand side: this lane supplies feature lc of every tile)
  int frow[NT]; bool fok[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int phi = 16 * t + lc, q = phi >> 1;
    fok[t] = (phi < NF);
    frow[t] = fok[t] ? ((phi & 1) ? mpad + q : q) : 0;
  }
  d4 TE[NT], TD[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { TE[t] = d4{0.0, 0.0, 0.0, 0.0}; TD[t] = d4{0.0, 0.0, 0.0, 0.0}; }
  double acc_v = 0.0, acc_l = 0.0;
  const int cw0 = blockIdx.z * col_seg, cw1 = tile_on ? min(n2, cw0 + col_seg) : cw0;
  const bool vec_ok = ((n2 & 1) == 0) && ((ldg & 1) == 0) && ((ldk & 1) == 0);
  // ---- separable envelope (cov.hip has the argument): with A / B the smallest / largest scaled row input a = z / l of
  // this workgroup's 64 rows, an entry whose column lies at least HYR_SEP below A or above B has
  //     exp(-s r) = exp(-s (a_i - A)) exp(-s (A - b_j))     or     exp(-s (B - a_i)) exp(-s (b_j - B)),   r = |a_i - b_j|
  // (relative deviation from the reference's sqrt((a - b)^2 + 1e-12): 1e-12 / (2 r) <= 5e-13).  The column factors of
  // the workgroup's segment are made once, into LDS (sign bit = "above the rows", NaN = inside the band: entry-by-entry
  // arithmetic there); the two row factors are per-lane constants.  A fast entry then costs a dozen float64 vector
  // instructions instead of ~45 (square root, reciprocal root and exp per entry).
  __shared__ __attribute__((aligned(16))) double cft[HYR_CF_MAX];
  __shared__ double wg_lo[4], wg_hi[4];
  constexpr double SENV = M52 ? 2.23606797749979 : 1.0;
  const int nseg_cols = min(n2, (int)(blockIdx.z * col_seg) + col_seg) - (int)(blockIdx.z * col_seg);
  const bool sep = (col_seg <= HYR_CF_MAX) && vec_ok;
  double rfp = 0.0, rfn = 0.0;
  if (sep) {
    const double ab = gx1[rowc] * inv_ls;           // the scaled input as the entry arithmetic below forms it (x * (1 / l))
    double lo = ab, hi = ab;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
    if (lane == 0) { wg_lo[wave] = tile_on ? lo : 1e300; wg_hi[wave] = tile_on ? hi : -1e300; }
    __syncthreads();
    const double A = fmin(fmin(wg_lo[0], wg_lo[1]), fmin(wg_lo[2], wg_lo[3]));
    const double B = fmax(fmax(wg_hi[0], wg_hi[1]), fmax(wg_hi[2], wg_hi[3]));
    for (int j = tid; j < nseg_cols; j += 256) {
      const double b = gx2[blockIdx.z * col_seg + j] * inv_ls;
      double v = __builtin_nan("");
      if (A - b >= HYR_SEP) v = gp_exp_neg(-SENV * (A - b), etab);
      else if (b - B >= HYR_SEP) v = -gp_exp_neg(-SENV * (b - B), etab);
      cft[j] = v;
    }
    rfp = gp_exp_neg(-SENV * (ab - A), etab);
    rfn = gp_exp_neg(-SENV * (B - ab), etab);
    __syncthreads();
  }
  // Whole 32-column stretches run from two register buffers in turn, the loads of block b + 1 issued before the
  // arithmetic of block b (two wavefronts per SIMD do not hide an HBM round trip per 16 columns by themselves), with
  // 16-byte loads and no column predicates; what is left (ragged ends, odd strides) goes block by block, checked.
  struct Blk { double g[4], k[4], x[4], gm[4], bf[NT][4]; };
  const int64_t rowg = (int64_t)rowc * ldg, rowk = (int64_t)rowc * ldk;
  unsigned foff[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) foff[t] = (unsigned)frow[t] * (unsigned)n2;
  auto load_fast = [&](int jb, Blk& B) {
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      const int c = jb + 8 * s2 + 2 * kq;      // this lane's pair of columns in sub-block s2
      if (G32) {
        const f2v g = *(pf2v)((hy_gcfptr)gG + rowg + c), kk = *(pf2v)((hy_gcfptr)gK + rowk + c);
        B.g[2 * s2] = g.x; B.g[2 * s2 + 1] = g.y; B.k[2 * s2] = kk.x; B.k[2 * s2 + 1] = kk.y;
      } else {
        const d2v g = *(pd2v)(gG + rowg + c), kk = *(pd2v)(gK + rowk + c);
        B.g[2 * s2] = g.x; B.g[2 * s2 + 1] = g.y; B.k[2 * s2] = kk.x; B.k[2 * s2 + 1] = kk.y;
      }
      const d2v xx = *(pd2v)(gx2 + c), gg = *(pd2v)(ggm + c);
      B.x[2 * s2] = xx.x; B.x[2 * s2 + 1] = xx.y; B.gm[2 * s2] = gg.x; B.gm[2 * s2 + 1] = gg.y;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const d2v b = *(pd2v)(gf2 + (foff[t] + (unsigned)c));
        B.bf[t][2 * s2] = fok[t] ? b.x : 0.0; B.bf[t][2 * s2 + 1] = fok[t] ? b.y : 0.0;
      }
    }
  };
  auto load_checked = [&](int jb, Blk& B) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = jb + 8 * (e >> 1) + 2 * kq + (e & 1);
      const bool on = (c < cw1);
      const int cc = on ? c : cw0;
      B.g[e] = on ? hy_ldg(gG, rowg + cc, G32) : 0.0;
      B.k[e] = on ? hy_ldg(gK, rowk + cc, G32) : 0.0;
      B.x[e] = on ? gx2[cc] : 0.0; B.gm[e] = on ? ggm[cc] : 0.0;
#pragma unroll
      for (int t = 0; t < NT; t++) B.bf[t][e] = (on && fok[t]) ? gf2[foff[t] + (unsigned)cc] : 0.0;
    }
  };
  const double ab_i = zi * inv_ls;
  auto compute = [&](int jb, const Blk& B, bool checked) {
    if (sep && !checked) {
      // column factors of this lane's four entries; the block is fast when none is in the band and all lie on one side
      const int cl = jb - cw0 + 2 * kq;
      const d2v c0 = *reinterpret_cast<const d2v*>(cft + cl), c1 = *reinterpret_cast<const d2v*>(cft + cl + 8);
      const double cf[4] = {c0.x, c0.y, c1.x, c1.y};
      const bool band = (cf[0] != cf[0]) || (cf[1] != cf[1]) || (cf[2] != cf[2]) || (cf[3] != cf[3]);
      const bool neg = (__double2hiint(cf[0]) < 0);
      const bool mixed = ((__double2hiint(cf[1]) < 0) != neg) || ((__double2hiint(cf[2]) < 0) != neg) || ((__double2hiint(cf[3]) < 0) != neg);
      const unsigned long long bad = __ballot(band || mixed), nb = __ballot(neg);
      if (bad == 0ull && (nb == 0ull || nb == ~0ull)) {
        const double rf = neg ? rfn : rfp;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const double xb = B.x[e];
          const double w = rowok ? fma(al, B.gm[e], B.g[e]) : 0.0;
          const double r = fabs(ab_i - xb * inv_ls);
          double E = rf * fabs(cf[e]), nratio = 1.0;
          if (M52) {
            const double s5 = 2.23606797749979, poly = 1.0 + s5 * r + (5.0 / 3.0) * r * r;
            E *= poly;
            nratio = (5.0 / 3.0) * r * (1.0 + s5 * r) / poly;
          }
          const double wvE = w * var * E, wd = wvE * (zi - xb);
          const double wk = w * B.k[e];
          acc_v += wk;
          acc_l = fma(M52 ? wk * nratio : wk, r * inv_ls, acc_l);
#pragma unroll
          for (int t = 0; t < NT; t++) {
            TE[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wvE, B.bf[t][e], TE[t], 0, 0, 0);
            TD[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wd, B.bf[t][e], TD[t], 0, 0, 0);
          }
        }
        return;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = jb + 8 * (e >> 1) + 2 * kq + (e & 1);
      const bool on = rowok && (!checked || c < cw1);
      const double xb = B.x[e];
      const double w = on ? fma(al, B.gm[e], B.g[e]) : 0.0;
      const double kv = B.k[e];
      const double bsc = xb * inv_ls, bb = __dmul_rn(bsc, bsc);
      const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(m2a, bsc), aa), bb);
      double r, rinv;
      gp_sqrt_rsqrt_pos(__dadd_rn(r2, 1e-12), r, rinv);
      double E, nratio;   // phi(r) and -phi'(r)/phi(r)
      if (!M52) { E = gp_exp_neg(-r, etab); nratio = 1.0; }
      else {
        const double s5 = 2.23606797749979, poly = 1.0 + s5 * r + (5.0 / 3.0) * r * r;
        E = poly * gp_exp_neg(-s5 * r, etab);
        nratio = (5.0 / 3.0) * r * (1.0 + s5 * r) / poly;
      }
      const double wvE = w * var * E, wd = wvE * (zi - xb);
      const double wk = w * kv;
      acc_v += wk;
      acc_l = fma(wk * nratio, (r2 * inv_ls) * rinv, acc_l);
#pragma unroll
      for (int t = 0; t < NT; t++) {
        TE[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wvE, B.bf[t][e], TE[t], 0, 0, 0);
        TD[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wd, B.bf[t][e], TD[t], 0, 0, 0);
      }
    }
  };
  int jb = cw0;
  if (vec_ok && cw1 - cw0 >= 32) {
    const int cfast = cw0 + ((cw1 - cw0) / 32) * 32;
    Blk b0, b1;
    load_fast(cw0, b0);
    for (; jb < cfast; jb += 32) {
      load_fast(jb + 16, b1);
      compute(jb, b0, false);
      load_fast(min(jb + 32, cfast - 16), b0);      // (the last one re-reads a block and is dropped)
      compute(jb + 16, b1, false);
    }
  }
  for (; jb < cw1; jb += 16) {
    Blk b;
    load_checked(jb, b);
    compute(jb, b, true);
  }
  // ---- finish: dot with the row features, reduce over the wavefront's k-lanes, then over the four wavefronts -------
  // element r of a tile: row kq + 4 r, feature phi = 16 t + lc.  SE[phi] = sum_i Zf[i][phi] TE[i][phi];
  // SD[phi] = sum_i Zf[i][phi] TD[i][phi ^ 1]  (partner feature = neighbouring lane)
  const int ns = 2 + 2 * m;
  auto wred = [&](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; };
  for (int t = lane; t < 2 + 2 * NT * 16; t += 64) red[wave][t] = 0.0;
  const double rv = wred(acc_v / var), rl = wred(acc_l);
#pragma unroll
  for (int t = 0; t < NT; t++) {
    double se = 0.0, sd = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int ri = i0 + kq + 4 * r;
      const double zf = (ri < n1 && fok[t]) ? gf1[(size_t)frow[t] * n1 + ri] : 0.0;
      se = fma(zf, TE[t][r], se);
      sd = fma(zf, hyr_swap1(TD[t][r]), sd);
    }
    se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
    sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
    if (kq == 0) { red[wave][2 + 16 * t + lc] = se; red[wave][2 + NT * 16 + 16 * t + lc] = sd; }
  }
  if (lane == 0) { red[wave][0] = rv; red[wave][1] = rl; }
  __syncthreads();
  if (tid < ns) {
    double out;
    auto tot = [&](int idx) { return (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]); };
    if (tid < 2) out = tot(tid);
    else if (tid < 2 + m) {           // d / d e_q ~ (SE[cos_q] + SE[sin_q]) / e_q
      const int q = tid - 2;
      out = (tot(2 + 2 * q) + tot(2 + 2 * q + 1)) / th[2 + q];
    } else {                          // d / d f_q ~ -2 pi (SD[sin_q] - SD[cos_q])
      const int q = tid - 2 - m;
      out = -6.283185307179586 * (tot(2 + NT * 16 + 2 * q + 1) - tot(2 + NT * 16 + 2 * q));
    }
    partials[((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * ns + tid] = out;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lean form of hyper_sm_rows_kernel (round 3) for the common case: Matern-1/2 envelope, whole 16-row tiles, whole
// 16-column blocks, 16-byte aligned strips.  Same row-streaming layout, same sums; what changed is everything AROUND the
// 8 NT matrix instructions of a 16-column block, because on this chip every vector instruction beside a float64 MFMA costs
// matrix time and the old form issued ~120 of them per block:
//  - per-COLUMN quantities (scaled input b_j = x_j / l, envelope column factor, gm_j) come from LDS tables built once per
//    workgroup and segment, not from per-block global loads and per-entry arithmetic; a byte per block says whether the
//    block is separable to one side (0 / 1) or needs the entry-by-entry envelope (2);
//  - var is folded into the row factors, l into the final sums: zi - xb = l (a_i - b_j), so an entry is
//    fma, sub, 4 mul, add, fma — 8 vector instructions instead of 13 + selects;
//  - feature lanes past the last feature read a real feature row instead of a select to zero: their output columns are
//    never used;
//  - G, Kuf and the column features are requested THREE blocks ahead into three register buffers that keep their identity
//    (loop unrolled over them), with scalar-base addresses (one lane offset per array for the whole kernel).
#define HYL_CF_MAX 2048
typedef char __attribute__((address_space(1))) const* hy_gcbytes;
__device__ __forceinline__ hy_gcbytes hyl_uniform(hy_gcbytes p) {
  const uint64_t b = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (hy_gcbytes)(((uint64_t)hi << 32) | lo);
}
template <int NT, bool G32>
__global__ void __launch_bounds__(256, 2) hyper_sm_rows_lean_kernel(DevKern k, const double* __restrict__ x1, int n1,
                                                                    const double* __restrict__ x2, int n2,
                                                                    const double* __restrict__ G, int64_t ldg,
                                                                    const double* __restrict__ alpha,
                                                                    const double* __restrict__ gm,
                                                                    const double* __restrict__ Kuf, int64_t ldk,
                                                                    const double* __restrict__ f1,
                                                                    const double* __restrict__ f2,
                                                                    double* __restrict__ partials, int col_seg,
                                                                    const HyperItem* __restrict__ items) {
  if (items) {     // one launch for a whole kernel family: blockIdx.y = item (latent GP)
    const HyperItem it = items[blockIdx.y];
    k = it.k; x1 = it.x1; n1 = it.n1; G = it.G; ldg = it.ldg; alpha = it.alpha; gm = it.gm;
    if (it.x2) { x2 = it.x2; n2 = it.n2; }
    Kuf = it.kvals; ldk = it.ldk; f1 = it.f1; f2 = it.f2; partials = it.partials;
  }
  typedef double d4 __attribute__((ext_vector_type(4)));
  typedef double d2v __attribute__((ext_vector_type(2)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  typedef const d2v __attribute__((address_space(1))) * pd2v;
  typedef const f2v __attribute__((address_space(1))) * pf2v;
  const int mpad = ((k.m + 3) / 4) * 4;
  const int NF = 2 * mpad;
  const hy_gcptr gx1 = (hy_gcptr)x1, gx2 = (hy_gcptr)x2, galpha = (hy_gcptr)alpha, ggm = (hy_gcptr)gm, gf1 = (hy_gcptr)f1,
                 th = (hy_gcptr)k.theta;
  __shared__ double etab[GP_EXP_TAB];
  __shared__ double red[4][2 + 2 * NT * 16];
  __shared__ __attribute__((aligned(16))) double t_cf[HYL_CF_MAX], t_b[HYL_CF_MAX], t_gm[HYL_CF_MAX];
  __shared__ unsigned char t_side[HYL_CF_MAX], t_cls[HYL_CF_MAX / 16];
  __shared__ double wg_lo[4], wg_hi[4];
  gp_exp_tab_init(etab);
  const double var = th[0], ls = th[1];
  const int m = k.m;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kq = lane >> 4;
  if (blockIdx.x * 64 >= n1) return;           // (uniform per workgroup: the grid is sized for the largest item)
  const int i0 = (blockIdx.x * 4 + wave) * 16;
  const bool tile_on = (i0 < n1);              // a whole wavefront past the last row: no columns, joins the final sum
  const int rowc = tile_on ? i0 + lc : n1 - 1;
  const double inv_ls = 1.0 / ls;
  const double al = galpha[rowc];
  const double ab_i = gx1[rowc] * inv_ls;      // the scaled input as the covariance build forms it (x * (1 / l))
  const double a_ = gx1[rowc] / ls, aa = __dmul_rn(a_, a_), m2a = -2.0 * a_;    // (entry-by-entry form: cov.hip's arithmetic)
  int frow[NT]; bool fok[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int phi = 16 * t + lc, q = phi >> 1;
    fok[t] = (phi < NF);
    frow[t] = fok[t] ? ((phi & 1) ? mpad + q : q) : 0;
  }
  const int cw0 = blockIdx.z * col_seg, cw1 = tile_on ? min(n2, cw0 + col_seg) : cw0;
  const int nseg_cols = min(n2, cw0 + col_seg) - cw0;
  // ---- tables of the segment's columns ---------------------------------------------------------------------------------
  {
    double lo = ab_i, hi = ab_i;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
    if (lane == 0) { wg_lo[wave] = tile_on ? lo : 1e300; wg_hi[wave] = tile_on ? hi : -1e300; }
  }
  __syncthreads();
  const double A = fmin(fmin(wg_lo[0], wg_lo[1]), fmin(wg_lo[2], wg_lo[3]));
  const double B = fmax(fmax(wg_hi[0], wg_hi[1]), fmax(wg_hi[2], wg_hi[3]));
  for (int j = tid; j < nseg_cols; j += 256) {
    const double b = gx2[cw0 + j] * inv_ls;
    double v = 0.0;
    unsigned char side = 2;
    if (A - b >= HYR_SEP) { v = gp_exp_neg(-(A - b), etab); side = 0; }
    else if (b - B >= HYR_SEP) { v = gp_exp_neg(-(b - B), etab); side = 1; }
    t_cf[j] = v; t_b[j] = b; t_gm[j] = ggm[cw0 + j]; t_side[j] = side;
  }
  __syncthreads();
  for (int blk = tid; blk * 16 < nseg_cols; blk += 256) {
    unsigned char c = t_side[blk * 16];
#pragma unroll
    for (int u = 1; u < 16; u++) c = (t_side[blk * 16 + u] == c) ? c : (unsigned char)2;
    t_cls[blk] = c;
  }
  const double rfp = var * gp_exp_neg(-(ab_i - A), etab), rfn = var * gp_exp_neg(-(B - ab_i), etab);
  __syncthreads();
  // ---- the stream -------------------------------------------------------------------------------------------------------
  struct Blk { d2v g[2], kv[2], bf[NT][2]; };
  const hy_gcbytes bG = (hy_gcbytes)G, bK = (hy_gcbytes)Kuf, bF = (hy_gcbytes)f2;
  constexpr int ES = G32 ? 4 : 8;              // strip element size
  const uint32_t voG = (uint32_t)(((int64_t)rowc * ldg + 2 * kq) * ES), voK = (uint32_t)(((int64_t)rowc * ldk + 2 * kq) * ES);
  uint32_t voF[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) voF[t] = (uint32_t)(((int64_t)frow[t] * n2 + 2 * kq) * 8);
  const int jlast = cw1 - 16;
  auto request = [&](int jb, Blk& b) {
    const int jc = min(jb, jlast);             // (past the end: the last block again, never consumed)
    const hy_gcbytes sG = hyl_uniform(bG + (int64_t)jc * ES), sK = hyl_uniform(bK + (int64_t)jc * ES),
                     sF = hyl_uniform(bF + (int64_t)jc * 8);
    uint32_t vg = voG, vk = voK;
    asm volatile("" : "+v"(vg), "+v"(vk));
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      if (G32) {
        const f2v g = *(pf2v)(sG + vg + s2 * 8 * ES), kk = *(pf2v)(sK + vk + s2 * 8 * ES);
        b.g[s2] = d2v{(double)g.x, (double)g.y}; b.kv[s2] = d2v{(double)kk.x, (double)kk.y};
      } else {
        b.g[s2] = *(pd2v)(sG + vg + s2 * 64); b.kv[s2] = *(pd2v)(sK + vk + s2 * 64);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
      uint32_t vf = voF[t];
      asm volatile("" : "+v"(vf));
      b.bf[t][0] = *(pd2v)(sF + vf); b.bf[t][1] = *(pd2v)(sF + vf + 64);
    }
  };
  d4 TE[NT], TD[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { TE[t] = d4{0.0, 0.0, 0.0, 0.0}; TD[t] = d4{0.0, 0.0, 0.0, 0.0}; }
  double acc_v = 0.0, acc_l = 0.0;
  auto consume = [&](int jb, const Blk& b) {
    const int cl = jb - cw0 + 2 * kq;
    const int cls = __builtin_amdgcn_readfirstlane((int)t_cls[(jb - cw0) >> 4]);
    const d2v cf0 = *reinterpret_cast<const d2v*>(t_cf + cl), cf1 = *reinterpret_cast<const d2v*>(t_cf + cl + 8);
    const d2v bs0 = *reinterpret_cast<const d2v*>(t_b + cl), bs1 = *reinterpret_cast<const d2v*>(t_b + cl + 8);
    const d2v gm0 = *reinterpret_cast<const d2v*>(t_gm + cl), gm1 = *reinterpret_cast<const d2v*>(t_gm + cl + 8);
    const double cfv[4] = {cf0.x, cf0.y, cf1.x, cf1.y}, bsv[4] = {bs0.x, bs0.y, bs1.x, bs1.y},
                 gmv[4] = {gm0.x, gm0.y, gm1.x, gm1.y};
    const double gv[4] = {b.g[0].x, b.g[0].y, b.g[1].x, b.g[1].y}, kv[4] = {b.kv[0].x, b.kv[0].y, b.kv[1].x, b.kv[1].y};
    double wvE[4], wd[4];
    if (cls != 2) {
      const double rf = cls ? rfn : rfp;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const double w = fma(al, gmv[e], gv[e]);
        const double d = ab_i - bsv[e];
        const double wk = w * kv[e];
        wvE[e] = w * (rf * cfv[e]); wd[e] = wvE[e] * d;
        acc_v += wk;
        acc_l = fma(wk, fabs(d), acc_l);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const double w = fma(al, gmv[e], gv[e]);
        const double bsc = bsv[e], bb = __dmul_rn(bsc, bsc);
        const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(m2a, bsc), aa), bb);
        double r, rinv;
        gp_sqrt_rsqrt_pos(__dadd_rn(r2, 1e-12), r, rinv);
        const double E = gp_exp_neg(-r, etab);
        const double wk = w * kv[e];
        wvE[e] = w * var * E; wd[e] = wvE[e] * (ab_i - bsc);
        acc_v += wk;
        acc_l = fma(wk, r2 * rinv, acc_l);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const double bfe = (e & 1) ? b.bf[t][e >> 1].y : b.bf[t][e >> 1].x;
        TE[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wvE[e], bfe, TE[t], 0, 0, 0);
        TD[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wd[e], bfe, TD[t], 0, 0, 0);
      }
  };
  if (cw1 - cw0 >= 16) {
    Blk b0, b1, b2;
    request(cw0, b0); request(cw0 + 16, b1); request(cw0 + 32, b2);
    for (int jb = cw0; jb < cw1;) {
      consume(jb, b0); request(jb + 48, b0); jb += 16; if (jb >= cw1) break;
      consume(jb, b1); request(jb + 48, b1); jb += 16; if (jb >= cw1) break;
      consume(jb, b2); request(jb + 48, b2); jb += 16;
    }
  }
  // ---- finish (as hyper_sm_rows_kernel; TD and the lengthscale sum carry the factors folded out of the loop) ------------
  const int ns = 2 + 2 * m;
  auto wred = [&](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; };
  for (int t = lane; t < 2 + 2 * NT * 16; t += 64) red[wave][t] = 0.0;
  const double rv = wred(acc_v / var), rl = wred(acc_l * inv_ls);
#pragma unroll
  for (int t = 0; t < NT; t++) {
    double se = 0.0, sd = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int ri = i0 + kq + 4 * r;
      const double zf = (ri < n1 && fok[t]) ? gf1[(size_t)frow[t] * n1 + ri] : 0.0;
      se = fma(zf, TE[t][r], se);
      sd = fma(zf, hyr_swap1(TD[t][r]), sd);
    }
    sd *= ls;
    se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
    sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
    if (kq == 0) { red[wave][2 + 16 * t + lc] = se; red[wave][2 + NT * 16 + 16 * t + lc] = sd; }
  }
  if (lane == 0) { red[wave][0] = rv; red[wave][1] = rl; }
  __syncthreads();
  if (tid < ns) {
    double out;
    auto tot = [&](int idx) { return (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]); };
    if (tid < 2) out = tot(tid);
    else if (tid < 2 + m) {           // d / d e_q ~ (SE[cos_q] + SE[sin_q]) / e_q
      const int q = tid - 2;
      out = (tot(2 + 2 * q) + tot(2 + 2 * q + 1)) / th[2 + q];
    } else {                          // d / d f_q ~ -2 pi (SD[sin_q] - SD[cos_q])
      const int q = tid - 2 - m;
      out = -6.283185307179586 * (tot(2 + NT * 16 + 2 * q + 1) - tot(2 + NT * 16 + 2 * q));
    }
    partials[((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * ns + tid] = out;
  }
}

#define HYL_DISPATCH(L, nt, g32)                                                                                   \
  do {                                                                                                             \
    switch (nt) {                                                                                                  \
      case 1: if (g32) L(1, true); else L(1, false); break;                                                         \
      case 2: if (g32) L(2, true); else L(2, false); break;                                                         \
      default: if (g32) L(3, true); else L(3, false); break;                                                        \
    }                                                                                                              \
  } while (0)
// Which form contracts a Mercer family's Kuf_bar strip with dK/dtheta: 2 = hyper_sm_rows_lean_kernel (NT = ceil(2 mpad / 16)
// <= 3 feature tiles, i.e. up to 24 partials, MercerMatern12sm, whole 16-tiles), 1 = hyper_sm_rows_kernel (NT <= 2: up to 16
// partials; any envelope, ragged shapes), 0 = the generic vector-pipe kernel.  The row-streaming kernels are instantiated
// only where they compile without register spills: NT = 4 (25-32 partials) spilled 99-135 VGPRs in either form and NT = 3 of
// the older form 7-26, so those shapes take the generic kernel (profiles/r04/kernel_resource_usage.txt).
static int hy_rows_form(int nt, bool lean_ok) {
  if (lean_ok && nt <= 3) return 2;
  return nt <= 2 ? 1 : 0;
}
#define HYR_DISPATCH_ENV(L, NT_, m52, g32)                                             \
  do {                                                                                  \
    if (m52) { if (g32) L(NT_, true, true); else L(NT_, true, false); }                 \
    else { if (g32) L(NT_, false, true); else L(NT_, false, false); }                   \
  } while (0)
#define HYR_DISPATCH(L, nt, m52, g32)                                                   \
  do {                                                                                  \
    switch (nt) {                                                                       \
      case 1: HYR_DISPATCH_ENV(L, 1, m52, g32); break;                                  \
      default: HYR_DISPATCH_ENV(L, 2, m52, g32); break;                                 \
    }                                                                                   \
  } while (0)

// grid of the row-streaming contraction: row tiles x column segments (a multiple of 128 columns each)
static void hyr_geometry(int n1, int n2, int count, int* col_seg, int* nseg) {
  const int rows64 = (n1 + 63) / 64;
  int segs = 1;
  while (segs < HYR_MAX_SEGS && (int64_t)rows64 * count * segs < 1024 && (n2 + segs * 2 - 1) / (segs * 2) >= HYR_COLSEG_MIN) segs *= 2;
  int cs = ((n2 + segs - 1) / segs + 31) / 32 * 32;
  *col_seg = cs;
  *nseg = (n2 + cs - 1) / cs;
}

// Broadcast-form kernels Matern12sm (matern12_spectral_mixture.py:38-56) and Matern32sm (kernels.py:204-258):
//   d = x1_i - x2_j + 1e-12,  r = |d|,  K = var * phi(r; ls) * sum_q e_q cos(2 pi f_q r),
//   phi = exp(-r / ls)  or  (1 + r1) exp(-r1), r1 = sqrt(3) r / ls.
// Same grid and partial layout as hyper_contract_kernel; m cosines per entry, as the reference.
template <bool GZ>
__global__ void __launch_bounds__(HY_THREADS) hyper_m12sm_kernel(DevKern k, const double* __restrict__ x1, int n1,
                                                                 const double* __restrict__ x2, int n2,
                                                                 const double* __restrict__ G, int64_t ldg,
                                                                 const double* __restrict__ alpha,
                                                                 const double* __restrict__ gm, int symmetric,
                                                                 double* __restrict__ partials,
                                                                 double* __restrict__ gz_part, int wg_rows, int g32,
                                                                 const HyperItem* __restrict__ items) {
  if (items) {
    const HyperItem it = items[blockIdx.z];
    k = it.k; x1 = it.x1; n1 = it.n1; G = it.G; ldg = it.ldg; alpha = it.alpha; gm = it.gm;
    if (it.x2) { x2 = it.x2; n2 = it.n2; }
    symmetric = it.symmetric; partials = it.partials; gz_part = it.gz; g32 = it.g32;
  }
  extern __shared__ double smem[];   // e[m] | omega[m] | reduction scratch [4][max(2+2m, HY_ROWS)]
  const double* th = k.theta;
  const double var = th[0], ls = th[1];
  const int m = k.m;
  double* en = smem;
  double* omega = smem + m;
  double* red = smem + 2 * m;
  if ((int)threadIdx.x < m) {
    en[threadIdx.x] = th[2 + threadIdx.x];
    omega[threadIdx.x] = __dmul_rn(6.283185307179586, th[2 + m + threadIdx.x]);
  }
  __syncthreads();
  const int j = blockIdx.x * HY_THREADS + threadIdx.x;
  const int i0 = blockIdx.y * wg_rows;             // wg_rows <= HY_ROWS rows of x1 per workgroup
  const int iend = min(i0 + wg_rows, n1);
  const bool live = (j < n2);
  const int jc = live ? j : n2 - 1;
  const double xb = x2[jc];
  const double gmj = (gm && live) ? gm[jc] : 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ns = 2 + 2 * m;
  double acc_v = 0.0, acc_l = 0.0;
  double acc_e[32], acc_f[32];
#pragma unroll
  for (int q = 0; q < 32; q++) { acc_e[q] = 0.0; acc_f[q] = 0.0; }
  for (int i = i0; i < iend; i++) {
    double w = 0.0;
    if (live) {
      w = hy_ld(G, (int64_t)i * ldg + j, g32);
      if (symmetric) w = 0.5 * (w + hy_ld(G, (int64_t)j * ldg + i, g32));
      if (alpha) w = fma(alpha[i], gmj, w);
    }
    const double d = __dadd_rn(__dadd_rn(x1[i], -xb), 1e-12);
    const double r = __dsqrt_rn(__dmul_rn(d, d));
    // dr/dx1 = sign(d).  On the Kuu side z_i enters K_ij as x1 and K_ji as x2: the two contributions are
    // sign(d_ij) and -sign(d_ji); they cancel on the diagonal (d_ii = +1e-12 both ways), where the generic
    // "twice the x1 derivative" shortcut of the symmetric mode would be wrong for this kinked kernel.
    double sg = d < 0.0 ? -1.0 : 1.0;
    if (symmetric) {
      const double dji = __dadd_rn(__dadd_rn(xb, -x1[i]), 1e-12);
      sg = 0.5 * (sg - (dji < 0.0 ? -1.0 : 1.0));
    }
    // envelope phi(r; l), d phi / d r, d phi / d l
    double E, dEr, dEl;
    if (k.type == GP_KERN_MATERN12SM) {
      E = exp(-(r / ls)); dEr = -E / ls; dEl = E * r / (ls * ls);
    } else {   // Matern32sm: r1 = sqrt(3) r / l
      const double s3 = 1.7320508075688772, r1 = s3 * (r / ls), e1 = exp(-r1);
      E = (1.0 + r1) * e1; dEr = -r1 * e1 * s3 / ls; dEl = r1 * r1 * e1 / ls;
    }
    const double wvE = w * var * E;
    double S = 0.0, Sf = 0.0;
#pragma unroll
    for (int q = 0; q < 32; q++) {
      if (q < m) {
        double sn, cs;
        sincos(__dmul_rn(omega[q], r), &sn, &cs);
        S = fma(en[q], cs, S);
        Sf = fma(en[q] * omega[q], sn, Sf);
        acc_e[q] = fma(wvE, cs, acc_e[q]);
        acc_f[q] = fma(wvE * en[q] * r, sn, acc_f[q]);
      }
    }
    acc_v = fma(w * E, S, acc_v);
    acc_l = fma(w * var * dEl, S, acc_l);
    if (GZ) {
      double dz = sg * (w * var * dEr * S - wvE * Sf);
      for (int o = 32; o > 0; o >>= 1) dz += __shfl_down(dz, o, 64);
      if (lane == 0) red[wave * HY_ROWS + (i - i0)] = dz;
    }
  }
  if (GZ) {
    __syncthreads();
    if ((int)threadIdx.x < wg_rows && i0 + (int)threadIdx.x < n1) {
      const int t = threadIdx.x;
      double s = (red[0 * HY_ROWS + t] + red[1 * HY_ROWS + t]) + (red[2 * HY_ROWS + t] + red[3 * HY_ROWS + t]);
      gz_part[(int64_t)blockIdx.x * n1 + i0 + t] = symmetric ? 2.0 * s : s;
    }
    __syncthreads();
  }
  auto wred = [&](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; };
  double rv = wred(acc_v), rl = wred(acc_l);
  if (lane == 0) { red[wave * ns + 0] = rv; red[wave * ns + 1] = rl; }
#pragma unroll
  for (int q = 0; q < 32; q++) {
    if (q < m) {
      double re = wred(acc_e[q]), rf = wred(acc_f[q]);
      if (lane == 0) {
        red[wave * ns + 2 + q] = re;
        red[wave * ns + 2 + m + q] = -6.283185307179586 * rf;
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < ns) {
    const int t = threadIdx.x;
    const double s = (red[0 * ns + t] + red[1 * ns + t]) + (red[2 * ns + t] + red[3 * ns + t]);
    partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ns + t] = s;
  }
}

template <int MPAD, bool SM, int KT = -1>
static void launch_hyper_t(gp_handle h, dim3 grid, size_t sh, DevKern k, const double* x1, int n1, const double* x2,
                           int n2, const double* G, int64_t ldg, const double* alpha, const double* gm, int symmetric,
                           const double* f1, const double* f2, double* partials, double* gz, int wg_rows, int g32,
                           const HyperItem* items = nullptr) {
  if (gz)
    hipLaunchKernelGGL((hyper_contract_kernel<MPAD, SM, true, KT>), grid, dim3(HY_THREADS), sh, h->stream, k, x1, n1, x2,
                       n2, G, ldg, alpha, gm, symmetric, f1, f2, partials, gz, wg_rows, g32, items);
  else
    hipLaunchKernelGGL((hyper_contract_kernel<MPAD, SM, false, KT>), grid, dim3(HY_THREADS), sh, h->stream, k, x1, n1,
                       x2, n2, G, ldg, alpha, gm, symmetric, f1, f2, partials, gz, wg_rows, g32, items);
}

gp_status launch_hyper_contract(gp_handle h, DevKern k, const double* x1, int n1, const double* x2, int n2,
                                const double* G, int64_t ldg, const double* alpha, const double* gm, int symmetric,
                                const double* feat, double* partials, int* nparts, double* gz_partials,
                                const double* kvals, int64_t ldk, int g32) {
  GpTimerScope ts(h, GP_TIMER_HYPER);
  const int wg_rows = hy_rows_for(n1, n2);
  if (gp_kern_is_broadcast(k.type)) {
    dim3 grid((n2 + HY_THREADS - 1) / HY_THREADS, (n1 + wg_rows - 1) / wg_rows);
    const int ns = 2 + 2 * k.m;
    const int redw = ns > HY_ROWS ? ns : HY_ROWS;
    const size_t sh = (2 * (size_t)k.m + 4 * (size_t)redw) * sizeof(double);
    if (gz_partials)
      hipLaunchKernelGGL((hyper_m12sm_kernel<true>), grid, dim3(HY_THREADS), sh, h->stream, k, x1, n1, x2, n2, G, ldg,
                         alpha, gm, symmetric, partials, gz_partials, wg_rows, g32, (const HyperItem*)nullptr);
    else
      hipLaunchKernelGGL((hyper_m12sm_kernel<false>), grid, dim3(HY_THREADS), sh, h->stream, k, x1, n1, x2, n2, G, ldg,
                         alpha, gm, symmetric, partials, gz_partials, wg_rows, g32, (const HyperItem*)nullptr);
    GP_HIP_CHECK(h, hipGetLastError());
    if (nparts) *nparts = grid.x * grid.y;
    return GP_OK;
  }
  const bool sm = gp_kern_is_mercer(k.type);
  const int mp = sm ? sm_mpad(k.m) : 0;
  const double* f1 = feat;
  const double* f2 = (x2 == x1 || !feat) ? feat : feat + gp_align_up((size_t)2 * mp * n1, 32);
  dim3 grid((n2 + HY_THREADS - 1) / HY_THREADS, (n1 + wg_rows - 1) / wg_rows);
  const int ns = 2 + 2 * k.m;
  const int redw = ns > HY_ROWS ? ns : HY_ROWS;
  size_t sh = ((sm ? (size_t)HY_ROWS * 2 * mp + mp : 0) + 4 * (size_t)redw) * sizeof(double);
  int col_seg = 0, nseg = 0;
  if (sm) hyr_geometry(n1, n2, 1, &col_seg, &nseg);
  const bool lean_ok = sm && k.type == GP_KERN_MERCER_MATERN12SM && (n1 % 16) == 0 && (n2 % 16) == 0 && col_seg <= HYL_CF_MAX &&
                       (ldg % 2) == 0 && (ldk % 2) == 0 && ((uintptr_t)G % 16) == 0 && ((uintptr_t)kvals % 16) == 0 && ((uintptr_t)f2 % 16) == 0;
  const int rows_form = sm ? hy_rows_form((2 * mp + 15) / 16, lean_ok) : 0;
  if (sm && kvals && !gz_partials && !symmetric && alpha && gm && rows_form > 0) {
    // Kuf side with the covariance strip still in memory: the matrix-core form (hyper_sm_rows_kernel)
    dim3 gridm((n1 + 63) / 64, 1, nseg);
    if (rows_form == 2) {
#define HYL_ONE(NT_, G32_) hipLaunchKernelGGL((hyper_sm_rows_lean_kernel<NT_, G32_>), gridm, dim3(256), 0, h->stream, k, x1, n1, x2, \
                                              n2, G, ldg, alpha, gm, kvals, ldk, f1, f2, partials, col_seg, (const HyperItem*)nullptr)
      HYL_DISPATCH(HYL_ONE, (2 * mp + 15) / 16, g32 != 0);
#undef HYL_ONE
      GP_HIP_CHECK(h, hipGetLastError());
      if (nparts) *nparts = gridm.x * gridm.z;
      return GP_OK;
    }
#define HY_MFMA(NT_, M52_, G32_) hipLaunchKernelGGL((hyper_sm_rows_kernel<NT_, M52_, G32_>), gridm, dim3(256), 0, h->stream, k, x1, \
                                                  n1, x2, n2, G, ldg, alpha, gm, kvals, ldk, f1, f2, partials, g32, col_seg,    \
                                                  (const HyperItem*)nullptr)
    HYR_DISPATCH(HY_MFMA, (2 * mp + 15) / 16, k.type != GP_KERN_MERCER_MATERN12SM, g32 != 0);
#undef HY_MFMA
    GP_HIP_CHECK(h, hipGetLastError());
    if (nparts) *nparts = gridm.x * gridm.z;
    return GP_OK;
  }
#define HY_ARGS grid, sh, k, x1, n1, x2, n2, G, ldg, alpha, gm, symmetric, f1, f2, partials, gz_partials, wg_rows, g32
  if (!sm) switch (k.type) {
    case GP_KERN_MATERN12: launch_hyper_t<1, false, GP_KERN_MATERN12>(h, HY_ARGS); break;
    case GP_KERN_MATERN32: launch_hyper_t<1, false, GP_KERN_MATERN32>(h, HY_ARGS); break;
    case GP_KERN_MATERN52: launch_hyper_t<1, false, GP_KERN_MATERN52>(h, HY_ARGS); break;
    default: launch_hyper_t<1, false, GP_KERN_RBF>(h, HY_ARGS); break;
  }
  else switch (mp) {
    case 4: launch_hyper_t<4, true>(h, HY_ARGS); break;
    case 8: launch_hyper_t<8, true>(h, HY_ARGS); break;
    case 12: launch_hyper_t<12, true>(h, HY_ARGS); break;
    case 16: launch_hyper_t<16, true>(h, HY_ARGS); break;
    case 20: launch_hyper_t<20, true>(h, HY_ARGS); break;
    case 24: launch_hyper_t<24, true>(h, HY_ARGS); break;
    case 28: launch_hyper_t<28, true>(h, HY_ARGS); break;
    default: launch_hyper_t<32, true>(h, HY_ARGS); break;
  }
#undef HY_ARGS
  GP_HIP_CHECK(h, hipGetLastError());
  if (nparts) *nparts = grid.x * grid.y;
  return GP_OK;
}

// g_theta[s] += sum of partials (+ kdiag term); g_z[i] += sum over column blocks
__global__ void __launch_bounds__(256) hyper_finish_kernel(DevKern k, const double* __restrict__ partials, int nparts,
                                                           const double* __restrict__ gv_sum,
                                                           double* __restrict__ g_theta,
                                                           const double* __restrict__ gz_part, int ncolblocks, int n1,
                                                           double* __restrict__ g_z) {
  const int ns = 2 + 2 * k.m;
  const int s = blockIdx.x;
  if (s < ns) {
    __shared__ double red[256];
    double a = 0.0;
    for (int c = threadIdx.x; c < nparts; c += 256) a += partials[(int64_t)c * ns + s];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      double v = red[0];
      if (gv_sum) {  // d kdiag / d theta contribution: kdiag = var (stationary) or var * sum(e)
        const double gs = gv_sum[0];
        const bool smk = gp_kern_kdiag_energy(k.type);
        if (s == 0) {
          double se = 1.0;
          if (smk) { se = 0.0; for (int q = 0; q < k.m; q++) se += k.theta[2 + q]; }
          v += gs * se;
        } else if (smk && s >= 2 && s < 2 + k.m) {
          v += gs * k.theta[0];
        }
      }
      g_theta[s] += v;
    }
  } else if (g_z && gz_part) {
    // remaining blocks: z-gradient, 256 rows per block
    const int i = (blockIdx.x - ns) * 256 + threadIdx.x;
    if (i < n1) {
      double a = 0.0;
      for (int c = 0; c < ncolblocks; c++) a += gz_part[(int64_t)c * n1 + i];
      g_z[i] += a;
    }
  }
}

gp_status launch_hyper_finish(gp_handle h, DevKern k, const double* partials, int nparts, const double* gv_sum,
                              double* g_theta, const double* gz_partials, int ncolblocks, int n1, double* g_z) {
  const int ns = 2 + 2 * k.m;
  int blocks = ns + ((g_z && gz_partials) ? (n1 + 255) / 256 : 0);
  hipLaunchKernelGGL(hyper_finish_kernel, dim3(blocks), dim3(256), 0, h->stream, k, partials, nparts, gv_sum, g_theta,
                     gz_partials, ncolblocks, n1, g_z);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// One launch for all latent GPs: block (s, item) adds the Kuf-side and the Kuu-side partial sums (and the Kdiag term)
// of entry s of that GP's theta gradient; the remaining blocks do the same for its inducing-input gradient.
size_t hyper_finish_item_bytes() { return sizeof(HyperFinishItem); }

__global__ void __launch_bounds__(256) hyper_finish_items_kernel(const HyperFinishItem* __restrict__ items) {
  const HyperFinishItem it = items[blockIdx.y];
  const int ns = 2 + 2 * it.k.m;
  const int s = blockIdx.x;
  if (s < ns) {
    __shared__ double red[256];
    double a = 0.0;
    for (int c = threadIdx.x; c < it.np_uf; c += 256) a += it.p_uf[(int64_t)c * ns + s];
    for (int c = threadIdx.x; c < it.np_uu; c += 256) a += it.p_uu[(int64_t)c * ns + s];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      double v = red[0];
      const double gs = it.gv_sum[0];      // d kdiag / d theta contribution: kdiag = var (stationary) or var * sum(e)
      const bool smk = gp_kern_kdiag_energy(it.k.type);
      if (s == 0) {
        double se = 1.0;
        if (smk) { se = 0.0; for (int q = 0; q < it.k.m; q++) se += it.k.theta[2 + q]; }
        v += gs * se;
      } else if (smk && s >= 2 && s < 2 + it.k.m) {
        v += gs * it.k.theta[0];
      }
      it.g_theta[s] += v;
    }
  } else if (it.g_z) {
    const int i = (blockIdx.x - ns) * 256 + threadIdx.x;
    if (i < it.n1) {
      double a = 0.0;
      for (int c = 0; c < it.cb_uf; c++) a += it.gz_uf[(int64_t)c * it.n1 + i];
      for (int c = 0; c < it.cb_uu; c++) a += it.gz_uu[(int64_t)c * it.n1 + i];
      it.g_z[i] += a;
    }
  }
}

gp_status launch_hyper_finish_items(gp_handle h, const HyperFinishItem* d_items, int count, int maxblocks) {
  if (count <= 0) return GP_OK;
  hipLaunchKernelGGL(hyper_finish_items_kernel, dim3(maxblocks, (unsigned)count), dim3(256), 0, h->stream, d_items);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// Many contractions of one kernel family (same type and partial count, same n1 x n2) in one launch: the generic
// (vector-pipe) kernels with an item array.  *nparts = partial records each item leaves.
gp_status launch_hyper_contract_items(gp_handle h, int type, int m, const HyperItem* d_items, int count, int n1, int n2,
                                      int with_gz, int* nparts, int use_mfma, const double* x2_shared, int g32_items, int lean_items) {
  if (count <= 0) return GP_OK;
  GpTimerScope ts(h, GP_TIMER_HYPER);
  int col_seg = 0, nseg = 0;
  if (gp_kern_is_mercer(type)) hyr_geometry(n1, n2, count, &col_seg, &nseg);
  const bool lean_ok = lean_items && type == GP_KERN_MERCER_MATERN12SM && (n1 % 16) == 0 && (n2 % 16) == 0 && col_seg <= HYL_CF_MAX;
  const int rows_form = gp_kern_is_mercer(type) ? hy_rows_form((2 * sm_mpad(m) + 15) / 16, lean_ok) : 0;
  if (use_mfma && gp_kern_is_mercer(type) && !with_gz && rows_form > 0) {
    dim3 gridm((n1 + 63) / 64, count, nseg);
    DevKern k0{type, m, nullptr};
    if (rows_form == 2) {
#define HYL_ITEMS(NT_, G32_) hipLaunchKernelGGL((hyper_sm_rows_lean_kernel<NT_, G32_>), gridm, dim3(256), 0, h->stream, k0,        \
                                                (const double*)nullptr, 0, x2_shared, n2, (const double*)nullptr, (int64_t)0,         \
                                                (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (int64_t)0,   \
                                                (const double*)nullptr, (const double*)nullptr, (double*)nullptr, col_seg, d_items)
      HYL_DISPATCH(HYL_ITEMS, (2 * sm_mpad(m) + 15) / 16, g32_items != 0);
#undef HYL_ITEMS
      GP_HIP_CHECK(h, hipGetLastError());
      if (nparts) *nparts = gridm.x * gridm.z;
      return GP_OK;
    }
#define HYI_MFMA(NT_, M52_, G32_) hipLaunchKernelGGL((hyper_sm_rows_kernel<NT_, M52_, G32_>), gridm, dim3(256), 0, h->stream, k0, \
                                                   (const double*)nullptr, 0, x2_shared, n2, (const double*)nullptr, (int64_t)0, \
                                                   (const double*)nullptr, (const double*)nullptr, (const double*)nullptr,       \
                                                   (int64_t)0, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, \
                                                   0, col_seg, d_items)
    HYR_DISPATCH(HYI_MFMA, (2 * sm_mpad(m) + 15) / 16, type != GP_KERN_MERCER_MATERN12SM, g32_items != 0);
#undef HYI_MFMA
    GP_HIP_CHECK(h, hipGetLastError());
    if (nparts) *nparts = gridm.x * gridm.z;
    return GP_OK;
  }
  const int wg_rows = hy_rows_for(n1, n2);
  dim3 grid((n2 + HY_THREADS - 1) / HY_THREADS, (n1 + wg_rows - 1) / wg_rows, count);
  const int ns = 2 + 2 * m;
  const int redw = ns > HY_ROWS ? ns : HY_ROWS;
  DevKern k0{type, m, nullptr};
  if (gp_kern_is_broadcast(type)) {
    const size_t sh = (2 * (size_t)m + 4 * (size_t)redw) * sizeof(double);
    if (with_gz)
      hipLaunchKernelGGL((hyper_m12sm_kernel<true>), grid, dim3(HY_THREADS), sh, h->stream, k0, (const double*)nullptr, 0,
                         x2_shared, n2, (const double*)nullptr, (int64_t)0, (const double*)nullptr,
                         (const double*)nullptr, 0, (double*)nullptr, (double*)nullptr, wg_rows, 0, d_items);
    else
      hipLaunchKernelGGL((hyper_m12sm_kernel<false>), grid, dim3(HY_THREADS), sh, h->stream, k0, (const double*)nullptr, 0,
                         x2_shared, n2, (const double*)nullptr, (int64_t)0, (const double*)nullptr,
                         (const double*)nullptr, 0, (double*)nullptr, (double*)nullptr, wg_rows, 0, d_items);
  } else {
    const bool sm = gp_kern_is_mercer(type);
    const int mp = sm ? sm_mpad(m) : 0;
    const size_t sh = ((sm ? (size_t)HY_ROWS * 2 * mp + mp : 0) + 4 * (size_t)redw) * sizeof(double);
    double* gzflag = with_gz ? (double*)(uintptr_t)8 : nullptr;     // only its null-ness selects the kernel variant
#define HYI_ARGS grid, sh, k0, (const double*)nullptr, 0, x2_shared, n2, (const double*)nullptr, (int64_t)0, \
                 (const double*)nullptr, (const double*)nullptr, 0, (const double*)nullptr, (const double*)nullptr,      \
                 (double*)nullptr, gzflag, wg_rows, 0, d_items
    if (!sm) switch (type) {
      case GP_KERN_MATERN12: launch_hyper_t<1, false, GP_KERN_MATERN12>(h, HYI_ARGS); break;
      case GP_KERN_MATERN32: launch_hyper_t<1, false, GP_KERN_MATERN32>(h, HYI_ARGS); break;
      case GP_KERN_MATERN52: launch_hyper_t<1, false, GP_KERN_MATERN52>(h, HYI_ARGS); break;
      default: launch_hyper_t<1, false, GP_KERN_RBF>(h, HYI_ARGS); break;
    }
    else switch (mp) {
      case 4: launch_hyper_t<4, true>(h, HYI_ARGS); break;
      case 8: launch_hyper_t<8, true>(h, HYI_ARGS); break;
      case 12: launch_hyper_t<12, true>(h, HYI_ARGS); break;
      case 16: launch_hyper_t<16, true>(h, HYI_ARGS); break;
      case 20: launch_hyper_t<20, true>(h, HYI_ARGS); break;
      case 24: launch_hyper_t<24, true>(h, HYI_ARGS); break;
      case 28: launch_hyper_t<28, true>(h, HYI_ARGS); break;
      default: launch_hyper_t<32, true>(h, HYI_ARGS); break;
    }
#undef HYI_ARGS
  }
  GP_HIP_CHECK(h, hipGetLastError());
  if (nparts) *nparts = grid.x * grid.y;
  return GP_OK;
}

// ---------------------------------------------------------------------------------------------
// orchestration
// slots below S_E are batched over all latent GPs, slots from S_E on over the GPs whose kernel gradients are needed.
// S_QW_* / S_GQ_* / S_WB_*: unwhitened model only (see pdgp_backward).
enum BwdSlot { S_H = 0, S_U, S_HLQ, S_QW_MU, S_QW_L, S_GQ_MU, S_GQ_L,
               S_E, S_EH, S_WBAR, S_LU, S_RANK1, S_R, S_ALPHA, S_G, S_T2, S_LBAR, S_P, S_T3, S_S, S_WB_R1, S_WB_L,
               S_COUNT };
static_assert(S_COUNT <= 24, "gp_pdgp_plan_s::off_bwd");


gp_status pdgp_upload_bwd(gp_pdgp_plan p, const double* params, const double* x, int n, double* grad) {
  (void)x;
  const int G = p->G;
  const size_t slot_bytes = gp_align_up(G * sizeof(GemmProblem), 256);
  size_t base = pdgp_kl_region_bytes(G);
  for (int s = 0; s < S_COUNT; s++) p->off_bwd[s] = base + s * slot_bytes;
  p->off_kl2 = base + 24 * slot_bytes;
  p->off_fin_items = p->off_kl2 + pdgp_kl_region_bytes(G);
  p->h_fin_items.clear();       // the descriptor block is rewritten: force a fresh upload of the finish items
  const bool white = p->whiten != 0;
  size_t slab_off = 0;
  p->kgps.clear();
  for (int g = 0; g < G; g++)
    if (p->gps[g].need_theta || p->gps[g].need_z) p->kgps.push_back(g);
  p->nK = (int)p->kgps.size();
  int kslot = 0;
  for (int g = 0; g < G; g++) {
    const PdgpGP& q = p->gps[g];
    const CondTask& t = p->cb.tasks[g];
    const BwdBufs& b = p->bw[g];
    const int M = q.M;
    const int64_t ldN = gp_strip_ld(n, q.f32 != 0);     // this GP's strips: float64 or float32 (per-GP precision)
    // unwhitened model: the chain runs on the equivalent whitened state q' = (W q_mu, W Lq) and its gradient
    // buffers; pdgp_backward maps the result back (see there)
    const double* q_mu = white ? params + q.off_qmu : b.qmu_w;
    const double* q_sqrt = white ? params + q.off_qsqrt : b.Lq_w;
    double* g_mu = white ? grad + q.off_qmu : b.g_qmu_w;
    double* g_sqrt = white ? grad + q.off_qsqrt : b.g_Lq_w;
    const double* gm = p->gFmu + (size_t)g * n;
    const double* gv = p->gFvar + (size_t)g * n;
    const bool kneed = (q.need_theta || q.need_z);
    auto P = [&](int slot) -> GemmProblem& {
      // the kernel-gradient chain (slots S_R..S_S) is batched over the GPs that need it only
      const bool kchain = (slot >= S_E);   // E, Wbar, R, alpha, Kuf_bar and the Cholesky-adjoint chain
      if (kchain && !kneed) { memset(&p->dummy_prob, 0, sizeof(p->dummy_prob)); return p->dummy_prob; }
      const int idx = kchain ? kslot : g;
      GemmProblem& r = *(GemmProblem*)(p->h_misc.data() + p->off_bwd[slot] + idx * sizeof(GemmProblem));
      memset(&r, 0, sizeof(r));
      r.M = M; r.N = M; r.K = M; r.lda = M; r.ldb = M; r.ldc = M;
      return r;
    };
    { GemmProblem& r = P(S_H); r.A = t.A; r.lda = ldN; r.B = t.A; r.ldb = ldN; r.K = n; r.v1 = gv; r.C = b.H;
      r.o2 = p->slabs + slab_off; slab_off += gp_align_up((size_t)p->nsplit * M * M * sizeof(double), 256) / sizeof(double);
      // fused u = A gm: partials per K-slice in o1, result in o0 and accumulated into grad q_mu (xa)
      r.v2 = gm; r.o1 = b.upart; r.o0 = b.u; r.xa = g_mu; }
    { GemmProblem& r = P(S_U); r.A = t.A; r.lda = ldN; r.N = n; r.v0 = gm; r.o0 = b.u; r.o1 = g_mu; r.a_f32 = q.f32; }
    { GemmProblem& r = P(S_HLQ); r.A = b.H; r.B = q_sqrt; r.C = g_sqrt; }
    if (!white) {
      const double* qm = params + q.off_qmu;
      const double* qs = params + q.off_qsqrt;
      { GemmProblem& r = P(S_QW_MU); r.A = t.W; r.v0 = qm; r.o0 = b.qmu_w; }
      { GemmProblem& r = P(S_QW_L); r.A = t.W; r.B = qs; r.C = b.Lq_w; }
      { GemmProblem& r = P(S_GQ_MU); r.A = t.W; r.v0 = b.g_qmu_w; r.o0 = grad + q.off_qmu; }
      { GemmProblem& r = P(S_GQ_L); r.A = t.W; r.B = b.g_Lq_w; r.C = grad + q.off_qsqrt; }
      { GemmProblem& r = P(S_WB_R1); r.C = b.Wbar; r.v0 = b.g_qmu_w; r.v1 = qm; }
      { GemmProblem& r = P(S_WB_L); r.A = b.g_Lq_w; r.B = qs; r.C = b.Wbar; }
      kl_item_fill(p->h_misc.data() + p->off_kl2 + g * kl_item_bytes(), b.qmu_w, b.Lq_w, M, p->kl_dummy + (size_t)g * GP_KL_BLOCKS, b.g_qmu_w,
                   b.g_Lq_w);
    }
    { GemmProblem& r = P(S_E); r.A = q_sqrt; r.B = q_sqrt; r.C = b.E; }
    { GemmProblem& r = P(S_EH); r.A = b.E; r.B = b.H; r.C = b.T1; }
    { GemmProblem& r = P(S_WBAR); r.A = b.T1; r.B = t.L; r.C = b.Wbar; }
    { GemmProblem& r = P(S_LU); r.A = t.L; r.v0 = b.u; r.o0 = b.Lu; }
    { GemmProblem& r = P(S_RANK1); r.C = b.Wbar; r.v0 = q_mu; r.v1 = b.Lu; }
    { GemmProblem& r = P(S_R); r.A = t.W; r.B = b.E; r.C = b.R; }
    { GemmProblem& r = P(S_ALPHA); r.A = t.W; r.v0 = q_mu; r.o0 = b.alpha; }
    { GemmProblem& r = P(S_G); r.A = b.R; r.B = t.A; r.ldb = ldN; r.N = n; r.v1 = gv; r.C = b.G; r.ldc = ldN; r.xb = b.R32;
      // (read only by the form that contracts Kuf_bar with dK/dtheta in its epilogue — gemm_strip.hip role 5)
      r.kern = t.kern; r.xa = params + q.off_z; r.v0 = b.alpha; r.v2 = gm; r.o0 = b.hyp_part; }
    { GemmProblem& r = P(S_T2); r.A = t.W; r.B = b.Wbar; r.C = b.T2; }
    { GemmProblem& r = P(S_LBAR); r.A = b.T2; r.B = t.W; r.C = b.T1; }
    { GemmProblem& r = P(S_P); r.A = t.L; r.B = b.T1; r.C = b.T2; }
    { GemmProblem& r = P(S_T3); r.A = t.W; r.B = b.T2; r.C = b.H; }
    { GemmProblem& r = P(S_S); r.A = b.H; r.B = t.W; r.C = b.E; }
    if (kneed) kslot++;
  }
  // Kuf-side contractions, one launch per kernel family (same type and partial count): item array in kgps order inside
  // each family.  x2 stays null in the items: the frames of the batch come with the launch (their pointer may change
  // from step to step without a descriptor upload).
  p->off_hy_items = p->off_fin_items + gp_align_up((size_t)G * hyper_finish_item_bytes(), 256);
  p->hy_fams.clear();
  for (size_t s = 0; s < p->kgps.size(); s++) {
    const int g = p->kgps[s];
    const PdgpGP& q = p->gps[g];
    const int key_m = gp_kern_has_partials(q.ktype) ? q.m : 0;
    int fi = -1;
    for (size_t f = 0; f < p->hy_fams.size(); f++)
      if (p->hy_fams[f].type == q.ktype && p->hy_fams[f].m == key_m && p->hy_fams[f].f32 == q.f32) fi = (int)f;
    if (fi < 0) { gp_pdgp_plan_s::HyFamily nf; nf.type = q.ktype; nf.m = key_m; nf.M = q.M; nf.f32 = q.f32; nf.batched = true; p->hy_fams.push_back(nf); fi = (int)p->hy_fams.size() - 1; }
    gp_pdgp_plan_s::HyFamily& fam = p->hy_fams[fi];
    fam.gps.push_back(g);
    if (q.M != fam.M || q.need_z || !q.need_theta) fam.batched = false;   // the per-GP path handles those
  }
  {
    HyperItem* items = (HyperItem*)(p->h_misc.data() + p->off_hy_items);
    int pos = 0;
    for (auto& fam : p->hy_fams) {
      fam.first = pos; fam.count = (int)fam.gps.size();
      fam.mfma = (gp_kern_is_mercer(fam.type) && fam.batched) ? 1 : 0;
      for (int g : fam.gps) {
        const PdgpGP& q = p->gps[g];
        const CondTask& t = p->cb.tasks[g];
        const BwdBufs& bb = p->bw[g];
        HyperItem& it = items[pos++];
        memset(&it, 0, sizeof(it));
        const int64_t ldN = gp_strip_ld(n, q.f32 != 0);
        it.k = t.kern; it.x1 = params + q.off_z; it.n1 = q.M; it.x2 = nullptr; it.n2 = n; it.G = bb.G; it.ldg = ldN;
        it.alpha = bb.alpha; it.gm = p->gFmu + (size_t)g * n; it.symmetric = 0; it.partials = bb.hyp_part; it.gz = nullptr;
        it.kvals = t.Kuf; it.ldk = ldN; it.g32 = q.f32;
        if (gp_kern_is_mercer(q.ktype) && t.feat) {
          it.f1 = t.feat;
          it.f2 = t.feat + gp_align_up((size_t)2 * sm_mpad(q.m) * q.M, 32);
        }
        // its Kuu-side twin (contraction of Kuu_bar = E with dK(z, z)): G entries further on
        HyperItem& iu = items[G + pos - 1];
        memset(&iu, 0, sizeof(iu));
        iu.k = t.kern; iu.x1 = params + q.off_z; iu.n1 = q.M; iu.x2 = iu.x1; iu.n2 = q.M; iu.G = bb.E; iu.ldg = q.M;
        iu.symmetric = 1; iu.partials = bb.hyp_part_uu;
        if (gp_kern_is_mercer(q.ktype) && t.feat) { iu.f1 = t.feat; iu.f2 = t.feat; }
      }
    }
  }
  return GP_OK;
}

// R = W^T (Lq Lq^T - I) and alpha = W^T q_mu depend on the parameters and on W only: when the helper stream exists they
// are enqueued on it during the FORWARD pass, right behind the Kuu factorisation it has just run (no wait on the main
// stream, which is busy with the forward GEMM strips), and the backward pass finds them ready.
// The whitened KL terms (parameters only) ride along: on the main stream they were one of five tiny kernels between
// the last forward strip product and the first backward one, with the device idle around them.
gp_status pdgp_prefetch_backward(gp_pdgp_plan p, int n, bool* kl_done) {
  gp_handle h = p->h;
  p->era_ready = false;
  if (kl_done) *kl_done = false;
  if (!(p->whiten && p->nK > 0 && n >= 4096 && p->overlap >= 2 && h->aux_stream && !h->aux_active)) return GP_OK;
  if (!h->ev_era && hipEventCreateWithFlags(&h->ev_era, hipEventDisableTiming) != hipSuccess) { h->ev_era = nullptr; return GP_OK; }
  auto D = [&](int slot) { return (const GemmProblem*)(p->d_misc + p->off_bwd[slot]); };
  const int nK = p->nK, maxM = p->maxM;
  hipStream_t mainq = h->stream;
  h->stream = h->aux_stream;
  gp_status st = GP_OK;
  GemmFlags f;
  f = GemmFlags(); f.triA = TRI_LOWER; f.transB = 1; f.triB = TRI_UPPER;
  st = launch_gemm_batched(h, D(S_E), nK, maxM, maxM, f);
  if (st == GP_OK) st = launch_sub_identity_batched(h, D(S_E), nK, maxM);
  f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER;
  if (st == GP_OK) st = launch_gemm_batched(h, D(S_R), nK, maxM, maxM, f);
  if (st == GP_OK) st = launch_matvec_batched(h, D(S_ALPHA), nK, maxM, 1);
  if (st == GP_OK && kl_done) {
    st = launch_kl_white(h, p->d_misc + p->off_kl_items, p->G);
    *kl_done = (st == GP_OK);
  }
  hipError_t e = hipEventRecord(h->ev_era, h->aux_stream);
  h->stream = mainq;
  GP_CHECK(st);
  if (e != hipSuccess) return gp_fail(h, GP_ERR_HIP, "hipEventRecord on the helper stream failed");
  p->era_ready = true;
  return GP_OK;
}

gp_status pdgp_backward(gp_pdgp_plan p, const double* params, const double* x, int n, double* grad) {
  gp_handle h = p->h;
  const int G = p->G, maxM = p->maxM;
  const int n64 = p->n64;     // latent GPs [0, n64): float64 strips, [n64, G): float32 strips
  auto D = [&](int slot) { return (const GemmProblem*)(p->d_misc + p->off_bwd[slot]); };
  GemmFlags f;
  const bool white = p->whiten != 0;
  if (!white) {
    // conditional(whiten=False) + gauss_kl(q_mu, q_sqrt, K) (pdgp.py:123-129, 147-155) is the whitened model at
    //   q_mu' = W q_mu,  Lq' = W Lq      (W = chol(Kuu + jitter I)^-1),
    // so the whitened chain below runs on (q_mu', Lq') with gradient buffers (g', G'), and afterwards
    //   grad q_mu = W^T g',  grad q_sqrt = tril(W^T G'),  Wbar += tril(g' q_mu^T + G' Lq^T).
    GP_HIP_CHECK(h, hipMemsetAsync(p->qw_block, 0, p->qw_doubles * sizeof(double), h->stream));
    GP_CHECK(launch_matvec_batched(h, D(S_QW_MU), G, maxM, 0));
    f = GemmFlags(); f.triA = TRI_LOWER; f.triB = TRI_LOWER; f.triC = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, D(S_QW_L), G, maxM, maxM, f));
    GP_CHECK(launch_kl_white(h, p->d_misc + p->off_kl2, G));   // accumulates -dKL/dq' into (g', G')
  }
  const int nK = p->nK;   // latent GPs whose kernel hyper-parameters / inducing inputs are trainable
  // sum_n gv  (kdiag term)
  // sum_n gv and the ELBO's final reduction (pdgp.hip: pdgp_finish) are wanted only when the step ends: they go to the end
  // of the helper stream's chain when there is one, and run here otherwise
  bool sums_done = false;
  auto late_sums = [&]() -> gp_status {
    if (sums_done) return GP_OK;
    sums_done = true;
    hipLaunchKernelGGL(batched_sum_kernel, dim3(G), dim3(256), 0, h->stream, p->gFvar, (int64_t)n, n, p->bw[0].gvsum);
    GP_HIP_CHECK(h, hipGetLastError());
    if (p->fin.pending) {
      p->fin.pending = false;
      GP_CHECK(launch_elbo_finish(h, p->fin.lik_partials, p->fin.nb, p->fin.kl, p->fin.nkl, p->fin.elbo, p->fin.g_noise));
    }
    return GP_OK;
  };
  // Everything that hangs off H = A diag(2 gv) A^T — grad q_sqrt, Wbar, the whole Kuu side — is independent of the
  // Kuf_bar product, which needs only R = W^T (Lq Lq^T - I) and alpha = W^T q_mu.  With early_fork the H chain
  // (the split-K product included) goes to the helper stream and the main stream starts Kuf_bar right away.
  const bool early_fork = white && nK > 0 && n >= 4096 && p->overlap >= 2;
  if (!early_fork) GP_CHECK(late_sums());
  auto h_chain_head = [&]() -> gp_status {
    GemmFlags f;
    // H = A diag(2 gv) A^T  (symmetric, split-K over the frames); u = A gm and grad q_mu += u are fused into it
    {
      int uni = ((n & 3) == 0) ? 1 : 0;
      for (int g = 0; g < G; g++) if (p->gps[g].M != maxM) uni = 0;
      if (n64 > 0) GP_CHECK(launch_gemm_nt_reduce_batched(h, D(S_H), n64, maxM, n, p->nsplit, 1, 1, 2.0, uni));
      if (n64 < G) GP_CHECK(launch_gemm_f32_nt_reduce_batched(h, D(S_H) + n64, G - n64, maxM, n, p->nsplit, 1, 1, 2.0, uni));
    }
    // grad q_sqrt += tril(H Lq)
    f = GemmFlags(); f.triB = TRI_LOWER; f.triC = TRI_LOWER; f.beta = 1.0;
    GP_CHECK(launch_gemm_batched(h, D(S_HLQ), G, maxM, maxM, f));
    return GP_OK;
  };
  auto wbar_chain = [&]() -> gp_status {
    GemmFlags f;
    // T1 = E H
    f = GemmFlags();
    GP_CHECK(launch_gemm_batched(h, D(S_EH), nK, maxM, maxM, f));
    // Wbar = tril(T1 L^T) + tril(mu (L u)^T)
    f = GemmFlags(); f.transB = 1; f.triB = TRI_UPPER; f.triC = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, D(S_WBAR), nK, maxM, maxM, f));
    GP_CHECK(launch_matvec_batched(h, D(S_LU), nK, maxM, 0));
    GP_CHECK(launch_rank1_tril_batched(h, D(S_RANK1), nK, maxM));
    if (!white) {
      GP_CHECK(launch_rank1_tril_batched(h, D(S_WB_R1), nK, maxM));
      f = GemmFlags(); f.triA = TRI_LOWER; f.transB = 1; f.triB = TRI_UPPER; f.triC = TRI_LOWER; f.beta = 1.0;
      GP_CHECK(launch_gemm_batched(h, D(S_WB_L), nK, maxM, maxM, f));
    }
    return GP_OK;
  };
  if (!early_fork) GP_CHECK(h_chain_head());
  if (nK > 0) {
    const bool pre = p->era_ready;   // E, R, alpha were computed on the helper stream during the forward pass
    p->era_ready = false;
    if (pre) {
      GP_HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->ev_era, 0));
    } else {
      // E = Lq Lq^T - I
      f = GemmFlags(); f.triA = TRI_LOWER; f.transB = 1; f.triB = TRI_UPPER;
      GP_CHECK(launch_gemm_batched(h, D(S_E), nK, maxM, maxM, f));
      GP_CHECK(launch_sub_identity_batched(h, D(S_E), nK, maxM));
    }
    if (!early_fork) GP_CHECK(wbar_chain());
    if (!pre) {
      // R = W^T E ; alpha = W^T mu
      f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER;
      GP_CHECK(launch_gemm_batched(h, D(S_R), nK, maxM, maxM, f));
      GP_CHECK(launch_matvec_batched(h, D(S_ALPHA), nK, maxM, 1));
    }
    // From here two independent chains remain: the Kuf side (the big Kuf_bar product and its contraction with
    // dK/dtheta over all frames) and the Kuu side (the Cholesky adjoint, six M x M products, and its contraction
    // over M x M).  The Kuu side is ~1.4 ms of small launches: it runs on the helper stream underneath Kuf_bar.
    const bool forked = (n >= 4096) && p->overlap >= 1 && gp_aux_fork(h);
    gp_status st = GP_OK;
    std::vector<int> np_uu(p->G, 0);
    auto kuu_side = [&]() -> gp_status {
      GemmFlags f;
      // Lbar = -tril(W^T Wbar W^T); P = Phi(L^T Lbar); S = W^T P W
      f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
      GP_CHECK(launch_gemm_batched(h, D(S_T2), nK, maxM, maxM, f));
      f = GemmFlags(); f.transB = 1; f.triB = TRI_UPPER; f.triC = TRI_LOWER; f.alpha = -1.0;
      GP_CHECK(launch_gemm_batched(h, D(S_LBAR), nK, maxM, maxM, f));
      f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
      GP_CHECK(launch_gemm_batched(h, D(S_P), nK, maxM, maxM, f));
      GP_CHECK(launch_phi_batched(h, D(S_P), nK, maxM));
      f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER;
      GP_CHECK(launch_gemm_batched(h, D(S_T3), nK, maxM, maxM, f));
      f = GemmFlags(); f.triB = TRI_LOWER;
      GP_CHECK(launch_gemm_batched(h, D(S_S), nK, maxM, maxM, f));
      // contraction of Kuu_bar with dK(z, z)/d(theta, z), partial sums only: one launch per kernel family (24 launches of
      // a few workgroups each otherwise: 2.6 ms at the end of the helper stream's chain), per GP where a family is mixed
      for (const auto& fam : p->hy_fams) {
        if (!fam.batched) continue;
        int np = 0;
        GP_CHECK(launch_hyper_contract_items(h, fam.type, fam.m, (const HyperItem*)(p->d_misc + p->off_hy_items) + p->G + fam.first,
                                             fam.count, fam.M, fam.M, 0, &np));
        for (int g : fam.gps) np_uu[g] = np;
      }
      for (int g : p->kgps) {
        bool in_batch = false;
        for (const auto& fam : p->hy_fams) if (fam.batched) for (int gg : fam.gps) if (gg == g) in_batch = true;
        if (in_batch) continue;
        const PdgpGP& q = p->gps[g];
        const CondTask& t = p->cb.tasks[g];
        const BwdBufs& bb = p->bw[g];
        const double* z = params + q.off_z;
        const int cb_uf = (n + HY_THREADS - 1) / HY_THREADS;
        double* gz_uu = q.need_z ? bb.gz_part + (size_t)cb_uf * q.M : nullptr;
        GP_CHECK(launch_hyper_contract(h, t.kern, z, q.M, z, q.M, bb.E, q.M, nullptr, nullptr, 1, t.feat, bb.hyp_part_uu,
                                       &np_uu[g], gz_uu));
      }
      return GP_OK;
    };
    if (!forked) GP_CHECK(late_sums());
    if (forked) {            // helper stream: [the late sums and the H chain when forked early,] Kuu side
      if (early_fork) { st = h_chain_head(); if (st == GP_OK) st = wbar_chain(); }
      if (st == GP_OK) st = kuu_side();
      if (st == GP_OK) st = late_sums();      // (at the END of the chain: ahead of the split-K product they delayed it)
      gp_status s2 = gp_aux_end(h);
      if (st == GP_OK) st = s2;
      GP_CHECK(st);
    }
    // Kuf_bar (dense part) = R (A diag(2 gv)), and its contraction with dK/dtheta over all frames.
    // The contractions are one launch per kernel family (item arrays built at bind time).  With exactly two families —
    // the transcription model: stationary activations, spectral-mixture components — whose GPs sit in contiguous runs
    // of the compacted batch, the product is issued family by family, the spectral-mixture family first: its contraction
    // (the long one: 40 % of it matrix-core work) then runs on the side stream UNDERNEATH the second family's product
    // instead of after it, and only the stationary family's short, HBM-bound contraction is left behind the product.
    std::vector<int> np_uf(p->G, 0);
    auto fam_slot0 = [&](const gp_pdgp_plan_s::HyFamily& fam) -> int {    // first slot in the compacted batch, -1 if scattered
      int s0 = -1;
      for (size_t s = 0; s < p->kgps.size(); s++) if (p->kgps[s] == fam.gps[0]) s0 = (int)s;
      for (size_t i = 0; i < fam.gps.size(); i++)
        if (s0 < 0 || s0 + (int)i >= (int)p->kgps.size() || p->kgps[s0 + i] != fam.gps[i]) return -1;
      return s0;
    };
    int kuf_uniform = ((n & 1) == 0) ? 1 : 0;        // every GP of the compacted batch M = maxM (R, A, G: arena buffers, even ld)
    for (int g : p->kgps) if (p->gps[g].M != maxM) kuf_uniform = 0;
    // (slots of the compacted batch keep the GPs' order: its float64 GPs come first, k64 of them)
    int k64 = 0;
    for (int g : p->kgps) if (!p->gps[g].f32) k64++;
    auto kuf_bar = [&](int slot0, int count, int fused_ktype = -1, int fused_f32 = 0) -> gp_status {
      GemmFlags f;
      f.big_tiles = 1; f.scale_mode = 1; f.alpha = 2.0; f.timer = GP_TIMER_KUF_BAR; f.role = 3;
      f.uniform_aligned = kuf_uniform;
      f.a32_ok = 1;          // (float32 GPs: b.R32 is in every problem's xb)
      if (fused_ktype >= 0) {      // the family's Kuf-side contraction as the product's epilogue, nothing stored (one precision per family)
        f.role = 5; f.epilogue = 0; f.aux_x = x; f.aux_ktype = fused_ktype;
        if (fused_f32) return launch_gemm_f32_role(h, D(S_G) + slot0, count, maxM, n, f);
        return launch_gemm_batched(h, D(S_G) + slot0, count, maxM, n, f);
      }
      const int c64 = (slot0 < k64) ? ((slot0 + count <= k64) ? count : k64 - slot0) : 0;
      if (c64 > 0) GP_CHECK(launch_gemm_batched(h, D(S_G) + slot0, c64, maxM, n, f));
      if (c64 < count) GP_CHECK(launch_gemm_f32_role(h, D(S_G) + slot0 + c64, count - c64, maxM, n, f));
      return GP_OK;
    };
    auto kuf_contract = [&](int g) -> gp_status {      // one GP (inducing-input gradients, mixed sizes)
      const PdgpGP& q = p->gps[g];
      const CondTask& t = p->cb.tasks[g];
      const BwdBufs& bb = p->bw[g];
      const double* z = params + q.off_z;
      const double* gm = p->gFmu + (size_t)g * n;
      double* gz_uf = q.need_z ? bb.gz_part : nullptr;
      const int64_t ldN = gp_strip_ld(n, q.f32 != 0);
      return launch_hyper_contract(h, t.kern, z, q.M, x, n, bb.G, ldN, bb.alpha, gm, 0, t.feat, bb.hyp_part, &np_uf[g], gz_uf,
                                   t.Kuf, ldN, q.f32);
    };
    auto contract_family = [&](const gp_pdgp_plan_s::HyFamily& fam) -> gp_status {
#ifdef GP_EXP_SKIP_STAT
      if (!fam.mfma) return GP_OK;     // timing experiment only (wrong gradients): the stationary family's contraction left out
#endif
      if (!fam.batched) { for (int g : fam.gps) GP_CHECK(kuf_contract(g)); return GP_OK; }
      int np = 0;
      GP_CHECK(launch_hyper_contract_items(h, fam.type, fam.m, (const HyperItem*)(p->d_misc + p->off_hy_items) + fam.first,
                                           fam.count, fam.M, n, 0, &np, fam.mfma, x, fam.f32, 1));
      for (int g : fam.gps) np_uf[g] = np;
      return GP_OK;
    };
    // A stationary family with fixed inducing inputs wants Kuf_bar for two sums per GP only: they come out of the product's
    // epilogue (gemm_strip.hip role 5; float64 strips, whole 128-tiles) and neither the strip nor the separate contraction
    // exists.  The choice depends on shapes and kernel types alone — never on the overlap level, whose settings must give
    // bit-identical results — and needs every family in one contiguous run of the compacted batch.
    const int nfam = (int)p->hy_fams.size();
    std::vector<int> fslot(nfam, -1), ffuse(nfam, 0);
    {
      bool contiguous = (white && nfam > 0);
      for (int fi = 0; fi < nfam; fi++) { fslot[fi] = fam_slot0(p->hy_fams[fi]); if (fslot[fi] < 0) contiguous = false; }
      for (int fi = 0; fi < nfam && contiguous; fi++) {
        const auto& fam = p->hy_fams[fi];
        bool ok = kuf_uniform && fam.batched && !fam.mfma && fam.M == maxM &&
                  (fam.f32 ? gemm_f32_fused_contraction_ok(maxM, n, fam.type) : gemm_strip_fused_contraction_ok(maxM, n, fam.type));
        for (int g : fam.gps) if (p->gps[g].need_z || !p->gps[g].need_theta) ok = false;
        ffuse[fi] = ok ? 1 : 0;
      }
    }
    auto kuf_bar_family = [&](int fi) -> gp_status {          // one family's product (contiguous slots), fused form if chosen
      const auto& fam = p->hy_fams[fi];
      if (!ffuse[fi]) return kuf_bar(fslot[fi], fam.count);
      GP_CHECK(kuf_bar(fslot[fi], fam.count, fam.type, fam.f32));
      for (int g : fam.gps) np_uf[g] = fam.f32 ? (gemm_wave_f32_takes(5, maxM, n, kuf_uniform) ? (maxM / 64) * (n / 64) : (maxM / 128) * (n / 128))
                                                : gemm_fused_contraction_records(maxM, n, fam.type);
      return GP_OK;
    };
    bool any_fused = false;
    for (int fi = 0; fi < nfam; fi++) any_fused |= (ffuse[fi] != 0);
    int sm_fam = -1, other_fam = -1, sm_slot = -1, other_slot = -1;
    // (switches.h; -1 = by precision: with float32 strips the spectral-mixture family goes first — its vector-ALU contraction then runs
    // beside the other family's float32 matrix product, which leaves the vector ALU free; the float64 MFMA holds it, so there the
    // stationary family goes first.  cfg3 3.90 -> 3.80 ms, headline 19.47 / 19.56 the other way round)
    int split_mode = gp_switches().kufbar_split;
    if (split_mode < 0) {
      split_mode = 2;
      for (const auto& fam : p->hy_fams) if (fam.mfma && fam.f32) split_mode = 1;
    }
    if (split_mode >= 1 && p->hy_fams.size() == 2 && forked && p->overlap >= 2) {
      for (int fi = 0; fi < 2; fi++) {
        if (p->hy_fams[fi].mfma) sm_fam = fi; else other_fam = fi;
      }
      if (sm_fam >= 0 && other_fam >= 0) { sm_slot = fam_slot0(p->hy_fams[sm_fam]); other_slot = fam_slot0(p->hy_fams[other_fam]); }
    }
    if (sm_slot >= 0 && other_slot >= 0 && split_mode == 2) {
      // the stationary family first: its contraction (an HBM read, next to no arithmetic) goes underneath the
      // spectral-mixture family's product, and the long contraction has the device to itself afterwards
      const auto& fs = p->hy_fams[sm_fam];
      const auto& fo = p->hy_fams[other_fam];
      if (ffuse[other_fam]) {
        GP_CHECK(kuf_bar_family(other_fam));
        GP_CHECK(kuf_bar(sm_slot, fs.count));
      } else {
        GP_CHECK(kuf_bar(other_slot, fo.count));
        const bool side = gp_side_begin(h);
        if (side) {
          gp_status st2 = contract_family(fo);
          gp_status s3 = gp_side_end(h);
          GP_CHECK(st2); GP_CHECK(s3);
        }
        GP_CHECK(kuf_bar(sm_slot, fs.count));
        if (!side) GP_CHECK(contract_family(fo));
      }
      GP_CHECK(contract_family(fs));
    } else if (sm_slot >= 0 && other_slot >= 0) {
      const auto& fs = p->hy_fams[sm_fam];
      const auto& fo = p->hy_fams[other_fam];
      GP_CHECK(kuf_bar(sm_slot, fs.count));
      const bool side = gp_side_begin(h);              // the side stream picks up once that product is through
      if (side) {
        gp_status st2 = contract_family(fs);
        gp_status s3 = gp_side_end(h);
        GP_CHECK(st2); GP_CHECK(s3);
      }
      GP_CHECK(kuf_bar_family(other_fam));
      if (!side) GP_CHECK(contract_family(fs));
      if (!ffuse[other_fam]) GP_CHECK(contract_family(fo));
    } else {
      if (any_fused) { for (int fi = 0; fi < nfam; fi++) GP_CHECK(kuf_bar_family(fi)); }
      else GP_CHECK(kuf_bar(0, nK));
      if (!forked) {
        if (early_fork) { GP_CHECK(h_chain_head()); GP_CHECK(wbar_chain()); }   // (no helper stream to be had)
        GP_CHECK(kuu_side());
      }
      // the families' contractions side by side: the matrix-core ones on this stream, the others (short, HBM-bound) on
      // the side stream
      const bool side = (p->hy_fams.size() > 1) && forked && p->overlap >= 2 && gp_side_begin(h);
      if (side) {
        gp_status st2 = GP_OK;
        for (int fi = 0; fi < nfam; fi++) if (!p->hy_fams[fi].mfma && !ffuse[fi] && st2 == GP_OK) st2 = contract_family(p->hy_fams[fi]);
        gp_status s3 = gp_side_end(h);
        GP_CHECK(st2); GP_CHECK(s3);
      }
      for (int fi = 0; fi < nfam; fi++) if ((!side || p->hy_fams[fi].mfma) && !ffuse[fi]) GP_CHECK(contract_family(p->hy_fams[fi]));
    }
    if (!white) {
      GP_CHECK(launch_matvec_batched(h, D(S_GQ_MU), G, maxM, 1));
      f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER; f.triC = TRI_LOWER;
      GP_CHECK(launch_gemm_batched(h, D(S_GQ_L), G, maxM, maxM, f));
    }
    GP_CHECK(gp_side_join(h));
    GP_CHECK(gp_aux_join(h));
    // all partial sums (Kuf side, Kuu side) are in: ONE finish launch adds them into the gradient vector (48 tiny
    // launches at the end of every step otherwise).  The item array is re-uploaded only when it changes.
    {
      std::vector<HyperFinishItem> items;
      int maxblocks = 0;
      for (int g : p->kgps) {
        const PdgpGP& q = p->gps[g];
        const CondTask& t = p->cb.tasks[g];
        const BwdBufs& bb = p->bw[g];
        const int cb_uf = (n + HY_THREADS - 1) / HY_THREADS, cb_uu = (q.M + HY_THREADS - 1) / HY_THREADS;
        HyperFinishItem it;
        memset(&it, 0, sizeof(it));
        it.k = t.kern; it.p_uf = bb.hyp_part; it.np_uf = np_uf[g]; it.p_uu = bb.hyp_part_uu; it.np_uu = np_uu[g];
        it.gv_sum = bb.gvsum; it.g_theta = grad + q.off_theta; it.n1 = q.M;
        if (q.need_z) {
          it.gz_uf = bb.gz_part; it.cb_uf = cb_uf; it.gz_uu = bb.gz_part + (size_t)cb_uf * q.M; it.cb_uu = cb_uu;
          it.g_z = grad + q.off_z;
        }
        const int blocks = 2 + 2 * t.kern.m + (q.need_z ? (q.M + 255) / 256 : 0);
        if (blocks > maxblocks) maxblocks = blocks;
        items.push_back(it);
      }
      const size_t bytes = items.size() * sizeof(HyperFinishItem);
      char* d_items = p->d_misc + p->off_fin_items;
      if (p->h_fin_items.size() != bytes || memcmp(p->h_fin_items.data(), items.data(), bytes) != 0) {
        p->h_fin_items.assign((const char*)items.data(), (const char*)items.data() + bytes);
        GP_HIP_CHECK(h, hipMemcpyAsync(d_items, p->h_fin_items.data(), bytes, hipMemcpyHostToDevice, h->stream));
      }
      hipLaunchKernelGGL(hyper_finish_items_kernel, dim3(maxblocks, (unsigned)items.size()), dim3(256), 0, h->stream,
                         (const HyperFinishItem*)d_items);
      GP_HIP_CHECK(h, hipGetLastError());
    }
  } else if (!white) {
    GP_CHECK(launch_matvec_batched(h, D(S_GQ_MU), G, maxM, 1));
    f = GemmFlags(); f.transA = 1; f.triA = TRI_UPPER; f.triB = TRI_LOWER; f.triC = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, D(S_GQ_L), G, maxM, maxM, f));
  }
  return GP_OK;
}
