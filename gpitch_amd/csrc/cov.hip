// cov.hip — covariance (Kuu / Kuf) assembly kernels for gfx950.
//
// Replaces the TF op sequence behind `Kern.K(X, X2)`:
//   MercerMatern12sm.K   gpitch/matern12_spectral_mixture.py:102-117 (+ phi_features :123-133)
//   Matern12sm.K         gpitch/matern12_spectral_mixture.py:38-56
//   GPflow Stationary.K  Matern12/32/52/RBF via euclid_dist
//
// HBM-bound by design: one pass, every output element written exactly once with 16-byte stores,
// row-major M x N with the long (frame) dimension contiguous so a wave writes 1 KiB per store.
// The squared distance keeps the reference's matmul-expansion and rounding order
//   r2 = ((-2*(a*b)) + a*a) + b*b,  a = x/l, b = x'/l     (each op rounded separately, no FMA)
// so that coincident inducing/data points give r = sqrt(r2 + 1e-12) exactly as TF does.
#include "common.h"
#include <type_traits>

#ifndef GP_COV_NT_STORE
#define GP_COV_NT_STORE 1
#endif
// Pointers that reach a kernel through a descriptor struct in memory are generic to the compiler, which then emits
// FLAT loads / stores: those count on lgkmcnt as well as vmcnt, so every LDS wait in a row loop (the exp table, the
// staged features, the transposition tile) would also wait for the covariance stores in flight.  Typed as
// address-space-1 they become global_load / global_store.
typedef const double __attribute__((address_space(1))) * cov_gcptr;
typedef double __attribute__((address_space(1))) * cov_gptr;
typedef float __attribute__((address_space(1))) * cov_gfptr;
typedef double cov_d2 __attribute__((ext_vector_type(2)));
typedef float cov_f2 __attribute__((ext_vector_type(2)));
typedef cov_d2 __attribute__((address_space(1))) * cov_gptr2;
typedef cov_f2 __attribute__((address_space(1))) * cov_gfptr2;
#define COV_THREADS 256
#define COV_ROWS 32  // rows (inducing points) handled per block

__device__ __forceinline__ double stat_profile(int type, double r2, double var, const double* __restrict__ etab) {
  // r2 is the literal expansion; the kernels below follow GPflow 0.5 Stationary subclasses
  if (type == GP_KERN_RBF) return var * gp_exp_neg(-r2 * 0.5, etab);
  double r = gp_sqrt_pos(__dadd_rn(r2, 1e-12));
  if (type == GP_KERN_MATERN12) return var * gp_exp_neg(-r, etab);
  if (type == GP_KERN_MATERN32) {
    const double s3 = 1.7320508075688772;
    return var * (1.0 + s3 * r) * gp_exp_neg(-s3 * r, etab);
  }
  // Matern52
  const double s5 = 2.23606797749979;
  return var * (1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * gp_exp_neg(-s5 * r, etab);
}

__device__ __forceinline__ double r2_expand(double a, double aa, double b, double bb) {
  return __dadd_rn(__dadd_rn(-2.0 * __dmul_rn(a, b), aa), bb);
}

// store CPT results of one row as float32 (the strips of a float32 plan; `off` and the leading dimension count floats)
template <int CPT>
__device__ __forceinline__ void cov_store_f32(cov_gptr out, size_t off, int cols_left, const double* res, int accumulate,
                                              int vec_ok) {
  const cov_gfptr o = (cov_gfptr)out + off;
  if (CPT == 2 && vec_ok && cols_left > 1) {
    cov_f2 v = cov_f2{(float)res[0], (float)res[CPT - 1]};
    if (accumulate) { const cov_f2 old = *(cov_gfptr2)o; v.x += old.x; v.y += old.y; }
    *(cov_gfptr2)o = v;
  } else {
#pragma unroll
    for (int c = 0; c < CPT; c++)
      if (c < cols_left) o[c] = accumulate ? o[c] + (float)res[c] : (float)res[c];
  }
}

// Precompute spectral-mixture features, zero-padded to MPAD partials so the consumers can unroll without
// guards:  f[q][j] = sqrt(e_q) cos(2 pi f_q x_j),  f[MPAD + q][j] = sqrt(e_q) sin(2 pi f_q x_j),  q < m;  0 for q >= m.
__global__ void __launch_bounds__(256) sm_features_kernel(DevKern k, const double* __restrict__ x, int n,
                                                          double* __restrict__ f, int mpad,
                                                          const FeatItem* __restrict__ items) {
  if (items) { const FeatItem it = items[blockIdx.z]; k = it.k; f = it.f; if (it.n >= 0) { x = it.x; n = it.n; } }
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int p = blockIdx.y;
  if (j >= n) return;
  double c = 0.0, s = 0.0;
  if (p < k.m) {
    const double* th = k.theta;
    double e = th[2 + p], fr = th[2 + k.m + p];
    double arg = __dmul_rn(__dmul_rn(6.283185307179586, fr), x[j]);
    sincos(arg, &s, &c);
    double se = __dsqrt_rn(e);
    c *= se; s *= se;
  }
  f[(size_t)p * n + j] = c;
  f[(size_t)(p + mpad) * n + j] = s;
}

// MODE 0: stationary (Matern12/32/52/RBF); MODE 1: Mercer spectral mixture (feature form) with envelope ENV
// (0: Matern-1/2, MercerMatern12sm; 2: Matern-5/2, the Matern52 * MercerCosMix product of init_models.py:183-198);
// MODE 2: broadcast cosine form (Matern12sm, Matern32sm).  CPT = columns per thread (16-byte stores when 2).
// MODE 0 takes the stationary kernel type as ENV, so the row loop carries no type switch.
template <int MODE, int CPT, int MPAD, int ENV = 0>
__global__ void __launch_bounds__(COV_THREADS) cov_build_kernel(DevKern k, const double* __restrict__ x1, int n1,
                                                                const double* __restrict__ x2, int n2,
                                                                double* __restrict__ out, int64_t ld,
                                                                int accumulate, double diag_add,
                                                                const double* __restrict__ f1,
                                                                const double* __restrict__ f2, int vec_ok,
                                                                const CovItem* __restrict__ items, int wg_rows, int f32out) {
  // wg_rows: rows handled per workgroup (<= COV_ROWS; fewer for small matrices, which otherwise occupy a handful of CUs)
  if (items) {
    const CovItem it = items[blockIdx.z];
    k = it.k; x1 = it.x1; n1 = it.n1; out = it.out; ld = it.ld;
    if (it.n2 >= 0) { x2 = it.x2; n2 = it.n2; }         // n2 < 0: every item shares the launch's x2 / n2 (the frames)
    accumulate = it.accumulate; diag_add = it.diag_add; f1 = it.f1; f2 = it.f2; vec_ok = it.vec_ok; f32out = it.f32out;
    if ((int)(blockIdx.y * wg_rows) >= n1) return;       // the grid is sized for the largest item
  }
  const cov_gcptr gx1 = (cov_gcptr)x1, gx2 = (cov_gcptr)x2, gf1 = (cov_gcptr)f1, gf2 = (cov_gcptr)f2;
  const cov_gptr gout = (cov_gptr)out;
  extern __shared__ double smem[];  // MODE 1: z-features for this block's rows [COV_ROWS][2m]
  __shared__ double row_a[COV_ROWS];  // x1[i] / lengthscale (the exact quotient, computed once per row, not per entry)
  __shared__ double etab[GP_EXP_TAB];
  gp_exp_tab_init(etab);
  const double* th = k.theta;
  const double var = th[0];
  const double ls = th[1];
  const int m = k.m;
  const int j0 = (blockIdx.x * COV_THREADS + threadIdx.x) * CPT;
  const int i0 = blockIdx.y * wg_rows;
  const int iend = min(i0 + wg_rows, n1);
  // K(x, x) + diag_add I: only the workgroups whose column range meets their row range carry the diagonal test
  const int jb0 = blockIdx.x * COV_THREADS * CPT;
  const bool self_cov = (x2 == x1) && (diag_add != 0.0) && (jb0 < i0 + wg_rows) && (jb0 + COV_THREADS * CPT > i0);

  if (threadIdx.x < COV_ROWS) row_a[threadIdx.x] = (i0 + (int)threadIdx.x < n1) ? gx1[i0 + threadIdx.x] / th[1] : 0.0;   // (rows beyond wg_rows unused)
  if (MODE != 1) __syncthreads();
  if (MODE == 1) {
    // stage this block's row features: smem[(i - i0) * 2*MPAD + q] = f1[q][i]
    for (int t = threadIdx.x; t < COV_ROWS * 2 * MPAD; t += COV_THREADS) {
      int q = t / COV_ROWS, ii = t % COV_ROWS;
      smem[ii * 2 * MPAD + q] = (ii < wg_rows && i0 + ii < n1) ? gf1[(size_t)q * n1 + i0 + ii] : 0.0;
    }
    __syncthreads();
  }
  if (j0 >= n2) return;

  double b[CPT], bb[CPT], xb[CPT];
#pragma unroll
  for (int c = 0; c < CPT; c++) {
    int j = min(j0 + c, n2 - 1);
    xb[c] = gx2[j];
    b[c] = xb[c] / ls;
    bb[c] = __dmul_rn(b[c], b[c]);
  }

  if (MODE == 1) {
    // column features live in registers for the whole row loop (2*MPAD values per column, no guards)
    double fx[CPT][2 * MPAD];
#pragma unroll
    for (int q = 0; q < 2 * MPAD; q++) {
#pragma unroll
      for (int c = 0; c < CPT; c++) {
        int j = min(j0 + c, n2 - 1);
        fx[c][q] = gf2[(size_t)q * n2 + j];
      }
    }
    for (int i = i0; i < iend; i++) {
      double a = row_a[i - i0], aa = __dmul_rn(a, a);
      const double* fz = &smem[(i - i0) * 2 * MPAD];
      double acc[CPT];
#pragma unroll
      for (int c = 0; c < CPT; c++) acc[c] = 0.0;
#pragma unroll
      for (int q = 0; q < 2 * MPAD; q++) {
        double z = fz[q];
#pragma unroll
        for (int c = 0; c < CPT; c++) acc[c] = fma(z, fx[c][q], acc[c]);
      }
      double res[CPT];
#pragma unroll
      for (int c = 0; c < CPT; c++) {
        double r = gp_sqrt_pos(__dadd_rn(r2_expand(a, aa, b[c], bb[c]), 1e-12));
        if (ENV == 0) {
          res[c] = var * gp_exp_neg(-r, etab) * acc[c];
        } else {   // GPflow Matern52.K profile
          const double s5 = 2.23606797749979;
          res[c] = var * ((1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * gp_exp_neg(-s5 * r, etab)) * acc[c];
        }
        if (self_cov && i == j0 + c) res[c] += diag_add;
      }
      if (f32out) { cov_store_f32<CPT>(gout, (size_t)i * ld + j0, n2 - j0, res, accumulate, vec_ok); continue; }
      const cov_gptr o = gout + (size_t)i * ld + j0;
      if (CPT == 2 && vec_ok && j0 + 1 < n2) {
        cov_d2 v = cov_d2{res[0], res[CPT - 1]};
        if (accumulate) { const cov_d2 old = *(cov_gptr2)o; v.x += old.x; v.y += old.y; }
        *(cov_gptr2)o = v;
      } else {
#pragma unroll
        for (int c = 0; c < CPT; c++)
          if (j0 + c < n2) o[c] = accumulate ? o[c] + res[c] : res[c];
      }
    }
    return;
  }

  // the diagonal test lives in its own copy of the row loop (a select per entry otherwise: 8 of ~90 instructions)
  auto rows = [&](auto diag_tag) {
    constexpr bool DIAG = decltype(diag_tag)::value;
    for (int i = i0; i < iend; i++) {
      double xa = gx1[i];
      double res[CPT];
      if (MODE == 0) {
        double a = row_a[i - i0], aa = __dmul_rn(a, a);
  #pragma unroll
        for (int c = 0; c < CPT; c++) res[c] = stat_profile(MODE == 0 ? ENV : k.type, r2_expand(a, aa, b[c], bb[c]), var, etab);
      } else {
        // Matern12sm (m12sm.py:46-56) / Matern32sm (kernels.py:232-247): r = sqrt((x - x' + 1e-12)^2)
  #pragma unroll
        for (int c = 0; c < CPT; c++) {
          double d = __dadd_rn(__dadd_rn(xa, -xb[c]), 1e-12);
          double r = __dsqrt_rn(__dmul_rn(d, d));
          double s = 0.0;
          for (int p = 0; p < m; p++)
            s += th[2 + p] * cos(__dmul_rn(__dmul_rn(6.283185307179586, th[2 + m + p]), r));
          if (k.type == GP_KERN_MATERN12SM) {
            res[c] = var * exp(-(r / ls)) * s;
          } else {   // Matern32sm: r1 = sqrt(3) r / l, (1 + r1) exp(-r1) sum_k variance_k cos(2 pi f_k r)
            const double r1 = 1.7320508075688772 * (r / ls);
            res[c] = var * ((1.0 + r1) * exp(-r1)) * s;
          }
        }
      }
  #pragma unroll
      for (int c = 0; c < CPT; c++)
        if (DIAG && i == j0 + c) res[c] += diag_add;
      if (f32out) { cov_store_f32<CPT>(gout, (size_t)i * ld + j0, n2 - j0, res, accumulate, vec_ok); continue; }
      const cov_gptr o = gout + (size_t)i * ld + j0;
      if (CPT == 2 && vec_ok && j0 + 1 < n2) {
        cov_d2 v = cov_d2{res[0], res[CPT - 1]};
        if (accumulate) { const cov_d2 old = *(cov_gptr2)o; v.x += old.x; v.y += old.y; }
        if (GP_COV_NT_STORE && !accumulate) __builtin_nontemporal_store(v, (cov_gptr2)o);
        else *(cov_gptr2)o = v;
      } else {
  #pragma unroll
        for (int c = 0; c < CPT; c++)
          if (j0 + c < n2) o[c] = accumulate ? o[c] + res[c] : res[c];
      }
    }
  };
  if (self_cov) rows(std::true_type{}); else rows(std::false_type{});
}

__global__ void __launch_bounds__(256) cov_diag_kernel(DevKern k, int n, double* __restrict__ out, int accumulate) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double* th = k.theta;
  double v = th[0];
  if (gp_kern_kdiag_energy(k.type)) {
    double s = th[2];
    for (int p = 1; p < k.m; p++) s += th[2 + p];
    v = v * s;
  }
  out[j] = accumulate ? out[j] + v : v;
}

int sm_mpad(int m) { return m <= 0 ? 0 : ((m + 3) / 4) * 4; }

size_t kernel_build_feat_ws_doubles(int m, int n1, int n2) {
  if (m <= 0) return 0;
  const int mp = sm_mpad(m);
  return gp_align_up((size_t)2 * mp * n1, 32) + gp_align_up((size_t)2 * mp * n2, 32);
}

template <int MPAD>
static void launch_mercer(gp_handle h, dim3 grid, DevKern k, const double* x1, int n1, const double* x2, int n2,
                          double* out, int64_t ld, int accumulate, double diag_add, const double* f1, const double* f2,
                          int vec_ok, const CovItem* items, int rows = COV_ROWS, int f32out = 0) {
  size_t sh = (size_t)COV_ROWS * 2 * MPAD * sizeof(double);
  if (k.type == GP_KERN_MERCER_MATERN12SM)
    hipLaunchKernelGGL((cov_build_kernel<1, 2, MPAD, 0>), grid, dim3(COV_THREADS), sh, h->stream, k, x1, n1, x2, n2, out,
                       ld, accumulate, diag_add, f1, f2, vec_ok, items, rows, f32out);
  else
    hipLaunchKernelGGL((cov_build_kernel<1, 2, MPAD, 2>), grid, dim3(COV_THREADS), sh, h->stream, k, x1, n1, x2, n2, out,
                       ld, accumulate, diag_add, f1, f2, vec_ok, items, rows, f32out);
}

// rows per workgroup: the full COV_ROWS for strips; small matrices (window-sized problems, Kuu) get more, smaller
// workgroups — a 64 x 2001 build is otherwise 8 workgroups on a 256-CU device
static int cov_rows_for(int n1, int n2) {
  const int64_t e = (int64_t)n1 * n2;
  if (e >= (1 << 20)) return COV_ROWS;
  if (e >= (1 << 17)) return 8;
  return 4;
}

// K = sum_p K_p for up to 8 MercerMatern12sm kernels in ONE pass (SGPRSS: the summed source kernel of sgpr_ss.py:29-71,
// `kern = sum of per-source kernels`): per entry the arithmetic of cov_build_kernel<1, 2, MPAD, 0> for every kernel, added in
// kernel order — the value that P accumulate launches leave (float64: bit-identical; float32 output: one rounding instead of P).
// The accumulate launches re-read and re-write the whole M x N strip P - 1 times (0.34 ms of a 5.1-ms evaluation at
// N = 65536, M = 512, P = 5).  Row accumulators of the workgroup's COV_ROWS rows stay in registers while the kernels go by.
struct CovSumArgs { int P; int pad_; DevKern k[8]; const double* f1[8]; const double* f2[8]; };
template <int MPAD>
__global__ void __launch_bounds__(COV_THREADS) cov_mercer_sum_kernel(CovSumArgs a, const double* __restrict__ x1, int n1,
                                                                     const double* __restrict__ x2, int n2,
                                                                     double* __restrict__ out, int64_t ld, double diag_add,
                                                                     int vec_ok, int wg_rows, int f32out) {
  constexpr int CPT = 2;
  const cov_gcptr gx1 = (cov_gcptr)x1, gx2 = (cov_gcptr)x2;
  const cov_gptr gout = (cov_gptr)out;
  __shared__ double fzs[COV_ROWS * 2 * MPAD];
  __shared__ double row_a[COV_ROWS];
  __shared__ double etab[GP_EXP_TAB];
  gp_exp_tab_init(etab);
  const int j0 = (blockIdx.x * COV_THREADS + threadIdx.x) * CPT;
  const int i0 = blockIdx.y * wg_rows;
  const int jb0 = blockIdx.x * COV_THREADS * CPT;
  const bool self_cov = (x2 == x1) && (diag_add != 0.0) && (jb0 < i0 + wg_rows) && (jb0 + COV_THREADS * CPT > i0);
  double res[COV_ROWS][CPT];
#pragma unroll
  for (int ii = 0; ii < COV_ROWS; ii++)
#pragma unroll
    for (int c = 0; c < CPT; c++) res[ii][c] = 0.0;
  double xb[CPT];
#pragma unroll
  for (int c = 0; c < CPT; c++) xb[c] = gx2[min(j0 + c, n2 - 1)];
  for (int p = 0; p < a.P; p++) {
    const double* th = a.k[p].theta;
    const double var = th[0], ls = th[1];
    const cov_gcptr gf1 = (cov_gcptr)a.f1[p], gf2 = (cov_gcptr)a.f2[p];
    __syncthreads();        // the previous kernel's row tables are still being read
    if (threadIdx.x < COV_ROWS) row_a[threadIdx.x] = (i0 + (int)threadIdx.x < n1) ? gx1[i0 + threadIdx.x] / ls : 0.0;
    for (int t = threadIdx.x; t < COV_ROWS * 2 * MPAD; t += COV_THREADS) {
      const int q = t / COV_ROWS, ii = t % COV_ROWS;
      fzs[ii * 2 * MPAD + q] = (ii < wg_rows && i0 + ii < n1) ? gf1[(size_t)q * n1 + i0 + ii] : 0.0;
    }
    __syncthreads();
    if (j0 < n2) {
      double b[CPT], bb[CPT], fx[CPT][2 * MPAD];
#pragma unroll
      for (int c = 0; c < CPT; c++) { b[c] = xb[c] / ls; bb[c] = __dmul_rn(b[c], b[c]); }
#pragma unroll
      for (int q = 0; q < 2 * MPAD; q++)
#pragma unroll
        for (int c = 0; c < CPT; c++) fx[c][q] = gf2[(size_t)q * n2 + min(j0 + c, n2 - 1)];
#pragma unroll
      for (int ii = 0; ii < COV_ROWS; ii++) {
        if (ii < wg_rows) {
          const double av = row_a[ii], aa = __dmul_rn(av, av);
          const double* fz = &fzs[ii * 2 * MPAD];
          double acc[CPT];
#pragma unroll
          for (int c = 0; c < CPT; c++) acc[c] = 0.0;
#pragma unroll
          for (int q = 0; q < 2 * MPAD; q++) {
            const double z = fz[q];
#pragma unroll
            for (int c = 0; c < CPT; c++) acc[c] = fma(z, fx[c][q], acc[c]);
          }
#pragma unroll
          for (int c = 0; c < CPT; c++) {
            const double r = gp_sqrt_pos(__dadd_rn(r2_expand(av, aa, b[c], bb[c]), 1e-12));
            res[ii][c] += var * gp_exp_neg(-r, etab) * acc[c];
          }
        }
      }
    }
  }
  if (j0 >= n2) return;
#pragma unroll
  for (int ii = 0; ii < COV_ROWS; ii++) {
    const int i = i0 + ii;
    if (ii < wg_rows && i < n1) {
#pragma unroll
      for (int c = 0; c < CPT; c++)
        if (self_cov && i == j0 + c) res[ii][c] += diag_add;
      if (f32out) { cov_store_f32<CPT>(gout, (size_t)i * ld + j0, n2 - j0, res[ii], 0, vec_ok); continue; }
      const cov_gptr o = gout + (size_t)i * ld + j0;
      if (vec_ok && j0 + 1 < n2) *(cov_gptr2)o = cov_d2{res[ii][0], res[ii][1]};
      else {
#pragma unroll
        for (int c = 0; c < CPT; c++)
          if (j0 + c < n2) o[c] = res[ii][c];
      }
    }
  }
}

// true = taken.  kernels: P MercerMatern12sm kernels of the same padded partial count; feats[p] = the kernel's feature
// workspace (layout of launch_sm_features: x1 block, then x2 block), already current.
bool launch_kernel_build_sum(gp_handle h, const DevKern* kernels, double* const* feats, int P, const double* x1, int n1,
                             const double* x2, int n2, double* out, int64_t ld, double diag_add, int f32out, gp_status* st) {
  const bool enabled = gp_switches().cov_sum != 0;
  if (!enabled || P < 2 || P > 8 || n1 <= 0 || n2 <= 0) return false;
  if (x2 == nullptr) { x2 = x1; n2 = n1; }
  const int mp = sm_mpad(kernels[0].m);
  if (mp != 4 && mp != 8) return false;
  for (int p = 0; p < P; p++)
    if (kernels[p].type != GP_KERN_MERCER_MATERN12SM || sm_mpad(kernels[p].m) != mp || !feats[p]) return false;
  if (f32out && x2 == x1 && diag_add != 0.0) return false;
  const bool big = (int64_t)n1 * n2 >= (1 << 20);
  GpTimerScope ts(h, !big ? GP_TIMER_SMALL_GEMM : GP_TIMER_KUF_BUILD_SM);
  CovSumArgs a;
  a.P = P; a.pad_ = 0;
  for (int p = 0; p < 8; p++) {
    const int q = p < P ? p : 0;
    a.k[p] = kernels[q];
    a.f1[p] = feats[q];
    a.f2[p] = (x2 == x1) ? feats[q] : feats[q] + gp_align_up((size_t)2 * mp * n1, 32);
  }
  const int vec_ok = ((ld % 2) == 0) && ((((uintptr_t)out) & 15) == 0);
  const int rows = cov_rows_for(n1, n2);
  dim3 grid((n2 + COV_THREADS * 2 - 1) / (COV_THREADS * 2), (n1 + rows - 1) / rows);
  if (mp == 4) hipLaunchKernelGGL((cov_mercer_sum_kernel<4>), grid, dim3(COV_THREADS), 0, h->stream, a, x1, n1, x2, n2, out, ld,
                                  diag_add, vec_ok, rows, f32out);
  else hipLaunchKernelGGL((cov_mercer_sum_kernel<8>), grid, dim3(COV_THREADS), 0, h->stream, a, x1, n1, x2, n2, out, ld,
                          diag_add, vec_ok, rows, f32out);
  hipError_t e = hipGetLastError();
  *st = (e == hipSuccess) ? GP_OK : gp_fail(h, GP_ERR_HIP, hipGetErrorString(e));
  return true;
}

gp_status launch_kernel_build(gp_handle h, DevKern k, const double* x1, int n1, const double* x2, int n2,
                              double* out, int64_t ld, int accumulate, double diag_add, double* feat_ws, int feat_ready,
                              int f32out) {
  // feat_ready: the feature tables in feat_ws (layout of launch_sm_features: x1 block, then x2 block) are current
  if (n1 <= 0 || n2 <= 0) return GP_OK;
  if (x2 == nullptr) { x2 = x1; n2 = n1; }
  const int vec_ok = ((ld % 2) == 0) && ((((uintptr_t)out) & 15) == 0);
  if (f32out && x2 == x1 && diag_add != 0.0) return gp_fail(h, GP_ERR_UNSUPPORTED, "float32 output is for Kuf strips, not Kuu");
  const bool big = (int64_t)n1 * n2 >= (1 << 20);   // M x N strips; the small Kuu builds are booked elsewhere
  GpTimerScope ts(h, !big ? GP_TIMER_SMALL_GEMM
                          : (gp_kern_is_mercer(k.type) ? GP_TIMER_KUF_BUILD_SM : GP_TIMER_KUF_BUILD));
  if (gp_kern_is_mercer(k.type)) {
    if (k.m < 1 || k.m > 32) return gp_fail(h, GP_ERR_UNSUPPORTED, "num_partials must be in [1, 32]");
    if (!feat_ws) return gp_fail(h, GP_ERR_WORKSPACE, "feature workspace missing");
    const int mp = sm_mpad(k.m);
    double* f1 = feat_ws;
    double* f2 = (x2 == x1) ? f1 : feat_ws + gp_align_up((size_t)2 * mp * n1, 32);
    dim3 g1((n1 + 255) / 256, mp);
    if (!feat_ready)
      hipLaunchKernelGGL(sm_features_kernel, g1, dim3(256), 0, h->stream, k, x1, n1, f1, mp, (const FeatItem*)nullptr);
    if (x2 != x1 && !feat_ready) {
      dim3 g2((n2 + 255) / 256, mp);
      hipLaunchKernelGGL(sm_features_kernel, g2, dim3(256), 0, h->stream, k, x2, n2, f2, mp, (const FeatItem*)nullptr);
    }
    const int rows = cov_rows_for(n1, n2);
    dim3 grid((n2 + COV_THREADS * 2 - 1) / (COV_THREADS * 2), (n1 + rows - 1) / rows);
    switch (mp) {
      case 4: launch_mercer<4>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
      case 8: launch_mercer<8>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
      case 12: launch_mercer<12>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
      case 16: launch_mercer<16>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
      case 20: launch_mercer<20>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
      case 24: launch_mercer<24>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
      case 28: launch_mercer<28>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
      default: launch_mercer<32>(h, grid, k, x1, n1, x2, n2, out, ld, accumulate, diag_add, f1, f2, vec_ok, (const CovItem*)nullptr, rows, f32out); break;
    }
  } else if (gp_kern_is_broadcast(k.type)) {
    if (k.m < 1) return gp_fail(h, GP_ERR_BAD_ARG, "num_partials must be >= 1");
    const int rows = cov_rows_for(n1, n2);
    dim3 grid((n2 + COV_THREADS - 1) / COV_THREADS, (n1 + rows - 1) / rows);
    hipLaunchKernelGGL((cov_build_kernel<2, 1, 1>), grid, dim3(COV_THREADS), 0, h->stream, k, x1, n1, x2, n2, out, ld,
                       accumulate, diag_add, nullptr, nullptr, vec_ok, (const CovItem*)nullptr, rows, f32out);
  } else if (k.type >= GP_KERN_MATERN12 && k.type <= GP_KERN_RBF) {
    const int rows = cov_rows_for(n1, n2);
    dim3 grid((n2 + COV_THREADS * 2 - 1) / (COV_THREADS * 2), (n1 + rows - 1) / rows);
#define COV_STAT(T) hipLaunchKernelGGL((cov_build_kernel<0, 2, 1, T>), grid, dim3(COV_THREADS), 0, h->stream, k, x1, n1, \
                                      x2, n2, out, ld, accumulate, diag_add, nullptr, nullptr, vec_ok, (const CovItem*)nullptr, rows, f32out)
    switch (k.type) {
      case GP_KERN_MATERN12: COV_STAT(GP_KERN_MATERN12); break;
      case GP_KERN_MATERN32: COV_STAT(GP_KERN_MATERN32); break;
      case GP_KERN_MATERN52: COV_STAT(GP_KERN_MATERN52); break;
      default: COV_STAT(GP_KERN_RBF); break;
    }
#undef COV_STAT
  } else {
    return gp_fail(h, GP_ERR_BAD_ARG, "unknown kernel type");
  }
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// Matrix-core form of the Mercer Kuf strip.  K[i][j] = var phi(r_ij) * sum_f Zf[i][f] Xf[j][f] with 2m features per
// point: the feature dot product is a (rows x 2m) x (2m x cols) GEMM on v_mfma_f64_16x16x4_f64, the vector pipe keeps
// only the envelope (distance, sqrt, exp).  One workgroup owns 256 columns (64 per wavefront) and walks all rows in
// chunks of 32:
//  - the column (frame) feature fragments of a wavefront's 64 columns do not change over the row loop: they are loaded
//    once, straight from the feature table, and stay in registers (4 column tiles x 2m/4 k-steps);
//  - the row (inducing-point) features of a chunk are staged once per workgroup in LDS and each A fragment read feeds
//    four MFMAs (one per column tile);
//  - the MFMA result layout puts 16 consecutive columns of FOUR different rows on a wavefront's lanes, so a direct store
//    writes 4 x 128-byte pieces per instruction (this kernel's first form: 2.1 TB/s).  Results go through a per-wave
//    LDS tile instead and leave as whole 512-byte row segments, 16 bytes per lane, with non-temporal stores — the
//    four wavefronts of a workgroup cover 2 KiB of every row.
// Kuf strips only (no diagonal term, no accumulation); blockIdx.y = item.
#ifndef CVM_ROWS
#define CVM_ROWS 64                  // rows per staged chunk (4 row tiles; 32: +2.5 % per launch)
#endif
#ifndef CVM_CT
#define CVM_CT 2                     // 16-column tiles per wavefront (see the kernel)
#endif
#define CVM_SEP 1.0                  // separable envelope: |z - x| / lengthscale at least this for every entry of a tile
// Measured on MI355X (profiles/r02): direct stores 0.82 ms per 12-GP launch; LDS-transposed 512-byte row stores 0.78;
// address-space-1 pointers (the descriptor's pointers are generic, so the stores were FLAT and every LDS wait of the
// row loop waited for them) 0.62.  What is left is arithmetic: the float64 MFMA and the float64 vector pipe do not
// overlap on this chip (the matrix rate equals the vector rate: the same units), so the 40 MFMAs (2.6 k cycles) and
// the ~530 vector instructions (2.9 k cycles) of a 16 x 64 tile add up whether one wavefront interleaves them (a
// software-pipelined form of this loop, sched_barrier-pinned, ran 0.82 ms) or two co-resident wavefronts run them out
// of phase (splitting the rows over more workgroups to shorten the last round changed nothing either: 0.65 ms).  The row
// range can be split over blockIdx.z; the launcher does so only when the column blocks alone would not fill the device.
// CT = 16-column tiles per wavefront: 4 (four wavefronts of 64 columns, two per SIMD) or 2 (eight wavefronts of 32
// columns, four per SIMD: half the loop-invariant column fragments per wavefront, twice the wavefronts to hide the LDS
// round trips of the transposition and the store addresses behind).  Either way a workgroup owns 256 columns.
template <int MPAD, int ENV, int CT>
__global__ void __launch_bounds__(1024 / CT, CT == 2 ? 2 : 2) cov_mercer_mfma_kernel(const CovItem* __restrict__ items,
                                                                                     const double* __restrict__ x2s, int n2s,
                                                                                     int row_seg) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int NF = 2 * MPAD;          // features per point (a multiple of 8)
  constexpr int KS = NF / 4;            // MFMA k-steps
  constexpr int FS = NF + 1;            // odd LDS stride
  constexpr int WCOLS = 16 * CT;        // columns per wavefront
  constexpr int NWV = 256 / WCOLS;      // wavefronts per workgroup
  constexpr int NTH = 64 * NWV;
  constexpr int TS = WCOLS + 2;         // LDS row stride of the transposition tile (doubles): rows stay 16-byte aligned
  const CovItem it = items[blockIdx.y];
  const cov_gcptr x1 = (cov_gcptr)it.x1;
  const int n1 = it.n1;
  const cov_gcptr x2 = (cov_gcptr)((it.n2 >= 0) ? it.x2 : x2s);
  const int n2 = (it.n2 >= 0) ? it.n2 : n2s;
  const cov_gptr out = (cov_gptr)it.out;
  const cov_gcptr gf1 = (cov_gcptr)it.f1, gf2 = (cov_gcptr)it.f2, th = (cov_gcptr)it.k.theta;
  const int64_t ld = it.ld;
  __shared__ double zf[CVM_ROWS * FS];            // row features of the current 32-row chunk
  __shared__ double rowa[CVM_ROWS];
  __shared__ double tile_lo[CVM_ROWS / 16], tile_hi[CVM_ROWS / 16];
  __shared__ double etab[GP_EXP_TAB];
  __shared__ __attribute__((aligned(16))) double tbuf[NWV][16 * TS];      // per-wave 16 x WCOLS transposition tile
  gp_exp_tab_init(etab);
  const double var = th[0], ls = th[1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  const int jw = blockIdx.x * 256 + wave * WCOLS;                     // first column of this wavefront
  const int rbeg = blockIdx.z * row_seg, rend = min(n1, rbeg + row_seg);   // this workgroup's rows (row_seg % CVM_ROWS == 0)
  if ((int)(blockIdx.x * 256) >= n2 || rbeg >= n1) return;
  // column-side operands, loop-invariant: B fragments B[k = 4 s + kq][j = lc] of each column tile, scaled inputs
  double bfr[CT][KS], bsc[CT], bb[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    const int j = jw + ct * 16 + lc;
    const bool on = (j < n2);
    const int jc = on ? j : n2 - 1;
#pragma unroll
    for (int s = 0; s < KS; s++) bfr[ct][s] = on ? gf2[(size_t)(4 * s + kq) * n2 + jc] : 0.0;
    bsc[ct] = x2[jc] / ls;
    bb[ct] = __dmul_rn(bsc[ct], bsc[ct]);
  }
  double* tw = tbuf[wave];
  constexpr int PPR = WCOLS / 2, RPI = 64 / PPR;           // store phase: RPI rows x PPR column pairs per instruction
  const int srow = lane / PPR, scol = (lane % PPR) * 2;
  const bool vec = it.vec_ok && ((jw & 1) == 0);
  // ---- separable envelope -------------------------------------------------------------------------------------
  // Away from the diagonal band the envelope factorises: with a = z_i / l, b = x_j / l and every a of a row tile on one
  // side of every b of this wavefront's 64 columns by at least CVM_SEP,
  //     exp(-s r_ij),  r_ij = sqrt((a - b)^2 + 1e-12)  ->  exp(-s (a_i - bmax)) * exp(-s (bmax - b_j))      (a above b)
  //                                                         exp(-s (bmin - a_i)) * exp(-s (b_j - bmin))      (a below b)
  // (s = 1 for the Matern-1/2 envelope, sqrt 5 for Matern-5/2): one exp per ROW of the tile and two multiplies per
  // entry instead of a square root and an exp per entry (~27 of the ~30 float64 vector instructions an entry costs — as
  // much matrix-core-equivalent time as the 40 MFMAs of the tile).  What is dropped is the 1e-12 under the root and the
  // rounding of the reference's expanded square (GPflow's square_dist): relative 1e-12 / (2 |a - b|) = 5e-13 at the
  // threshold |a - b| >= 1 (a lengthscale apart), below the 1e-11 the kernel tests hold; inside the band — and for any
  // tile that straddles it — the entry-by-entry form runs as before.  Column factors are loop-invariant registers.
  double bmin_w, bmax_w;
  {
    double lo = bsc[0], hi = bsc[0];
#pragma unroll
    for (int ct = 1; ct < CT; ct++) { lo = fmin(lo, bsc[ct]); hi = fmax(hi, bsc[ct]); }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
    bmin_w = lo; bmax_w = hi;
  }
  constexpr double SENV = (ENV == 0) ? 1.0 : 2.23606797749979;
  const bool sep_ok = (SENV * (bmax_w - bmin_w) < 300.0);       // column factors stay normal numbers
  __syncthreads();                                                 // etab is ready
  double cfp[CT], cfn[CT];                                         // var * exp(-s (bmax - b)),  var * exp(-s (b - bmin))
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    cfp[ct] = var * gp_exp_neg(-SENV * (bmax_w - bsc[ct]), etab);
    cfn[ct] = var * gp_exp_neg(-SENV * (bsc[ct] - bmin_w), etab);
  }
  for (int r0 = rbeg; r0 < rend; r0 += CVM_ROWS) {
    __syncthreads();
    for (int t = tid; t < CVM_ROWS * NF; t += NTH) {
      const int f = t / CVM_ROWS, ii = t % CVM_ROWS;
      zf[ii * FS + f] = (r0 + ii < n1) ? gf1[(size_t)f * n1 + r0 + ii] : 0.0;
    }
    if (tid < CVM_ROWS) rowa[tid] = x1[min(r0 + tid, n1 - 1)] / ls;        // (rows past the end repeat the last one)
    else if (tid < CVM_ROWS + CVM_ROWS / 16) {       // scaled-input range of every 16-row tile of the chunk
      const int t16 = (tid - CVM_ROWS) * 16;
      double lo = x1[min(r0 + t16, n1 - 1)] / ls, hi = lo;
      for (int q = 1; q < 16; q++) { const double v = x1[min(r0 + t16 + q, n1 - 1)] / ls; lo = fmin(lo, v); hi = fmax(hi, v); }
      tile_lo[tid - CVM_ROWS] = lo; tile_hi[tid - CVM_ROWS] = hi;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < CVM_ROWS / 16; rt++) {
      if (r0 + rt * 16 >= n1) break;
      d4 acc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ct++) acc[ct] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const double af = zf[(rt * 16 + lc) * FS + 4 * s + kq];            // A[i = lc][k = kq]
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bfr[ct][s], acc[ct], 0, 0, 0);
      }
      // envelope; element r of acc[ct]: row 16 rt + kq + 4 r, column 16 ct + lc of this wavefront's strip
      const bool above = sep_ok && (tile_lo[rt] - bmax_w >= CVM_SEP), below = sep_ok && (bmin_w - tile_hi[rt] >= CVM_SEP);
      if (above || below) {                 // (wavefront-uniform)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double a = rowa[rt * 16 + kq + 4 * r];
          const double rf = gp_exp_neg(-SENV * (above ? a - bmax_w : bmin_w - a), etab);
#pragma unroll
          for (int ct = 0; ct < CT; ct++) {
            double e = rf * (above ? cfp[ct] : cfn[ct]);
            if (ENV != 0) { const double rr = fabs(a - bsc[ct]); e *= 1.0 + SENV * rr + (5.0 / 3.0) * (rr * rr); }
            tw[(kq + 4 * r) * TS + ct * 16 + lc] = e * acc[ct][r];
          }
        }
      } else
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int li = rt * 16 + kq + 4 * r;
        const double a = rowa[li], aa = __dmul_rn(a, a), m2a = -2.0 * a;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
          // (-2 (a b) + a a) + b b, every operation rounded on its own (scaling by -2 is exact)
          const double rr = gp_sqrt_pos(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m2a, bsc[ct]), aa), bb[ct]), 1e-12));
          double env;
          if (ENV == 0) env = gp_exp_neg(-rr, etab);
          else { const double s5 = 2.23606797749979; env = (1.0 + s5 * rr + (5.0 / 3.0) * (rr * rr)) * gp_exp_neg(-s5 * rr, etab); }
          tw[(kq + 4 * r) * TS + ct * 16 + lc] = var * env * acc[ct][r];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();          // LDS operations of one wavefront complete in order
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 16 / RPI; q++) {
        const int lr = RPI * q + srow, i = r0 + rt * 16 + lr, j = jw + scol;
        const cov_d2 v = *reinterpret_cast<const cov_d2*>(tw + lr * TS + scol);
        if (i < n1 && j < n2) {
          if (it.f32out) {
            const cov_gfptr o = (cov_gfptr)out + (size_t)i * ld + j;
            if (vec && j + 1 < n2) __builtin_nontemporal_store(cov_f2{(float)v.x, (float)v.y}, (cov_gfptr2)o);
            else { o[0] = (float)v.x; if (j + 1 < n2) o[1] = (float)v.y; }
          } else {
            const cov_gptr o = out + (size_t)i * ld + j;
            if (vec && j + 1 < n2) __builtin_nontemporal_store(v, (cov_gptr2)o);
            else { o[0] = v.x; if (j + 1 < n2) o[1] = v.y; }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();          // the tile is rewritten by the next row tile
    }
  }
}

// Lean free-running form (round 3).  Measured on MI355X (same box, tools/bench_kuf.py): whatever the staged forms above
// do to the vector work, the m = 20 build takes 0.45 ms; with its stores removed 0.38 ms; the 20 MFMAs of a tile alone
// are 0.21 ms.  A float64 MFMA occupies the SIMD's vector ALU for its whole 64 cycles — it runs on the float64 vector
// lanes — so EVERY vector instruction of any wavefront on that SIMD (integer address arithmetic, v_cndmask, cross-lane
// moves, the IEEE division of z / lengthscale: ~200 per tile in the forms above) adds to the matrix time instead of
// hiding under it.  This form keeps the row loop's vector work to the envelope products themselves (~25 per tile):
//  - no LDS staging of the row features and no workgroup barrier in the loop: a wavefront loads its A fragments itself
//    (scalar base + 32-bit vector offset, one vector add per tile), one tile ahead, two register sets alternating;
//  - z / lengthscale of the workgroup's rows, the per-wavefront row factors exp(-s (a - bmax)) / exp(-s (bmin - a))
//    (tile-permuted so a lane's four rows are one 32-byte read) are LDS tables filled once; which tiles lie off the band
//    is a 64-bit mask in scalar registers; the column factors are registers as before;
//  - results leave through the per-wavefront LDS transposition tile as whole row pieces (4 rows x 256 bytes per store
//    instruction: 16 x 64-byte pieces straight from the accumulators measured 3.3 TB/s against 5.6), all LDS and
//    global offsets loop-invariant or advanced by one scalar-register stride.
// Whole tiles of the engine's strips only (n2 a multiple of the workgroup's columns, 16-byte alignment, row segments of
// at most CVL_MAXR): the launcher falls back to cov_mercer_mfma_kernel otherwise.  Entries are bit-identical to it.
#define CVL_MAXR 512                 // rows per workgroup the LDS tables hold
template <int MPAD, int ENV, int CT, int NWV>
__global__ void __launch_bounds__(64 * NWV, 2) cov_mercer_mfma_lean_kernel(const CovItem* __restrict__ items,
                                                                          const double* __restrict__ x2s, int n2s, int row_seg) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  typedef const char __attribute__((address_space(1))) * gcbytes;
  typedef char __attribute__((address_space(1))) * gbytes;
  constexpr int NF = 2 * MPAD, KS = NF / 4;
  constexpr int WCOLS = 16 * CT, BCOLS = WCOLS * NWV;
  constexpr int TS = WCOLS + 2;         // LDS row stride of the transposition tile (doubles): rows stay 16-byte aligned
  const CovItem it = items[blockIdx.y];
  const cov_gcptr x1 = (cov_gcptr)it.x1;
  const int n1 = it.n1;
  const cov_gcptr x2 = (cov_gcptr)((it.n2 >= 0) ? it.x2 : x2s);
  const int n2 = (it.n2 >= 0) ? it.n2 : n2s;
  const cov_gcptr gf1 = (cov_gcptr)it.f1, gf2 = (cov_gcptr)it.f2, th = (cov_gcptr)it.k.theta;
  const int64_t ld = it.ld;
  const bool F32O = it.f32out != 0;
  __shared__ double etab[GP_EXP_TAB];
  // z_i / lengthscale of this workgroup's rows (rows past the end repeat the last) and, per wavefront, the separable
  // envelope's row factor of every row — both TILE-PERMUTED: row 16 t + q sits at 16 t + 4 (q & 3) + (q >> 2), so the four
  // rows kq, kq + 4, kq + 8, kq + 12 a lane owns in the accumulator are contiguous
  __shared__ __attribute__((aligned(16))) double a_t[CVL_MAXR];
  __shared__ __attribute__((aligned(16))) double rf_t[NWV][CVL_MAXR];
  __shared__ __attribute__((aligned(16))) double tbuf[NWV][16 * TS];
  gp_exp_tab_init(etab);
  const double var = th[0], ls = th[1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  const int jw = blockIdx.x * BCOLS + wave * WCOLS;
  const int rbeg = blockIdx.z * row_seg, rend = min(n1, rbeg + row_seg);
  if ((int)(blockIdx.x * BCOLS) >= n2 || rbeg >= n1) return;
  const int nrow = rend - rbeg, ntile = (nrow + 15) / 16;
  for (int r = tid; r < ntile * 16; r += 64 * NWV) {
    const int q = r & 15;
    a_t[(r & ~15) + 4 * (q & 3) + (q >> 2)] = x1[min(rbeg + r, n1 - 1)] / ls;
  }
  double bfr[CT][KS], bsc[CT], bb[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    const int j = jw + ct * 16 + lc;
#pragma unroll
    for (int s = 0; s < KS; s++) bfr[ct][s] = gf2[(size_t)(4 * s + kq) * n2 + j];
    bsc[ct] = x2[j] / ls;
    bb[ct] = __dmul_rn(bsc[ct], bsc[ct]);
  }
  double bmin_w, bmax_w;
  {
    double lo = bsc[0], hi = bsc[0];
#pragma unroll
    for (int ct = 1; ct < CT; ct++) { lo = fmin(lo, bsc[ct]); hi = fmax(hi, bsc[ct]); }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
    bmin_w = lo; bmax_w = hi;
  }
  constexpr double SENV = (ENV == 0) ? 1.0 : 2.23606797749979;
  const bool sep_ok = (SENV * (bmax_w - bmin_w) < 300.0);
  __syncthreads();                                                 // etab and a_t are ready (the only workgroup barrier)
  double cfp[CT], cfn[CT];                                         // var * exp(-s (bmax - b)),  var * exp(-s (b - bmin))
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    cfp[ct] = var * gp_exp_neg(-SENV * (bmax_w - bsc[ct]), etab);
    cfn[ct] = var * gp_exp_neg(-SENV * (bsc[ct] - bmin_w), etab);
  }
  double* rfw = rf_t[wave];
  for (int r = lane; r < ntile * 16; r += 64) {      // (a pure function of a: the permuted slots map one to one)
    const double a = a_t[r];
    double v = 0.0;
    if (a >= bmax_w) v = gp_exp_neg(-SENV * (a - bmax_w), etab);
    else if (a <= bmin_w) v = gp_exp_neg(-SENV * (bmin_w - a), etab);
    rfw[r] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();          // the table is the wavefront's own: its LDS operations complete in order
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double* tw = tbuf[wave];
  constexpr int PPR = WCOLS / 2, RPI = 64 / PPR;           // store phase: RPI rows x PPR column pairs per instruction
  const int srow = lane / PPR, scol = (lane % PPR) * 2;
  // A fragments: element (4 s + kq) * n1 + row of the feature table = uniform base of k-step s + one 32-bit byte offset
  const uint32_t frow_last = (uint32_t)(kq * n1 + n1 - 1) * 8u;
  uint32_t foff = min((uint32_t)(kq * n1 + rbeg + lc) * 8u, frow_last);
  gcbytes fbs[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) {
    const uint64_t b = (uint64_t)gf1 + (uint64_t)s * ((uint64_t)n1 * 32u);       // 4 feature rows per k-step
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    fbs[s] = (gcbytes)(((uint64_t)hi << 32) | lo);
  }
  // output: one 32-bit byte offset per lane (row rbeg + srow, column jw + scol); store q of a tile adds q * RPI rows
  // (a scalar stride folded into the scalar base), every tile 16 rows
  const uint32_t esz = F32O ? 4u : 8u;
  uint32_t ooff = (uint32_t)(((int64_t)(rbeg + srow) * ld + jw + scol) * esz);
  const uint32_t ostep = (uint32_t)(16 * ld * esz);
  gbytes obq[16 / RPI];
#pragma unroll
  for (int q = 0; q < 16 / RPI; q++) {
    const uint64_t b = (uint64_t)it.out + (uint64_t)q * (uint64_t)(RPI * ld * esz);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    obq[q] = (gbytes)(((uint64_t)hi << 32) | lo);
  }
  uint64_t amask = 0, bmask = 0;
  // one tile: the NEXT tile's fragments are requested first (into the other register set), then this tile's MFMAs,
  // envelope, transposition and stores.  Two register sets alternate (tiles are walked in pairs): no copies.
  auto do_tile = [&](const int t, const int tbit, const double (&afc)[KS], double (&afn)[KS]) {
    foff = min(foff + 128u, frow_last);                             // rows past the end repeat the last one (never stored)
#pragma unroll
    for (int s = 0; s < KS; s++) afn[s] = *(cov_gcptr)(fbs[s] + foff);
    d4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) acc[ct] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; s++)
#pragma unroll
      for (int ct = 0; ct < CT; ct++) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(afc[s], bfr[ct][s], acc[ct], 0, 0, 0);
    // element r of acc[ct]: row 16 t + kq + 4 r, column 16 ct + lc of this wavefront's strip
    const bool above = (amask >> tbit) & 1, below = (bmask >> tbit) & 1;     // scalar
    const cov_d2 q01 = *reinterpret_cast<const cov_d2*>((above || below ? rfw : a_t) + t * 16 + 4 * kq);
    const cov_d2 q23 = *reinterpret_cast<const cov_d2*>((above || below ? rfw : a_t) + t * 16 + 4 * kq + 2);
    const double rv[4] = {q01.x, q01.y, q23.x, q23.y};              // row factors (off the band) or scaled inputs (inside it)
    if (above || below) {
      cov_d2 a01, a23;
      if (ENV != 0) { a01 = *reinterpret_cast<const cov_d2*>(a_t + t * 16 + 4 * kq); a23 = *reinterpret_cast<const cov_d2*>(a_t + t * 16 + 4 * kq + 2); }
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
          double e = rv[r] * (above ? cfp[ct] : cfn[ct]);
          if (ENV != 0) {
            const double a = (r == 0) ? a01.x : (r == 1) ? a01.y : (r == 2) ? a23.x : a23.y;
            const double rr = fabs(a - bsc[ct]);
            e *= 1.0 + SENV * rr + (5.0 / 3.0) * (rr * rr);
          }
          tw[(kq + 4 * r) * TS + ct * 16 + lc] = e * acc[ct][r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double a = rv[r], aa = __dmul_rn(a, a), m2a = -2.0 * a;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
          // (-2 (a b) + a a) + b b, every operation rounded on its own (scaling by -2 is exact)
          const double rr = gp_sqrt_pos(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m2a, bsc[ct]), aa), bb[ct]), 1e-12));
          double env;
          if (ENV == 0) env = gp_exp_neg(-rr, etab);
          else { const double s5 = 2.23606797749979; env = (1.0 + s5 * rr + (5.0 / 3.0) * (rr * rr)) * gp_exp_neg(-s5 * rr, etab); }
          tw[(kq + 4 * r) * TS + ct * 16 + lc] = var * env * acc[ct][r];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();          // LDS operations of one wavefront complete in order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 16 / RPI; q++) {
      const int lr = RPI * q + srow;
      const cov_d2 v = *reinterpret_cast<const cov_d2*>(tw + lr * TS + scol);
      if (t * 16 + lr < nrow) {               // (only the last tile of a ragged row range masks lanes)
        if (F32O) __builtin_nontemporal_store(cov_f2{(float)v.x, (float)v.y}, (cov_gfptr2)(obq[q] + ooff));
        else __builtin_nontemporal_store(v, (cov_gptr2)(obq[q] + ooff));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();          // the tile is rewritten by the next row tile
    ooff += ostep;
  };
  double afA[KS], afB[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) afA[s] = *(cov_gcptr)(fbs[s] + foff);
  for (int tg = 0; tg < ntile; tg += 64) {
    // which of the next 64 tiles lie wholly on one side of this wavefront's columns (lane = tile)
    {
      const int t = min(tg + lane, ntile - 1);
      double lo = a_t[t * 16], hi = lo;
#pragma unroll
      for (int q = 1; q < 16; q++) { const double v = a_t[t * 16 + q]; lo = fmin(lo, v); hi = fmax(hi, v); }
      amask = __ballot(sep_ok && (lo - bmax_w >= CVM_SEP));
      bmask = __ballot(sep_ok && (bmin_w - hi >= CVM_SEP));
    }
    const int tend = min(ntile, tg + 64);
    int t = tg;
    for (; t + 1 < tend; t += 2) {
      do_tile(t, t - tg, afA, afB);
      do_tile(t + 1, t + 1 - tg, afB, afA);
    }
    if (t < tend) {                          // odd tile count: one more, then put the prefetched set back in place
      do_tile(t, t - tg, afA, afB);
#pragma unroll
      for (int s = 0; s < KS; s++) afA[s] = afB[s];
    }
  }
}

// ---- grouped launches: all matrices of one kernel family (same type, same padded partial count) in one launch ----
void cov_item_fill(CovItem* it, DevKern k, const double* x1, int n1, const double* x2, int n2, double* out, int64_t ld,
                   int accumulate, double diag_add, double* feat_ws, int f32out) {
  it->f32out = f32out; it->pad_ = 0;
  if (!x2 && n2 >= 0) { x2 = x1; n2 = n1; }      // n2 < 0: x2 / n2 come from the launch (shared frames)
  it->k = k; it->x1 = x1; it->n1 = n1; it->x2 = x2; it->n2 = n2; it->out = out; it->ld = ld;
  it->accumulate = accumulate; it->diag_add = diag_add;
  it->vec_ok = ((ld % 2) == 0) && ((((uintptr_t)out) & 15) == 0);
  it->f1 = it->f2 = nullptr;
  if (gp_kern_is_mercer(k.type) && feat_ws) {
    const int mp = sm_mpad(k.m);
    it->f1 = feat_ws;
    it->f2 = (n2 >= 0 && x2 == x1) ? feat_ws : feat_ws + gp_align_up((size_t)2 * mp * n1, 32);
  }
}

gp_status launch_sm_features_items(gp_handle h, const FeatItem* d_items, int count, int max_n, int mpad,
                                   const double* x_shared, int n_shared) {
  if (count <= 0 || max_n <= 0) return GP_OK;
  hipLaunchKernelGGL(sm_features_kernel, dim3((max_n + 255) / 256, mpad, count), dim3(256), 0, h->stream, DevKern{0, 0, nullptr},
                     x_shared, n_shared, (double*)nullptr, mpad, d_items);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// type / m describe the whole group (the feature tables of a Mercer group must already be built)
gp_status launch_kernel_build_items(gp_handle h, int type, int m, const CovItem* d_items, int count, int max_n1,
                                    int max_n2, const double* x2_shared, int n2_shared, int engine_strips) {
  const bool lean_items = engine_strips != 0;
  if (count <= 0 || max_n1 <= 0 || max_n2 <= 0) return GP_OK;
  const bool big = (int64_t)max_n1 * max_n2 >= (1 << 20);
  GpTimerScope ts(h, !big ? GP_TIMER_SMALL_GEMM : (gp_kern_is_mercer(type) ? GP_TIMER_KUF_BUILD_SM : GP_TIMER_KUF_BUILD));
  DevKern k0{type, m, nullptr};
  if (gp_kern_is_mercer(type)) {
    if (m < 1 || m > 32) return gp_fail(h, GP_ERR_UNSUPPORTED, "num_partials must be in [1, 32]");
    if (x2_shared && big) {
      // Kuf strips (shared frames, nothing accumulated, no diagonal): matrix-core form
      // rows split only while the grid is smaller than the 512 workgroups the device holds (row segments are multiples of 32)
      const int colblk = (max_n2 + 255) / 256;
      int nseg = 1;
      while (nseg < 8 && (int64_t)colblk * count * nseg < 512 && (max_n1 + nseg * 2 - 1) / (nseg * 2) >= 2 * CVM_ROWS) nseg *= 2;
      constexpr int FREE_NWV = 4;
      // the lean form takes whole tiles of the engine's strips only, rows in segments its LDS tables hold
      // (m <= 8 partials: the build is store-bound either way and the staged form's four wavefronts per SIMD are 6 % ahead)
      // (float32 strips, engine_strips == 2: half the store bytes, so from 5 partials on the lean form wins there too — cfg3
      // 0.18 -> 0.12 ms per launch, same-box)
      const int min_mpad = (engine_strips == 2 ? 8 : 12);
      const bool lean_ok = lean_items && (n2_shared % (16 * CVM_CT * FREE_NWV) == 0) && sm_mpad(m) >= min_mpad;
      if (lean_ok) while ((max_n1 + nseg - 1) / nseg > CVL_MAXR) nseg *= 2;
      const int row_seg = ((max_n1 + nseg - 1) / nseg + CVM_ROWS - 1) / CVM_ROWS * CVM_ROWS;
      dim3 gm(colblk, count, (max_n1 + row_seg - 1) / row_seg);
      const dim3 gfree((max_n2 + 16 * CVM_CT * FREE_NWV - 1) / (16 * CVM_CT * FREE_NWV), count, gm.z);
#define COV_MFMA(MP)                                                                                                   \
      do {                                                                                                             \
        if (lean_ok && type == GP_KERN_MERCER_MATERN12SM)                                                              \
          hipLaunchKernelGGL((cov_mercer_mfma_lean_kernel<MP, 0, CVM_CT, FREE_NWV>), gfree, dim3(64 * FREE_NWV), 0, h->stream, d_items, x2_shared, n2_shared, row_seg); \
        else if (lean_ok)                                                                                              \
          hipLaunchKernelGGL((cov_mercer_mfma_lean_kernel<MP, 2, CVM_CT, FREE_NWV>), gfree, dim3(64 * FREE_NWV), 0, h->stream, d_items, x2_shared, n2_shared, row_seg); \
        else if (type == GP_KERN_MERCER_MATERN12SM)                                                                         \
          hipLaunchKernelGGL((cov_mercer_mfma_kernel<MP, 0, CVM_CT>), gm, dim3(1024 / CVM_CT), 0, h->stream, d_items, x2_shared, n2_shared, row_seg); \
        else                                                                                                           \
          hipLaunchKernelGGL((cov_mercer_mfma_kernel<MP, 2, CVM_CT>), gm, dim3(1024 / CVM_CT), 0, h->stream, d_items, x2_shared, n2_shared, row_seg); \
      } while (0)
      switch (sm_mpad(m)) {
        case 4: COV_MFMA(4); break;
        case 8: COV_MFMA(8); break;
        case 12: COV_MFMA(12); break;
        case 16: COV_MFMA(16); break;
        case 20: COV_MFMA(20); break;
        case 24: COV_MFMA(24); break;
        case 28: COV_MFMA(28); break;
        default: COV_MFMA(32); break;
      }
#undef COV_MFMA
      GP_HIP_CHECK(h, hipGetLastError());
      return GP_OK;
    }
    dim3 grid((max_n2 + COV_THREADS * 2 - 1) / (COV_THREADS * 2), (max_n1 + COV_ROWS - 1) / COV_ROWS, count);
#define COV_MERCER_ITEMS(MP) launch_mercer<MP>(h, grid, k0, nullptr, 0, x2_shared, n2_shared, nullptr, 0, 0, 0.0, nullptr, nullptr, 0, d_items)
    switch (sm_mpad(m)) {
      case 4: COV_MERCER_ITEMS(4); break;
      case 8: COV_MERCER_ITEMS(8); break;
      case 12: COV_MERCER_ITEMS(12); break;
      case 16: COV_MERCER_ITEMS(16); break;
      case 20: COV_MERCER_ITEMS(20); break;
      case 24: COV_MERCER_ITEMS(24); break;
      case 28: COV_MERCER_ITEMS(28); break;
      default: COV_MERCER_ITEMS(32); break;
    }
#undef COV_MERCER_ITEMS
  } else if (gp_kern_is_broadcast(type)) {
    dim3 grid((max_n2 + COV_THREADS - 1) / COV_THREADS, (max_n1 + COV_ROWS - 1) / COV_ROWS, count);
    hipLaunchKernelGGL((cov_build_kernel<2, 1, 1>), grid, dim3(COV_THREADS), 0, h->stream, k0, (const double*)nullptr, 0,
                       x2_shared, n2_shared, (double*)nullptr, (int64_t)0, 0, 0.0, (const double*)nullptr,
                       (const double*)nullptr, 0, d_items, COV_ROWS, 0);
  } else {
    dim3 grid((max_n2 + COV_THREADS * 2 - 1) / (COV_THREADS * 2), (max_n1 + COV_ROWS - 1) / COV_ROWS, count);
#define COV_STAT_ITEMS(T) hipLaunchKernelGGL((cov_build_kernel<0, 2, 1, T>), grid, dim3(COV_THREADS), 0, h->stream, k0, \
                                            (const double*)nullptr, 0, x2_shared, n2_shared, (double*)nullptr,             \
                                            (int64_t)0, 0, 0.0, (const double*)nullptr, (const double*)nullptr, 0, d_items, COV_ROWS, 0)
    switch (type) {
      case GP_KERN_MATERN12: COV_STAT_ITEMS(GP_KERN_MATERN12); break;
      case GP_KERN_MATERN32: COV_STAT_ITEMS(GP_KERN_MATERN32); break;
      case GP_KERN_MATERN52: COV_STAT_ITEMS(GP_KERN_MATERN52); break;
      default: COV_STAT_ITEMS(GP_KERN_RBF); break;
    }
#undef COV_STAT_ITEMS
  }
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// (re)build the spectral-mixture feature tables of kernel k for (x1, x2) without building a covariance
gp_status launch_sm_features(gp_handle h, DevKern k, const double* x1, int n1, const double* x2, int n2, double* feat_ws) {
  if (!gp_kern_is_mercer(k.type)) return GP_OK;
  const int mp = sm_mpad(k.m);
  double* f1 = feat_ws;
  double* f2 = feat_ws + gp_align_up((size_t)2 * mp * n1, 32);
  hipLaunchKernelGGL(sm_features_kernel, dim3((n1 + 255) / 256, mp), dim3(256), 0, h->stream, k, x1, n1, f1, mp, (const FeatItem*)nullptr);
  if (x2 && x2 != x1)
    hipLaunchKernelGGL(sm_features_kernel, dim3((n2 + 255) / 256, mp), dim3(256), 0, h->stream, k, x2, n2, f2, mp, (const FeatItem*)nullptr);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

gp_status launch_kernel_diag(gp_handle h, DevKern k, int n, double* out, int accumulate) {
  if (n <= 0) return GP_OK;
  hipLaunchKernelGGL(cov_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, k, n, out, accumulate);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
