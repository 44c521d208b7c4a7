// chol.hip — dense fp64 Cholesky and triangular inverse on Kuu (gfx950).
//
// Replaces tf.cholesky / tf.matrix_triangular_solve on the M x M inducing covariance
// (GPflow conditional from gpitch/pdgp.py:147; gpitch/sgpr_ss.py:44,48,51,53,89-94).
//
// One workgroup (8 wavefronts, two per SIMD) per matrix, many matrices per launch (one per latent GP).
// Right-looking blocked factorisation, panel width 32:
//   * the 32 x 32 diagonal block is factorised (and inverted) by ONE wavefront entirely in registers — lane i
//     holds row i, pivots/columns are broadcast with v_readlane (no LDS round trips, no barriers);
//   * the panel below is X = A_panel L_kk^-T on the matrix cores against the inverse held in LDS;
//   * the trailing SYRK update runs on the matrix cores (v_mfma_f64_16x16x4_f64) in 32 x 32 macro tiles (four
//     independent accumulators) drawn from an LDS work counter; the wavefront that draws the next diagonal
//     block factorises it immediately (look-ahead), hiding the serial factorisation behind the update.
// The inverse W = L^-1 inverts the diagonal blocks in registers the same way, then each wavefront
// walks one block column with MFMA products; a 16x16 f64 accumulator register r is exactly the
// B-fragment of k-step r, so the chained product -W_ii * (sum_k L_ik W_kj) needs no data movement.
#include "common.h"
#include <string.h>
#include <atomic>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CH_NB 32
#define CH_THREADS 512
#define CH_WAVES (CH_THREADS / 64)

#ifdef CH_STAMPS
// diagnostic build (never shipped; tools/chol_stamps.py): s_memtime stamps of matrix 0's workgroup, per 32-column panel
// [panel start, panel product done, trailing update done, look-ahead diagonal block start, end]
__device__ unsigned long long ch_stamps[5 * 64];
extern "C" int gp_debug_chol_stamps(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(ch_stamps), sizeof(ch_stamps)) == hipSuccess ? 0 : -1;
}
#define CH_STAMP(k, i) do { if (b == 0 && (k) < 64) ch_stamps[5 * (k) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CH_STAMP(k, i) do { } while (0)
#endif

// broadcast lane `src` (a compile-time constant after unrolling) through SGPRs: v_readlane_b32 x2
__device__ __forceinline__ double lane_bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// sqrt(x) and 1/sqrt(x) of a pivot from one v_rsq_f64 and fused corrections (both within 1.5 ulp): the pivot
// chain of the in-register factorisation is serial, so the library's sqrt + division (each with scaling and class
// handling) is what bounds it.  Valid for finite x > 0 (no intermediate leaves the double range); the caller
// rejects everything else as a bad pivot.
__device__ __forceinline__ void pivot_sqrt_recip(double x, double& s, double& rinv) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  double r = h + h;
  r = fma(fma(-g, r, 1.0), r, r);
  s = g; rinv = r;
}

// Factor the 32 x 32 diagonal block at (k0, k0) entirely in the registers of ONE wavefront (lane i holds row i;
// pivots / columns are broadcast with v_readlane), write L_kk to global and its inverse to Dinv (LDS).
// PT: the matrix pointer type.  Generic `double*` for the LDS-resident form (chol_inverse_lds_kernel hands LDS
// addresses); address-space-1 for matrices in device memory — a pointer fetched from a device array is generic to the
// compiler and its accesses become FLAT, which count on lgkmcnt too: every LDS wait of the factorisation then also waits
// for the matrix loads / stores in flight.
typedef double __attribute__((address_space(1))) * ch_gptr;
typedef const double __attribute__((address_space(1))) * ch_gcptr;
template <typename PT>
__device__ __forceinline__ void chol_diag_block(PT A, int64_t ld, int M, int k0, int lane,
                                                double (*Dinv)[CH_NB + 1], int* __restrict__ status, int b, int pbase) {
  const int nb = min(CH_NB, M - k0);
  double row[CH_NB];
  double rinv[CH_NB];   // 1 / L_jj, wave-uniform (lives in SGPRs)
  const int i = lane & 31;
  // unconditional loads from clamped (always valid) addresses, then select: no divergent branches
  const auto arow = A + (int64_t)(k0 + min(i, nb - 1)) * ld + k0;
#pragma unroll
  for (int c = 0; c < CH_NB; c++) {
    const double v = arow[min(c, nb - 1)];
    row[c] = (i < nb && c <= i) ? v : (i == c ? 1.0 : 0.0);
  }
  int bad = -1;        // first non-positive / non-finite pivot (wave-uniform), reported once after the loop
#pragma unroll
  for (int j = 0; j < CH_NB; j++) {
    double djj = lane_bcast(row[j], j);
    const bool ok = (djj > 0.0) && (djj <= 1.7976931348623157e308);
    bad = (!ok && bad < 0 && j < nb) ? j : bad;
    djj = ok ? djj : 1.0;                       // keep going finite
    double s, ri;
    pivot_sqrt_recip(djj, s, ri);
    rinv[j] = lane_bcast(ri, 0);
    double lij = (i > j) ? row[j] * ri : (i == j ? s : 0.0);
    row[j] = lij;
#pragma unroll
    for (int c = j + 1; c < CH_NB; c++) {
      double lcj = lane_bcast(lij, c);
      row[c] = fma(-lij, lcj, row[c]);  // meaningful for i >= c only
    }
  }
  if (bad >= 0 && lane == 0) {
    if (atomicCAS(&status[0], 0, 1) == 0) { status[1] = pbase + k0 + bad; status[2] = b; }
  }
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < CH_NB; c++)
      if (i < nb && c <= i) A[(int64_t)(k0 + i) * ld + k0 + c] = row[c];
  }
  // inverse of the diagonal factor, column c in lane c: x[r] = (L_kk^-1)[r][c], by forward substitution taken
  // column by column (x[t] final -> 31 - t independent updates).  L_rt is lane r of row[t], the very value the
  // factor loop broadcast: laundering row[] keeps the compiler from holding all 496 of them live in SGPRs
  // (it spilled them lane by lane into VGPRs) instead of simply reading the lane again.
#pragma unroll
  for (int c = 0; c < CH_NB; c++) asm volatile("" : "+v"(row[c]));
  double x[CH_NB];
#pragma unroll
  for (int r = 0; r < CH_NB; r++) x[r] = (i == r) ? 1.0 : 0.0;
#pragma unroll
  for (int t = 0; t < CH_NB; t++) {
    x[t] *= rinv[t];
#pragma unroll
    for (int r = t + 1; r < CH_NB; r++) x[r] = fma(-lane_bcast(row[t], r), x[t], x[r]);
  }
  // pin x[] here: otherwise the substitution is sunk into the lane-conditional stores below while its
  // (convergent) lane broadcasts stay outside, all 496 of them live at once and spilled
#pragma unroll
  for (int r = 0; r < CH_NB; r++) asm volatile("" : "+v"(x[r]));
  if (lane < 32) {
#pragma unroll
    for (int r = 0; r < CH_NB; r++) Dinv[r][i] = x[r];
  }
}

// Cholesky of one matrix by one workgroup (body shared by chol_kernel and chol_inverse_kernel)
template <typename PT>
__device__ __forceinline__ void chol_body(PT A, const int M, const int64_t ld, int* __restrict__ status,
                                          const int b, const int panel_rows_cap, const int pivot_base) {
  extern __shared__ __attribute__((aligned(16))) double chol_smem[];
  // [ D: 2 x 32 x 33 (inverse of the current / next diagonal factor) | P: panel X, (M-32) x 33 when it fits ]
  double (*D)[CH_NB][CH_NB + 1] = reinterpret_cast<double (*)[CH_NB][CH_NB + 1]>(chol_smem);
  double* P = chol_smem + 2 * CH_NB * (CH_NB + 1);
  __shared__ int tile_counter;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kq = lane >> 4, lc = lane & 15;
  const bool p_lds = (panel_rows_cap >= M - CH_NB);   // operands of the trailing update come from LDS

  if (wave == 0) chol_diag_block(A, ld, M, 0, lane, D[0], status, b, pivot_base);
  __syncthreads();

  int pb = 0;
  for (int k0 = 0; k0 < M; k0 += CH_NB, pb ^= 1) {
    const int nb = min(CH_NB, M - k0);
    const int r0 = k0 + nb;  // first trailing row
    const int R = M - r0;
    if (R <= 0) break;
    if (tid == 0) CH_STAMP(k0 / CH_NB, 0);
    // ---- panel: X = A_panel * L_kk^-T on the matrix cores, 16 rows per wavefront step ----------
    for (int rg = wave; rg * 16 < R; rg += CH_WAVES) {
      const int ra = r0 + rg * 16 + lc;
      double af[8];
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        const int kc = kk * 4 + kq;
        af[kk] = (ra < M && kc < nb) ? A[(int64_t)ra * ld + k0 + kc] : 0.0;
      }
      d4 o[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {  // B[k][j] = (L_kk^-1)[j][k]; two independent accumulators
        o[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], D[pb][lc][kk * 4 + kq], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], D[pb][16 + lc][kk * 4 + kq], o[1], 0, 0, 0);
      }
#pragma unroll
      for (int tj = 0; tj < 2; tj++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int rw = r0 + rg * 16 + kq + 4 * r, cw = tj * 16 + lc;
          if (rw < M && cw < nb) A[(int64_t)rw * ld + k0 + cw] = o[tj][r];
          if (p_lds && rw < M) P[(rw - r0) * (CH_NB + 1) + cw] = (cw < nb) ? o[tj][r] : 0.0;
        }
    }
    if (tid == 0) tile_counter = 0;
    __threadfence_block();
    __syncthreads();
    if (tid == 0) CH_STAMP(k0 / CH_NB, 1);
    // ---- trailing update A22 -= X X^T (lower triangle), 32 x 32 macro tiles handed out dynamically.  Macro tile
    //      0 is the NEXT diagonal block: the wavefront that draws it factorises that block straight away
    //      (look-ahead), overlapping the serial 32-column factorisation with the other waves' updates.
    const int nmt = (R + 31) >> 5;
    const int total = nmt * (nmt + 1) / 2;
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&tile_counter, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= total) break;
      int mi = (int)((__dsqrt_rn(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((mi + 1) * (mi + 2) / 2 <= t) mi++;
      while (mi * (mi + 1) / 2 > t) mi--;
      const int mj = t - mi * (mi + 1) / 2;
      const int rbase = r0 + mi * 32, cbase = r0 + mj * 32;
      double af0[8], af1[8], bf0[8], bf1[8];
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        const int kc = kk * 4 + kq;
        const bool kin = kc < nb;
        const int ra0 = rbase + lc, ra1 = rbase + 16 + lc, rb0 = cbase + lc, rb1 = cbase + 16 + lc;
        if (p_lds) {   // rows >= M were never written: guard; columns >= nb hold zeros
          af0[kk] = (ra0 < M) ? P[(ra0 - r0) * (CH_NB + 1) + kc] : 0.0;
          af1[kk] = (ra1 < M) ? P[(ra1 - r0) * (CH_NB + 1) + kc] : 0.0;
          bf0[kk] = (rb0 < M) ? P[(rb0 - r0) * (CH_NB + 1) + kc] : 0.0;
          bf1[kk] = (rb1 < M) ? P[(rb1 - r0) * (CH_NB + 1) + kc] : 0.0;
        } else {
          af0[kk] = (kin && ra0 < M) ? A[(int64_t)ra0 * ld + k0 + kc] : 0.0;
          af1[kk] = (kin && ra1 < M) ? A[(int64_t)ra1 * ld + k0 + kc] : 0.0;
          bf0[kk] = (kin && rb0 < M) ? A[(int64_t)rb0 * ld + k0 + kc] : 0.0;
          bf1[kk] = (kin && rb1 < M) ? A[(int64_t)rb1 * ld + k0 + kc] : 0.0;
        }
      }
      // C tile: issue the loads now, consume them after the MFMAs
      double cold[4][4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rw0 = rbase + kq + 4 * r, rw1 = rw0 + 16, cw0 = cbase + lc, cw1 = cw0 + 16;
        cold[0][r] = (rw0 < M && cw0 <= rw0) ? A[(int64_t)rw0 * ld + cw0] : 0.0;
        cold[1][r] = (rw0 < M && cw1 <= rw0) ? A[(int64_t)rw0 * ld + cw1] : 0.0;
        cold[2][r] = (rw1 < M && cw0 <= rw1) ? A[(int64_t)rw1 * ld + cw0] : 0.0;
        cold[3][r] = (rw1 < M && cw1 <= rw1) ? A[(int64_t)rw1 * ld + cw1] : 0.0;
      }
      d4 c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[kk], bf0[kk], c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[kk], bf1[kk], c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[kk], bf0[kk], c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[kk], bf1[kk], c11, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rw0 = rbase + kq + 4 * r, rw1 = rw0 + 16, cw0 = cbase + lc, cw1 = cw0 + 16;
        if (rw0 < M && cw0 <= rw0) A[(int64_t)rw0 * ld + cw0] = cold[0][r] - c00[r];
        if (rw0 < M && cw1 <= rw0) A[(int64_t)rw0 * ld + cw1] = cold[1][r] - c01[r];
        if (rw1 < M && cw0 <= rw1) A[(int64_t)rw1 * ld + cw0] = cold[2][r] - c10[r];
        if (rw1 < M && cw1 <= rw1) A[(int64_t)rw1 * ld + cw1] = cold[3][r] - c11[r];
      }
      if (t == 0) {
        __threadfence_block();  // this wave's own updates of the block it is about to read back
        if (lane == 0) CH_STAMP(k0 / CH_NB, 3);
        chol_diag_block(A, ld, M, r0, lane, D[pb ^ 1], status, b, pivot_base);
        if (lane == 0) CH_STAMP(k0 / CH_NB, 4);
      }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) CH_STAMP(k0 / CH_NB, 2);
  }
  // zero the strictly-upper triangle so L can be used as a dense operand
  for (int64_t idx = tid; idx < (int64_t)M * M; idx += CH_THREADS) {
    int i = (int)(idx / M), j = (int)(idx % M);
    if (j > i) A[(int64_t)i * ld + j] = 0.0;
  }
}

__global__ void __launch_bounds__(CH_THREADS) chol_kernel(double* const* __restrict__ mats, const int* __restrict__ Ms,
                                                          const int* __restrict__ lds_, int* __restrict__ status,
                                                          double* single_mat, int single_M, int single_ld, int panel_rows_cap,
                                                          int pivot_base) {
  const int b = blockIdx.x;
  chol_body((ch_gptr)(mats ? mats[b] : single_mat), mats ? Ms[b] : single_M, mats ? (int64_t)lds_[b] : (int64_t)single_ld, status, b,
            panel_rows_cap, pivot_base);
}

// W = L^-1 (lower) of one matrix by one workgroup (body shared by tri_inverse_kernel and chol_inverse_kernel).
template <typename CPT, typename PT>
__device__ __forceinline__ void tri_inverse_body(CPT L, PT W, const int M, const int64_t ld) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int T = (M + CH_NB - 1) / CH_NB;

  // zero-fill W (upper part and everything not yet written)
  for (int64_t idx = tid; idx < (int64_t)M * M; idx += CH_THREADS) {
    int i = (int)(idx / M), j = (int)(idx % M);
    W[(int64_t)i * ld + j] = 0.0;
  }
  __threadfence_block();
  __syncthreads();

  // ---- step A: invert the diagonal blocks in registers -----------------------------------------
  for (int d = wave; d < T; d += CH_WAVES) {
    const int d0 = d * CH_NB;
    const int nb = min(CH_NB, M - d0);
    const int i = lane & 31;
    double lrow[CH_NB];  // lane i: row i of the diagonal block of L (identity-padded)
    const auto lsrc = L + (int64_t)(d0 + min(i, nb - 1)) * ld + d0;   // clamped: loads are unconditional
#pragma unroll
    for (int c = 0; c < CH_NB; c++) {
      const double v = lsrc[min(c, nb - 1)];
      lrow[c] = (i < nb && c <= i) ? v : (i == c ? 1.0 : 0.0);
    }
    // one division per lane up front (lane i: 1 / L_ii), so the serial substitution below only multiplies
    const double rec = 1.0 / ((i < nb) ? lsrc[min(i, nb - 1)] : 1.0);
    double x[CH_NB];  // lane c: column c of the inverse, x[t] = X[t][c]; substitution column by column
#pragma unroll
    for (int r = 0; r < CH_NB; r++) x[r] = (i == r) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < CH_NB; t++) {
      x[t] *= lane_bcast(rec, t);
#pragma unroll
      for (int r = t + 1; r < CH_NB; r++) x[r] = fma(-lane_bcast(lrow[t], r), x[t], x[r]);
    }
#pragma unroll
    for (int r = 0; r < CH_NB; r++) asm volatile("" : "+v"(x[r]));   // as in chol_diag_block: no sinking into the stores
    if (lane < 32 && i < nb) {
#pragma unroll
      for (int r = 0; r < CH_NB; r++)
        if (r < nb && r >= i) W[(int64_t)(d0 + r) * ld + d0 + i] = x[r];
    }
  }
  __threadfence_block();
  __syncthreads();

  // ---- step B: block columns; W_ij = -W_ii * sum_{k=j}^{i-1} L_ik W_kj ---------------------------
  const int kq = lane >> 4, lc = lane & 15;
  for (int j = wave; j < T; j += CH_WAVES) {
    const int c0 = j * CH_NB;
    for (int i = j + 1; i < T; i++) {
      const int i0 = i * CH_NB;
      d4 S[2][2];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++) S[a][c] = d4{0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < i; k++) {
        const int kb = k * CH_NB;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          const int kcol = kb + kk * 4 + kq;  // < M always (k < i <= T-1 so block k is full)
          double a0, a1, b0, b1;
          {
            int ra = i0 + lc, rb = i0 + 16 + lc;
            a0 = (ra < M) ? L[(int64_t)ra * ld + kcol] : 0.0;
            a1 = (rb < M) ? L[(int64_t)rb * ld + kcol] : 0.0;
            b0 = W[(int64_t)kcol * ld + c0 + lc];
            b1 = W[(int64_t)kcol * ld + c0 + 16 + lc];
          }
          S[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, S[0][0], 0, 0, 0);
          S[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, S[0][1], 0, 0, 0);
          S[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, S[1][0], 0, 0, 0);
          S[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, S[1][1], 0, 0, 0);
        }
      }
      // out(ta, tc) = - sum_{tk, kk} Wii(ta, tk)[., 4kk + kq] * S[tk][tc].reg[kk]
      d4 O[2][2];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++) O[a][c] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int tk = 0; tk < 2; tk++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const int kcol = i0 + tk * 16 + kk * 4 + kq;
          int ra = i0 + lc, rb = i0 + 16 + lc;
          double a0 = (ra < M && kcol < M) ? W[(int64_t)ra * ld + kcol] : 0.0;
          double a1 = (rb < M && kcol < M) ? W[(int64_t)rb * ld + kcol] : 0.0;
          O[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, S[tk][0][kk], O[0][0], 0, 0, 0);
          O[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, S[tk][1][kk], O[0][1], 0, 0, 0);
          O[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, S[tk][0][kk], O[1][0], 0, 0, 0);
          O[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, S[tk][1][kk], O[1][1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            int rw = i0 + a * 16 + kq + 4 * r, cw = c0 + c * 16 + lc;
            if (rw < M) W[(int64_t)rw * ld + cw] = -O[a][c][r];
          }
      __threadfence_block();  // this wave re-reads the block it just wrote as the next B operand
    }
  }
}

// dynamic LDS: the two 32 x 33 inverse blocks plus, when it fits in 150 KiB, the whole panel below the first
// diagonal block ((maxM - 32) rows x 33 doubles).  *cap = rows the panel buffer can hold (0 = operands from global).
__global__ void __launch_bounds__(CH_THREADS) tri_inverse_kernel(const double* const* __restrict__ Ls,
                                                                 double* const* __restrict__ Ws,
                                                                 const int* __restrict__ Ms,
                                                                 const int* __restrict__ lds_, const double* single_L,
                                                                 double* single_W, int single_M, int single_ld) {
  const int b = blockIdx.x;
  tri_inverse_body((ch_gcptr)(Ls ? Ls[b] : single_L), (ch_gptr)(Ws ? Ws[b] : single_W), Ls ? Ms[b] : single_M,
                   Ls ? (int64_t)lds_[b] : (int64_t)single_ld);
}

// Factor and invert in ONE launch: the workgroup of a matrix stays resident from the first pivot to the last block of
// W.  Used when device-filling kernels (the Kuf strip builds) run beside the factorisation: a 512-thread,
// 256-VGPR workgroup needs an empty CU, which a second launch would not get until those kernels have drained.
__global__ void __launch_bounds__(CH_THREADS) chol_inverse_kernel(double* const* __restrict__ mats,
                                                                  double* const* __restrict__ Ws,
                                                                  const int* __restrict__ Ms, const int* __restrict__ lds_,
                                                                  int* __restrict__ status, int panel_rows_cap,
                                                                  double* single_mat, double* single_W, int single_M,
                                                                  int single_ld) {
  const int b = blockIdx.x;
  double* A = mats ? mats[b] : single_mat;
  double* W = mats ? Ws[b] : single_W;
  const int M = mats ? Ms[b] : single_M;
  const int64_t ld = mats ? (int64_t)lds_[b] : (int64_t)single_ld;
  chol_body((ch_gptr)A, M, ld, status, b, panel_rows_cap, 0);
  __threadfence();
  __syncthreads();
  tri_inverse_body((ch_gcptr)A, (ch_gptr)W, M, ld);
}

// Matrices of at most 64 rows (the window-sized SGPR problems, small inducing sets): the whole factor + inverse runs on
// a copy in LDS.  The bodies above take generic pointers, so they are simply handed LDS addresses (flat accesses to
// the LDS aperture): every dependent step — pivot, panel, update, the substitution chain of the inverse — then waits
// on an LDS round trip instead of an L2 one.
#define CHS_M 64
#define CHS_LD (CHS_M + 1)
#define CHS_OFF (2 * CH_NB * (CH_NB + 1) + (CHS_M - CH_NB) * (CH_NB + 1))     // chol_body's own D | P region
#define CHS_DOUBLES (CHS_OFF + 2 * CHS_M * CHS_LD)
__global__ void __launch_bounds__(CH_THREADS) chol_inverse_lds_kernel(double* const* __restrict__ mats,
                                                                      double* const* __restrict__ Ws,
                                                                      const int* __restrict__ Ms, const int* __restrict__ lds_,
                                                                      int* __restrict__ status, double* single_mat,
                                                                      double* single_W, int single_M, int single_ld) {
  extern __shared__ __attribute__((aligned(16))) double chol_smem[];
  const int b = blockIdx.x;
  double* A = mats ? mats[b] : single_mat;
  double* W = mats ? Ws[b] : single_W;
  const int M = mats ? Ms[b] : single_M;
  const int64_t ld = mats ? (int64_t)lds_[b] : (int64_t)single_ld;
  double* Al = chol_smem + CHS_OFF;
  double* Wl = Al + CHS_M * CHS_LD;
  for (int idx = threadIdx.x; idx < M * M; idx += CH_THREADS) {
    const int i = idx / M, j = idx % M;
    Al[i * CHS_LD + j] = A[(int64_t)i * ld + j];
  }
  __syncthreads();
  chol_body(Al, M, CHS_LD, status, b, CHS_M - CH_NB, 0);
  __threadfence_block();
  __syncthreads();
  tri_inverse_body((const double*)Al, Wl, M, CHS_LD);
  __threadfence_block();
  __syncthreads();
  for (int idx = threadIdx.x; idx < M * M; idx += CH_THREADS) {
    const int i = idx / M, j = idx % M;
    A[(int64_t)i * ld + j] = Al[i * CHS_LD + j];
    W[(int64_t)i * ld + j] = Wl[i * CHS_LD + j];
  }
}

static size_t chol_smem_bytes(int maxM, int* cap) {
  const size_t dbytes = (size_t)2 * CH_NB * (CH_NB + 1) * sizeof(double);
  const int rows = maxM > CH_NB ? maxM - CH_NB : 0;
  const size_t pbytes = (size_t)rows * (CH_NB + 1) * sizeof(double);
  if (dbytes + pbytes <= 150 * 1024) { *cap = rows; return dbytes + pbytes; }
  *cap = 0;
  return dbytes;
}
static gp_status chol_set_attr(gp_handle h) {
  // per-DEVICE attribute: one bit per device (several handles / threads may race here: a repeated call is harmless)
  static std::atomic<uint32_t> done{0};
  const uint32_t bit = 1u << (h->device & 31);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    done.fetch_or(bit, std::memory_order_release);
  }
  return GP_OK;
}

gp_status launch_cholesky_batched(gp_handle h, double* const* d_mats, const int* d_M, const int* d_ld, int batch,
                                  int maxM, int pivot_base) {
  (void)maxM;
  if (batch <= 0) return GP_OK;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  int cap = 0;
  size_t sh = chol_smem_bytes(maxM, &cap);
  GP_CHECK(chol_set_attr(h));
  hipLaunchKernelGGL(chol_kernel, dim3(batch), dim3(CH_THREADS), sh, h->stream, d_mats, d_M, d_ld, h->d_status,
                     (double*)nullptr, 0, 0, cap, pivot_base);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

static gp_status chol_inverse_set_attr(gp_handle h) {
  static std::atomic<uint32_t> done{0};
  const uint32_t bit = 1u << (h->device & 31);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)chol_inverse_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(CHS_DOUBLES * sizeof(double))));
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)chol_inverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    done.fetch_or(bit, std::memory_order_release);
  }
  return GP_OK;
}

gp_status launch_cholesky_inverse_batched(gp_handle h, double* const* d_mats, double* const* d_W, const int* d_M,
                                          const int* d_ld, int batch, int maxM) {
  if (batch <= 0) return GP_OK;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  int cap = 0;
  size_t sh = chol_smem_bytes(maxM, &cap);
  GP_CHECK(chol_inverse_set_attr(h));
  if (maxM <= CHS_M)
    hipLaunchKernelGGL(chol_inverse_lds_kernel, dim3(batch), dim3(CH_THREADS), CHS_DOUBLES * sizeof(double), h->stream, d_mats,
                       d_W, d_M, d_ld, h->d_status, (double*)nullptr, (double*)nullptr, 0, 0);
  else
    hipLaunchKernelGGL(chol_inverse_kernel, dim3(batch), dim3(CH_THREADS), sh, h->stream, d_mats, d_W, d_M, d_ld, h->d_status,
                       cap, (double*)nullptr, (double*)nullptr, 0, 0);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// one matrix: A -> L in place, W = L^-1, one launch (the window-sized SGPR problems are chains of dependent launches)
gp_status launch_cholesky_inverse_single(gp_handle h, double* A, double* W, int M, int64_t ld) {
  if (M <= 0) return GP_OK;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  int cap = 0;
  size_t sh = chol_smem_bytes(M, &cap);
  GP_CHECK(chol_inverse_set_attr(h));
  if (M <= CHS_M)
    hipLaunchKernelGGL(chol_inverse_lds_kernel, dim3(1), dim3(CH_THREADS), CHS_DOUBLES * sizeof(double), h->stream,
                       (double* const*)nullptr, (double* const*)nullptr, (const int*)nullptr, (const int*)nullptr,
                       h->d_status, A, W, M, (int)ld);
  else
    hipLaunchKernelGGL(chol_inverse_kernel, dim3(1), dim3(CH_THREADS), sh, h->stream, (double* const*)nullptr,
                       (double* const*)nullptr, (const int*)nullptr, (const int*)nullptr, h->d_status, cap, A, W, M, (int)ld);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

gp_status launch_cholesky_single(gp_handle h, double* A, int M, int64_t ld, int pivot_base) {
  if (M <= 0) return GP_OK;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  int cap = 0;
  size_t sh = chol_smem_bytes(M, &cap);
  GP_CHECK(chol_set_attr(h));
  hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(CH_THREADS), sh, h->stream, (double* const*)nullptr,
                     (const int*)nullptr, (const int*)nullptr, h->d_status, A, M, (int)ld, cap, pivot_base);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

gp_status launch_tri_inverse_single(gp_handle h, const double* L, double* Linv, int M, int64_t ld) {
  if (M <= 0) return GP_OK;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  hipLaunchKernelGGL(tri_inverse_kernel, dim3(1), dim3(CH_THREADS), 0, h->stream, (const double* const*)nullptr,
                     (double* const*)nullptr, (const int*)nullptr, (const int*)nullptr, L, Linv, M, (int)ld);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

gp_status launch_tri_inverse_batched(gp_handle h, const double* const* d_L, double* const* d_W, const int* d_M,
                                     const int* d_ld, int batch) {
  if (batch <= 0) return GP_OK;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  hipLaunchKernelGGL(tri_inverse_kernel, dim3(batch), dim3(CH_THREADS), 0, h->stream, d_L, d_W, d_M, d_ld,
                     (const double*)nullptr, (double*)nullptr, 0, 0);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// One LARGE matrix (the exact-GP N x N covariance of SGPRSS.predict_s, N = 2001 per window: sgpr_ss.py:88-94):
// the one-workgroup kernels above would serialise 2 N^3/3 flops on a single CU (72 ms at N = 2001), so the
// factorisation is blocked on the host over 128-column panels and the O(N^3) parts go to the matrix-core GEMMs:
//   for each panel k:  L_kk = chol(A_kk) and W_kk = L_kk^-1 (one-workgroup kernels, 128 x 128)
//                      P = A[k+1:, k] W_kk^T                 (GEMM)   -> written back as L[k+1:, k]
//                      A[k+1:, k+1:] -= P P^T  (lower part)  (GEMM)
//   W = L^-1 by block rows:  W[i, :i] = -W_ii (L[i, :i] W[:i, :i])      (2 GEMMs per block row)
// A is overwritten by L (lower triangle; the strict upper triangle is NOT cleared), W gets zeros above the diagonal.
#define CHL_NB 128
// `count` matrices of the same size go through the SAME launch sequence (every launch batched over the matrices):
// the N = 2001 factorisation is ~80 dependent launches of small grids, so W windows cost little more than one.
size_t cholesky_large_batched_workspace_bytes(int N, int count) {
  const int nblk = (N + CHL_NB - 1) / CHL_NB;
  const size_t ldT = (size_t)((N + 1) & ~1);
  const size_t c = (size_t)(count < 1 ? 1 : count);
  return gp_align_up((size_t)4 * nblk * c * sizeof(GemmProblem), 256) + gp_align_up((size_t)2 * nblk * c * sizeof(double*), 256) +
         gp_align_up((size_t)(nblk + 1) * c * sizeof(int), 256) + c * gp_align_up((size_t)CHL_NB * ldT * sizeof(double), 256) + 512;
}
size_t cholesky_large_workspace_bytes(int N) { return cholesky_large_batched_workspace_bytes(N, 1); }

// C (rows x cols, ldc) -> the strided destination of the same problem's A operand: panel write-back after P = A W_kk^T
__global__ void __launch_bounds__(256) chl_panel_copy_kernel(const GemmProblem* __restrict__ probs) {
  const GemmProblem p = probs[blockIdx.y];
  const int64_t total = (int64_t)p.M * p.N;
  double* dst = const_cast<double*>(p.A);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t i = idx / p.N, j = idx - i * p.N;
    dst[i * p.lda + j] = p.C[i * p.ldc + j];
  }
}
__global__ void __launch_bounds__(256) chl_zero_kernel(double* const* __restrict__ mats, int64_t n) {
  double* m = mats[blockIdx.y];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) m[idx] = 0.0;
}

gp_status launch_cholesky_large_batched(gp_handle h, double* const* A, double* const* W, int count, int N, int64_t ld,
                                        void* ws, size_t ws_bytes) {
  if (N <= 0 || count <= 0) return GP_OK;
  if ((ld & 1) || ld < N) return gp_fail(h, GP_ERR_BAD_ARG, "launch_cholesky_large: ld must be even and >= N");
  GpArena ar(ws, ws_bytes);
  const int nblk = (N + CHL_NB - 1) / CHL_NB;
  const size_t C = (size_t)count;
  GemmProblem* d_probs = ar.take<GemmProblem>((size_t)4 * nblk * C);      // [step][kind][matrix]
  double** d_ptrs = ar.take<double*>((size_t)2 * nblk * C);               // [step][L_kk | W_kk][matrix]
  int* d_ints = ar.take<int>((size_t)(nblk + 1) * C);                     // [step][matrix] block sizes, then the ld's
  const int64_t ldT = (N + 1) & ~1;
  const size_t t_stride = gp_align_up((size_t)CHL_NB * ldT * sizeof(double), 256) / sizeof(double);
  double* T0 = ar.take<double>(t_stride * C);     // per matrix: panel (N x 128, ld 128) or block row (128 x N, ld ldT)
  if (!ar.ok) return gp_fail(h, GP_ERR_WORKSPACE, "launch_cholesky_large: workspace too small");
  std::vector<GemmProblem> hp((size_t)4 * nblk * C);
  std::vector<double*> hptr((size_t)2 * nblk * C);
  std::vector<int> hint((size_t)(nblk + 1) * C);
  memset(hp.data(), 0, hp.size() * sizeof(GemmProblem));
  for (int k = 0; k < nblk; k++) {
    const int c0 = k * CHL_NB, nb = (N - c0 < CHL_NB) ? N - c0 : CHL_NB, r0 = c0 + nb, mrem = N - r0;
    for (size_t w = 0; w < C; w++) {
      double* Aw = A[w]; double* Ww = W[w]; double* T = T0 + w * t_stride;
      double* Wkk = Ww + (int64_t)c0 * ld + c0;
      hptr[((size_t)2 * k + 0) * C + w] = Aw + (int64_t)c0 * ld + c0;
      hptr[((size_t)2 * k + 1) * C + w] = Wkk;
      hint[(size_t)k * C + w] = nb;
      { GemmProblem& r = hp[((size_t)4 * k + 0) * C + w];   // P = A[r0:, c0:c0+nb] W_kk^T
        r.A = Aw + (int64_t)r0 * ld + c0; r.lda = ld; r.B = Wkk; r.ldb = ld; r.C = T; r.ldc = CHL_NB; r.M = mrem; r.N = nb; r.K = nb; }
      { GemmProblem& r = hp[((size_t)4 * k + 1) * C + w];   // A[r0:, r0:] -= P P^T
        r.A = T; r.lda = CHL_NB; r.B = T; r.ldb = CHL_NB; r.C = Aw + (int64_t)r0 * ld + r0; r.ldc = ld; r.M = mrem; r.N = mrem; r.K = nb; }
      { GemmProblem& r = hp[((size_t)4 * k + 2) * C + w];   // T = L[c0:c0+nb, :c0] W[:c0, :c0]
        r.A = Aw + (int64_t)c0 * ld; r.lda = ld; r.B = Ww; r.ldb = ld; r.C = T; r.ldc = ldT; r.M = nb; r.N = c0; r.K = c0; }
      { GemmProblem& r = hp[((size_t)4 * k + 3) * C + w];   // W[c0:c0+nb, :c0] = -W_kk T
        r.A = Wkk; r.lda = ld; r.B = T; r.ldb = ldT; r.C = Ww + (int64_t)c0 * ld; r.ldc = ld; r.M = nb; r.N = c0; r.K = nb; }
    }
  }
  for (size_t w = 0; w < C; w++) hint[(size_t)nblk * C + w] = (int)ld;
  GP_HIP_CHECK(h, hipMemcpyAsync(d_probs, hp.data(), hp.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(d_ptrs, hptr.data(), hptr.size() * sizeof(double*), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(d_ints, hint.data(), hint.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // the staging vectors are stack objects
  // the W matrices start as zeros (their upper blocks are never written): step 0's W_kk pointers are their bases
  hipLaunchKernelGGL(chl_zero_kernel, dim3(512, count), dim3(256), 0, h->stream, (double* const*)(d_ptrs + C), (int64_t)N * ld);
  GP_HIP_CHECK(h, hipGetLastError());
  const int* d_ld = d_ints + (size_t)nblk * C;
  for (int k = 0; k < nblk; k++) {
    const int c0 = k * CHL_NB, nb = (N - c0 < CHL_NB) ? N - c0 : CHL_NB, r0 = c0 + nb, mrem = N - r0;
    double* const* dL = d_ptrs + ((size_t)2 * k + 0) * C;
    double* const* dW = d_ptrs + ((size_t)2 * k + 1) * C;
    const int* dM = d_ints + (size_t)k * C;
    GP_CHECK(launch_cholesky_batched(h, dL, dM, d_ld, count, nb, c0));
    GP_CHECK(launch_tri_inverse_batched(h, (const double* const*)dL, dW, dM, d_ld, count));
    if (mrem > 0) {
      GemmFlags f;
      f.transB = 1; f.triB = TRI_UPPER;
      GP_CHECK(launch_gemm_batched(h, d_probs + ((size_t)4 * k + 0) * C, count, mrem, nb, f));
      int blocks = (int)(((int64_t)mrem * nb + 255) / 256);
      if (blocks > 256) blocks = 256;
      hipLaunchKernelGGL(chl_panel_copy_kernel, dim3(blocks, count), dim3(256), 0, h->stream, d_probs + ((size_t)4 * k + 0) * C);
      GP_HIP_CHECK(h, hipGetLastError());
      f = GemmFlags();
      f.transB = 1; f.triC = TRI_LOWER; f.alpha = -1.0; f.beta = 1.0;
      GP_CHECK(launch_gemm_batched(h, d_probs + ((size_t)4 * k + 1) * C, count, mrem, mrem, f));
    }
  }
  for (int k = 1; k < nblk; k++) {
    const int c0 = k * CHL_NB, nb = (N - c0 < CHL_NB) ? N - c0 : CHL_NB;
    GemmFlags f;
    f.triB = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, d_probs + ((size_t)4 * k + 2) * C, count, nb, c0, f));
    f = GemmFlags();
    f.triA = TRI_LOWER; f.alpha = -1.0;
    GP_CHECK(launch_gemm_batched(h, d_probs + ((size_t)4 * k + 3) * C, count, nb, c0, f));
  }
  return GP_OK;
}

gp_status launch_cholesky_large(gp_handle h, double* A, double* W, int N, int64_t ld, void* ws, size_t ws_bytes) {
  double* a[1] = {A};
  double* w[1] = {W};
  return launch_cholesky_large_batched(h, a, w, 1, N, ld, ws, ws_bytes);
}

// One M x M matrix (128 < M <= 1024) as the ELBO engine factors its Kuu batch (engine.hip: cond_batch_factorize, resident
// form): the whole factor from ONE one-workgroup launch, then the inverse blocked over 128-column panels — all diagonal
// blocks in one tri_inverse launch, the block rows below them as two GEMMs each.  The one-workgroup factor + inverse of a
// 512 x 512 matrix takes 1.37 ms (the inverse is the slower half); this takes 0.9.  The descriptors are built and
// uploaded ONCE (chol_inverse_blocked_prepare: the caller's buffers do not move), so a run is launches only and can sit
// inside a recorded launch sequence.
size_t chol_inverse_blocked_workspace_bytes(int M) {
  const int nblk = (M + CHL_NB - 1) / CHL_NB;
  const size_t ldT = (size_t)((M + 1) & ~1);
  return gp_align_up((size_t)2 * nblk * sizeof(GemmProblem), 256) + gp_align_up((size_t)(2 * nblk + 2) * sizeof(double*), 256) +
         gp_align_up((size_t)(2 * nblk + 2) * sizeof(int), 256) + gp_align_up((size_t)CHL_NB * ldT * sizeof(double), 256) + 512;
}

struct CholBlockedLayout {
  GemmProblem* probs; double** ptrs; int* ints; double* T; int nblk;
  // ptrs: [0] A, [1] W, [2 .. 2 + nblk) L_kk, [2 + nblk .. 2 + 2 nblk) W_kk;  ints: [0] M, [1] ld, [2 .. 2 + nblk) nb_k, then nblk x ld
};
static bool chol_blocked_layout(int M, void* ws, size_t ws_bytes, CholBlockedLayout* L) {
  GpArena ar(ws, ws_bytes);
  L->nblk = (M + CHL_NB - 1) / CHL_NB;
  const int64_t ldT = (M + 1) & ~1;
  L->probs = ar.take<GemmProblem>((size_t)2 * L->nblk);
  L->ptrs = ar.take<double*>((size_t)2 * L->nblk + 2);
  L->ints = ar.take<int>((size_t)2 * L->nblk + 2);
  L->T = ar.take<double>((size_t)CHL_NB * ldT);
  return ar.ok;
}

gp_status chol_inverse_blocked_prepare(gp_handle h, double* A, double* W, int M, int64_t ld, void* ws, size_t ws_bytes) {
  if (M <= CHL_NB || (ld & 1) || ld < M) return gp_fail(h, GP_ERR_BAD_ARG, "chol_inverse_blocked: M > 128 and an even ld >= M");
  CholBlockedLayout L;
  if (!chol_blocked_layout(M, ws, ws_bytes, &L)) return gp_fail(h, GP_ERR_WORKSPACE, "chol_inverse_blocked: workspace too small");
  const int nblk = L.nblk;
  const int64_t ldT = (M + 1) & ~1;
  std::vector<GemmProblem> hp((size_t)2 * nblk);
  std::vector<double*> hptr((size_t)2 * nblk + 2);
  std::vector<int> hint((size_t)2 * nblk + 2);
  memset(hp.data(), 0, hp.size() * sizeof(GemmProblem));
  hptr[0] = A; hptr[1] = W; hint[0] = M; hint[1] = (int)ld;
  for (int k = 0; k < nblk; k++) {
    const int c0 = k * CHL_NB, nb = (M - c0 < CHL_NB) ? M - c0 : CHL_NB;
    double* Wkk = W + (int64_t)c0 * ld + c0;
    hptr[2 + k] = A + (int64_t)c0 * ld + c0; hptr[2 + nblk + k] = Wkk;
    hint[2 + k] = nb; hint[2 + nblk + k] = (int)ld;
    { GemmProblem& r = hp[2 * k + 0];   // T = L[c0:c0+nb, :c0] W[:c0, :c0]
      r.A = A + (int64_t)c0 * ld; r.lda = ld; r.B = W; r.ldb = ld; r.C = L.T; r.ldc = ldT; r.M = nb; r.N = c0; r.K = c0; }
    { GemmProblem& r = hp[2 * k + 1];   // W[c0:c0+nb, :c0] = -W_kk T
      r.A = Wkk; r.lda = ld; r.B = L.T; r.ldb = ldT; r.C = W + (int64_t)c0 * ld; r.ldc = ld; r.M = nb; r.N = c0; r.K = nb; }
  }
  GP_HIP_CHECK(h, hipMemcpyAsync(L.probs, hp.data(), hp.size() * sizeof(GemmProblem), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(L.ptrs, hptr.data(), hptr.size() * sizeof(double*), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipMemcpyAsync(L.ints, hint.data(), hint.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // the staging vectors are stack objects
  return GP_OK;
}

gp_status chol_inverse_blocked_run(gp_handle h, int M, int64_t ld, void* ws, size_t ws_bytes) {
  CholBlockedLayout L;
  if (!chol_blocked_layout(M, ws, ws_bytes, &L)) return gp_fail(h, GP_ERR_WORKSPACE, "chol_inverse_blocked: workspace too small");
  const int nblk = L.nblk;
  GP_CHECK(launch_cholesky_batched(h, (double* const*)L.ptrs, L.ints, L.ints + 1, 1, M, 0));     // the whole factor, one workgroup
  hipLaunchKernelGGL(chl_zero_kernel, dim3(256, 1), dim3(256), 0, h->stream, (double* const*)(L.ptrs + 1), (int64_t)M * ld);
  GP_HIP_CHECK(h, hipGetLastError());
  GP_CHECK(launch_tri_inverse_batched(h, (const double* const*)(L.ptrs + 2), (double* const*)(L.ptrs + 2 + nblk), L.ints + 2,
                                      L.ints + 2 + nblk, nblk));
  for (int k = 1; k < nblk; k++) {
    const int c0 = k * CHL_NB, nb = (M - c0 < CHL_NB) ? M - c0 : CHL_NB;
    GemmFlags f;
    f.triB = TRI_LOWER;
    GP_CHECK(launch_gemm_batched(h, L.probs + 2 * k + 0, 1, nb, c0, f));
    f = GemmFlags();
    f.triA = TRI_LOWER; f.alpha = -1.0;
    GP_CHECK(launch_gemm_batched(h, L.probs + 2 * k + 1, 1, nb, c0, f));
  }
  return GP_OK;
}

// after a panel-blocked factorisation: clear the blocks strictly above the block diagonal (the one-workgroup kernel
// clears the upper triangle only inside the diagonal blocks it is given), so that L is usable as a dense operand
__global__ void __launch_bounds__(256) zero_upper_blocks_kernel(double* const* __restrict__ mats, const int* __restrict__ Ms,
                                                                const int* __restrict__ lds_, int nb) {
  double* A = mats[blockIdx.y];
  const int M = Ms[blockIdx.y];
  const int64_t ld = lds_[blockIdx.y];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)M * M; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / M), j = (int)(idx % M);
    if (j / nb > i / nb) A[(int64_t)i * ld + j] = 0.0;
  }
}

gp_status launch_zero_upper_blocks_batched(gp_handle h, double* const* d_mats, const int* d_M, const int* d_ld, int batch,
                                           int maxM, int nb) {
  if (batch <= 0 || maxM <= nb) return GP_OK;
  int blocks = (int)(((int64_t)maxM * maxM + 255) / 256);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(zero_upper_blocks_kernel, dim3(blocks, batch), dim3(256), 0, h->stream, d_mats, d_M, d_ld, nb);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// Reads the device status word (syncs the stream).  Clears it afterwards.
gp_status check_not_pd(gp_handle h) {
  int32_t st[4] = {0, 0, 0, 0};
  GP_HIP_CHECK(h, hipMemcpyAsync(st, h->d_status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
  GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  if (st[0] != 0) {
    h->not_pd_index = st[1];
    char buf[160];
    snprintf(buf, sizeof(buf), "Cholesky failed: matrix %d is not positive definite (pivot %d)", st[2], st[1]);
    h->last_error = buf;
    GP_HIP_CHECK(h, hipMemsetAsync(h->d_status, 0, sizeof(st), h->stream));
    return GP_ERR_NOT_PD;
  }
  return GP_OK;
}
