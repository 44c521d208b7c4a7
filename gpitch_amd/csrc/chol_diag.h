// chol_diag.h — the in-register factorisation of a 32 x 32 diagonal block by one wavefront, shared by the one-workgroup
// kernels (chol.hip) and the workgroup-cluster kernel (chol_cluster.hip)
#pragma once
#include "common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define CH_NB 32
#define CH_THREADS 512
#define CH_WAVES (CH_THREADS / 64)
#define CH_LDP 34     // LDS row stride (doubles) of the 32-column panel / inverse blocks: 2 (mod 32), so the 16 x 4 lanes of a ds_read_b64 fragment read (row lc, column 4 kk + kq) hit 32 different bank pairs per half-wavefront

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
__device__ __noinline__ void chol_diag_block(PT A, int64_t ld, int M, int k0, int lane,
                                                double (*Dinv)[CH_LDP], int* __restrict__ status, int b, int pbase) {
  // Factor and inverse in ONE pass over the pivots, the two halves of the wavefront doing one each:
  //   lanes 0..31  ("low"):  lane i holds row i of the block, v[c] = A[i][c] -> L[i][c];
  //   lanes 32..63 ("high"): lane 32 + c holds column c of the inverse, v[r] = X[r][c] (starts as the identity).
  // Step j of the factorisation makes column j of L final; the column-oriented forward substitution X = L^-1 needs
  // exactly that column at its own step j (x_j *= 1 / L_jj;  x_r -= L_rj x_j for r > j), and the trailing update of the
  // factor needs the same broadcasts (row_c -= L_ij L_cj for c > j): ONE v_fma per broadcast serves both halves,
  //   v[c] = fma(-v[j], L_cj, v[c])     with v[j] = L_ij (low lanes) or x_j (high lanes).
  // The inverse therefore costs nothing beyond the factor (it was a second 496-broadcast pass before: the serial
  // 32-pivot block is the critical path of the one-workgroup factorisation for all but the first panels).
  __builtin_amdgcn_s_setprio(3);   // the co-resident wavefront is issuing 64-cycle float64 MFMAs on the same vector ALU
  const int nb = min(CH_NB, M - k0);
  double v[CH_NB];
  const bool low = lane < 32;
  const int i = lane & 31;
  const int ie = low ? i : 1 << 20;      // "row index" for the select below: high lanes always scale by 1 / L_jj
  // unconditional loads from clamped (always valid) addresses, then select: no divergent branches; 32-bit element
  // offsets off the wave-uniform base (a whole block: one offset register and immediates)
  const unsigned roff = (unsigned)(k0 + min(i, nb - 1)) * (unsigned)ld + (unsigned)k0;
  // (every lane loads — the high half the same rows again — and the loaded values are pinned before the select: a load
  // that only the low half consumes is otherwise sunk into a divergent branch of its own, 32 serial round trips)
#pragma unroll
  for (int c = 0; c < CH_NB; c++) v[c] = A[roff + (unsigned)min(c, nb - 1)];
#pragma unroll
  for (int c = 0; c < CH_NB; c++) asm volatile("" : "+v"(v[c]));
#pragma unroll
  for (int c = 0; c < CH_NB; c++) v[c] = (low && i < nb && c <= i) ? v[c] : (i == c ? 1.0 : 0.0);
  int bad = -1;        // first non-positive / non-finite pivot (wave-uniform), reported once after the loop
  // pivot j's sqrt / reciprocal chain (~15 dependent float64 operations) is started as soon as column j has had its
  // last update — the FIRST thing step j - 1 does — so that the other 30 - j updates of step j - 1 issue in its shadow
  double s, ri;
  auto pivot = [&](int j) {
    double djj = lane_bcast(v[j], j);
    const bool ok = (djj > 0.0) && (djj <= 1.7976931348623157e308);
    bad = (!ok && bad < 0 && j < nb) ? j : bad;
    djj = ok ? djj : 1.0;                       // keep going finite
    pivot_sqrt_recip(djj, s, ri);
  };
  pivot(0);
#pragma unroll
  for (int j = 0; j < CH_NB; j++) {
    const double nv = (ie > j) ? v[j] * ri : (ie == j ? s : 0.0);
    v[j] = nv;
    if (j + 1 < CH_NB) {
      v[j + 1] = fma(-nv, lane_bcast(nv, j + 1), v[j + 1]);
      pivot(j + 1);
    }
#ifndef CH_DIAG_PROBE_NO_UPDATES        // (timing probe only: wrong results)
#pragma unroll
    for (int c = j + 2; c < CH_NB; c++) v[c] = fma(-nv, lane_bcast(nv, c), v[c]);   // lane c (low half) holds L_cj
#endif
  }
  if (bad >= 0 && lane == 0) {
    if (atomicCAS(&status[0], 0, 1) == 0) { status[1] = pbase + k0 + bad; status[2] = b; }
  }
  // pin v[] here: otherwise the arithmetic is sunk into the lane-conditional stores below while its (convergent) lane
  // broadcasts stay outside, hundreds of them live at once and spilled
#pragma unroll
  for (int c = 0; c < CH_NB; c++) asm volatile("" : "+v"(v[c]));
  if (low) {
    const unsigned soff = (unsigned)(k0 + i) * (unsigned)ld + (unsigned)k0;
#pragma unroll
    for (int c = 0; c < CH_NB; c++)
      if (i < nb && c <= i) A[soff + (unsigned)c] = v[c];
  } else {
#pragma unroll
    for (int r = 0; r < CH_NB; r++) Dinv[r][i] = v[r];
  }
  __builtin_amdgcn_s_setprio(0);
}
