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

#include "chol_diag.h"


#ifdef CH_STAMPS
// diagnostic build (never shipped; tools/chol_stamps.py): s_memtime stamps of matrix 0's workgroup, per 32-column panel
// [panel start, panel product done, trailing update done, look-ahead diagonal block start, end]
__device__ unsigned long long ch_stamps[5 * 64 + 8 * 8];
extern "C" int gp_debug_chol_stamps(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(ch_stamps), sizeof(ch_stamps)) == hipSuccess ? 0 : -1;
}
#define CH_STAMP(k, i) do { if (b == 0 && (k) < 64) ch_stamps[5 * (k) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// per-wavefront phase sums of panel 0's update loop: [operand reads + next-tile draw / loads issued, MFMAs done, stores done, tiles]
#define CH_PH_DECL unsigned long long ph_[4] = {0, 0, 0, 0}, pt_ = 0
#define CH_PH_T0 do { if (b == 0 && k0 == 0) pt_ = __builtin_amdgcn_s_memtime(); } while (0)
#define CH_PH(i) do { if (b == 0 && k0 == 0) { if ((i) == 1 || (i) == 2) __builtin_amdgcn_s_waitcnt(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_[i] += n_ - pt_; pt_ = n_; } } while (0)
#define CH_PH_OUT do { if (b == 0 && k0 == 0 && lane == 0) { for (int q_ = 0; q_ < 4; q_++) ch_stamps[5 * 64 + 8 * wave + q_] = ph_[q_]; } } while (0)
#else
#define CH_PH_DECL
#define CH_PH_T0 do { } while (0)
#define CH_PH(i) do { } while (0)
#define CH_PH_OUT do { } while (0)
#define CH_STAMP(k, i) do { } while (0)
#endif


// byte pointer of the matrix pointer's address space, and a wave-uniform (SGPR) copy of one: accesses are written as
// (uniform byte base) + (32-bit per-lane byte offset, fixed for the whole kernel) + (literal), which is the scalar-base
// form of global_load / global_store — no vector instruction per access.  The float64 matrix instruction holds the SIMD's
// vector ALU for its 64 cycles, so every vector instruction a wavefront issues between its matrix phases
// queues behind the co-resident wavefront's MFMAs: the update loop below ran at 8-9 k cycles per 32 x 32 tile and
// wavefront for 2 k of matrix time until its address arithmetic, accumulator subtraction and register copies were gone.
template <typename PT> struct ChBytes;
template <> struct ChBytes<ch_gptr> { typedef char __attribute__((address_space(1))) * type; };
template <> struct ChBytes<double*> { typedef char* type; };
template <typename BT> __device__ __forceinline__ BT ch_uni(BT p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (BT)(((uint64_t)hi << 32) | lo);
}
#define CH_TAB 528        // (row, column) table of the update's macro tiles: every panel of a matrix of up to 1056 rows

// One 32 x 32 macro tile of the trailing update in its general form (rows past M, a panel narrower than 32, operands
// from the matrix instead of the LDS panel, the block diagonal's c <= i mask): everything the fast path of chol_body does
// not take.  Out of line so that its registers and branches stay out of the hot loop.
template <typename PT>
__device__ __noinline__ void chol_tile_slow(PT A, unsigned ldu, int M, int nb, int k0, int r0, int rb, int cb,
                                            const double* P, int p_lds, int lane) {
  const int kq = lane >> 4, lc = lane & 15;
  double af0[8], af1[8], bf0[8], bf1[8];
#pragma unroll
  for (int kk = 0; kk < 8; kk++) {
    const int kc = kk * 4 + kq;
    const bool kin = kc < nb;
    const int ra0 = rb + lc, ra1 = rb + 16 + lc, rb0 = cb + lc, rb1 = cb + 16 + lc;
    if (p_lds) {   // rows >= M were never written: guard; columns >= nb hold zeros
      af0[kk] = (ra0 < M) ? P[(ra0 - r0) * CH_LDP + kc] : 0.0;
      af1[kk] = (ra1 < M) ? P[(ra1 - r0) * CH_LDP + kc] : 0.0;
      bf0[kk] = (rb0 < M) ? P[(rb0 - r0) * CH_LDP + kc] : 0.0;
      bf1[kk] = (rb1 < M) ? P[(rb1 - r0) * CH_LDP + kc] : 0.0;
    } else {
      af0[kk] = (kin && ra0 < M) ? A[(unsigned)ra0 * ldu + (unsigned)(k0 + kc)] : 0.0;
      af1[kk] = (kin && ra1 < M) ? A[(unsigned)ra1 * ldu + (unsigned)(k0 + kc)] : 0.0;
      bf0[kk] = (kin && rb0 < M) ? A[(unsigned)rb0 * ldu + (unsigned)(k0 + kc)] : 0.0;
      bf1[kk] = (kin && rb1 < M) ? A[(unsigned)rb1 * ldu + (unsigned)(k0 + kc)] : 0.0;
    }
  }
  d4 c[4];
  const unsigned o0 = (unsigned)(rb + kq) * ldu + (unsigned)(cb + lc);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int rw0 = rb + kq + 4 * r, rw1 = rw0 + 16, cw0 = cb + lc, cw1 = cw0 + 16;
    const unsigned o = o0 + (unsigned)(4 * r) * ldu;
    c[0][r] = (rw0 < M && cw0 <= rw0) ? A[o] : 0.0;
    c[1][r] = (rw0 < M && cw1 <= rw0) ? A[o + 16u] : 0.0;
    c[2][r] = (rw1 < M && cw0 <= rw1) ? A[o + 16u * ldu] : 0.0;
    c[3][r] = (rw1 < M && cw1 <= rw1) ? A[o + 16u * ldu + 16u] : 0.0;
  }
#pragma unroll
  for (int kk = 0; kk < 8; kk++) {
    c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[kk], bf0[kk], c[0], 0, 0, 1);
    c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[kk], bf1[kk], c[1], 0, 0, 1);
    c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[kk], bf0[kk], c[2], 0, 0, 1);
    c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[kk], bf1[kk], c[3], 0, 0, 1);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int rw0 = rb + kq + 4 * r, rw1 = rw0 + 16, cw0 = cb + lc, cw1 = cw0 + 16;
    const unsigned o = o0 + (unsigned)(4 * r) * ldu;
    if (rw0 < M && cw0 <= rw0) A[o] = c[0][r];
    if (rw0 < M && cw1 <= rw0) A[o + 16u] = c[1][r];
    if (rw1 < M && cw0 <= rw1) A[o + 16u * ldu] = c[2][r];
    if (rw1 < M && cw1 <= rw1) A[o + 16u * ldu + 16u] = c[3][r];
  }
}

// Cholesky of one matrix by one workgroup (body shared by chol_kernel and chol_inverse_kernel)
template <typename PT>
__device__ __forceinline__ void chol_body(PT A, const int M, const int64_t ld, int* __restrict__ status,
                                          const int b, const int panel_rows_cap, const int pivot_base) {
  typedef typename ChBytes<PT>::type BT;
  extern __shared__ __attribute__((aligned(16))) double chol_smem[];
  // [ D: 2 x 32 x CH_LDP (inverse of the current / next diagonal factor) | P: panel X, (M-32) x CH_LDP when it fits ]
  double (*D)[CH_NB][CH_LDP] = reinterpret_cast<double (*)[CH_NB][CH_LDP]>(chol_smem);
  double* P = chol_smem + 2 * CH_NB * CH_LDP;
  __shared__ int tile_counter;
  __shared__ unsigned tile_tab[CH_TAB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, lc = lane & 15;
  const bool p_lds = (panel_rows_cap >= M - CH_NB);   // operands of the trailing update come from LDS
  const unsigned ldu = (unsigned)ld;
  const size_t ld8 = (size_t)ldu * 8;
  const BT Ab = (BT)A;
  const uint32_t vC = (uint32_t)((kq * ldu + lc) * 8u);   // lane part of element (row kq, column lc): C tiles, panel stores
  const uint32_t vA = (uint32_t)((lc * ldu + kq) * 8u);   // lane part of element (row lc, column kq): panel operand loads
  auto at = [](BT base, uint32_t voff, int imm) -> decltype(auto) { return (*(PT)(base + voff + imm)); };
  // a lane offset re-defined where it is used: the scalar-base addressing mode is matched per basic block, and a 32-bit
  // offset hoisted out of the loop arrives there already widened to a 64-bit register pair
  auto here = [](uint32_t v) { asm volatile("" : "+v"(v)); return v; };

  // macro tile t of a trailing update = (block row mi, block column mj) of the lower triangle, row by row: the same
  // numbering for every panel, so it is tabulated once
  const int tmax = (M + 31) / 32;
  const int ntab = min(tmax * (tmax + 1) / 2, CH_TAB);
  for (int t = tid; t < ntab; t += CH_THREADS) {
    int mi = (int)((__fsqrt_rn(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((mi + 1) * (mi + 2) / 2 <= t) mi++;
    while (mi * (mi + 1) / 2 > t) mi--;
    tile_tab[t] = (unsigned)mi | ((unsigned)(t - mi * (mi + 1) / 2) << 16);
  }
  if (wave == 0) chol_diag_block(A, ld, M, 0, lane, D[0], status, b, pivot_base);
  __syncthreads();

  int pb = 0;
  for (int k0 = 0; k0 < M; k0 += CH_NB, pb ^= 1) {
    const int nb = min(CH_NB, M - k0);
    const int r0 = k0 + nb;  // first trailing row
    const int R = M - r0;
    if (R <= 0) break;
    if (tid == 0) CH_STAMP(k0 / CH_NB, 0);
    // ---- panel: X = A_panel * L_kk^-T on the matrix cores, 16 rows per wavefront step; the operand rows of the next
    //      step are requested before the products of this one (a dependent L2 round trip per step otherwise) ----------
    auto panel_load = [&](int rg, double (&af)[8]) {
      const int rowb = r0 + rg * 16;
      if (nb == CH_NB && rowb + 16 <= M) {
        const BT pbase = ch_uni(Ab + ((size_t)rowb * ldu + k0) * 8);
        const uint32_t va = here(vA);
#pragma unroll
        for (int kk = 0; kk < 8; kk++) af[kk] = at(pbase, va, kk * 32);
      } else {
        const int ra = rowb + lc;
        const unsigned o = (unsigned)min(ra, M - 1) * ldu + (unsigned)(k0 + kq);
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          const int kc = kk * 4 + kq;
          const double a = A[o + (unsigned)(kc < nb ? kk * 4 : 0)];
          af[kk] = (ra < M && kc < nb) ? a : 0.0;
        }
      }
    };
    double afc[8];
    if (wave * 16 < R) panel_load(wave, afc);
    for (int rg = wave; rg * 16 < R; rg += CH_WAVES) {
      double afn[8];
      const bool more = (rg + CH_WAVES) * 16 < R;      // wave-uniform
      if (more) panel_load(rg + CH_WAVES, afn);
      d4 o[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
      {
        const double* Dl = &D[pb][lc][kq];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {  // B[k][j] = (L_kk^-1)[j][k]; two independent accumulators
          o[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(afc[kk], Dl[kk * 4], o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(afc[kk], Dl[16 * CH_LDP + kk * 4], o[1], 0, 0, 0);
        }
      }
      const int rowb = r0 + rg * 16;
      if (nb == CH_NB && rowb + 16 <= M) {
        const BT pbase = ch_uni(Ab + ((size_t)rowb * ldu + k0) * 8);
        double* Pl = P + (rowb - r0 + kq) * CH_LDP + lc;
        const uint32_t vc = here(vC);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const BT rb_ = ch_uni(pbase + (size_t)(4 * r) * ld8);
          at(rb_, vc, 0) = o[0][r]; at(rb_, vc, 128) = o[1][r];
          if (p_lds) { Pl[4 * r * CH_LDP] = o[0][r]; Pl[4 * r * CH_LDP + 16] = o[1][r]; }
        }
      } else {
#pragma unroll
        for (int tj = 0; tj < 2; tj++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int rw = rowb + kq + 4 * r, cw = tj * 16 + lc;
            if (rw < M && cw < nb) A[(unsigned)rw * ldu + (unsigned)(k0 + cw)] = o[tj][r];
            if (p_lds && rw < M) P[(rw - r0) * CH_LDP + cw] = (cw < nb) ? o[tj][r] : 0.0;
          }
      }
      if (more) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) afc[kk] = afn[kk];
      }
    }
    if (tid == 0) tile_counter = 0;
    __threadfence_block();
    __syncthreads();
    if (tid == 0) CH_STAMP(k0 / CH_NB, 1);
    // ---- trailing update A22 -= X X^T, 32 x 32 macro tiles handed out dynamically.  Macro tile 0 is the NEXT diagonal
    //      block: the wavefront that draws it factorises that block straight away (look-ahead), overlapping the serial
    //      32-column factorisation with the other waves' updates.
    //      A wavefront draws its NEXT tile and requests that tile's C block before the products of the current one; the C
    //      block is loaded straight into the accumulators and the product runs with the A operand negated (the neg
    //      modifier of the float64 MFMA), so the read-modify-write is load -> 32 MFMAs -> store with no vector arithmetic
    //      in between; tiles on the block diagonal are updated whole (the part above the diagonal is never read: the
    //      diagonal-block factorisation selects c <= i, and it is cleared at the end).
    const int nmt = (R + 31) >> 5;
    const int total = nmt * (nmt + 1) / 2;
    int t, rbase = 0, cbase = 0;
    bool fast = false;
    auto draw = [&]() {       // next tile of this wavefront: t, (rbase, cbase), fast = whole tile, operands in the LDS panel
      int tt = 0;
      unsigned pk = 0;
      if (lane == 0) {
        tt = atomicAdd(&tile_counter, 1);
        if (tt < ntab) pk = tile_tab[tt];
      }
      t = __builtin_amdgcn_readfirstlane(tt);
      pk = __builtin_amdgcn_readfirstlane(pk);
      int mi = (int)(pk & 0xffffu), mj = (int)(pk >> 16);
      if (t >= CH_TAB) {      // (matrices beyond the table: arithmetic decode)
        mi = (int)((__fsqrt_rn(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((mi + 1) * (mi + 2) / 2 <= t) mi++;
        while (mi * (mi + 1) / 2 > t) mi--;
        mj = t - mi * (mi + 1) / 2;
      }
      rbase = r0 + mi * 32; cbase = r0 + mj * 32;
      fast = p_lds && (nb == CH_NB) && (rbase + 32 <= M) && (t < total);
    };
    // C block of a whole tile (rb, cb) -> c[2 h + w][r] = element (rb + kq + 4 r + 16 h, cb + lc + 16 w); every row base
    // goes through ch_uni: the compiler otherwise re-associates (base + row step) + lane offset into a 64-bit vector add
    auto load_c = [&](int rb, int cb, d4 (&c)[4]) {
      const BT tb = Ab + ((size_t)rb * ldu + cb) * 8;
      const uint32_t vc = here(vC);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const BT r0_ = ch_uni(tb + (size_t)(4 * r) * ld8), r1_ = ch_uni(tb + (size_t)(4 * r + 16) * ld8);
        c[0][r] = at(r0_, vc, 0); c[1][r] = at(r0_, vc, 128); c[2][r] = at(r1_, vc, 0); c[3][r] = at(r1_, vc, 128);
      }
    };
    auto after_tile0 = [&]() {     // the wavefront that has just updated the next diagonal block factorises it
      __threadfence_block();  // this wave's own updates of the block it is about to read back
      if (lane == 0) CH_STAMP(k0 / CH_NB, 3);
      chol_diag_block(A, ld, M, r0, lane, D[pb ^ 1], status, b, pivot_base);
      if (lane == 0) CH_STAMP(k0 / CH_NB, 4);
    };
    CH_PH_DECL;
    // `cbuf` holds the C block of the tile in hand (requested a tile ago).  The first k-step reads it as the C operand and
    // writes the accumulators, which frees it for the NEXT tile's block right away: requested before the other 28
    // products, no register copy, no second buffer.
    d4 cbuf[4];
    draw();
    if (fast) load_c(rbase, cbase, cbuf);
    while (t < total) {
      const int tc = t, rb = rbase, cb = cbase;
      if (!fast) {       // general form, no prefetch
        chol_tile_slow(A, ldu, M, nb, k0, r0, rb, cb, P, (int)p_lds, lane);
        if (tc == 0) after_tile0();
        draw();
        if (fast) load_c(rbase, cbase, cbuf);
        continue;
      }
      CH_PH_T0;
      double af0[8], af1[8], bf0[8], bf1[8];
      {
        const double* Pa = P + (rb - r0 + lc) * CH_LDP + kq;
        const double* Pb = P + (cb - r0 + lc) * CH_LDP + kq;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          af0[kk] = Pa[kk * 4]; af1[kk] = Pa[16 * CH_LDP + kk * 4];
          bf0[kk] = Pb[kk * 4]; bf1[kk] = Pb[16 * CH_LDP + kk * 4];
        }
      }
      d4 acc[4];       // C - X X^T: neg:[1,0,0] negates the A operand
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[0], bf0[0], cbuf[0], 0, 0, 1);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[0], bf1[0], cbuf[1], 0, 0, 1);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[0], bf0[0], cbuf[2], 0, 0, 1);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[0], bf1[0], cbuf[3], 0, 0, 1);
      // the next tile (the wavefront about to factorise the diagonal block draws its next one afterwards: a tile
      // claimed now would sit out the whole factorisation)
      if (tc != 0) {
        draw();
        if (fast) load_c(rbase, cbase, cbuf);
      }
      CH_PH(0);
#pragma unroll
      for (int kk = 1; kk < 8; kk++) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[kk], bf0[kk], acc[0], 0, 0, 1);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af0[kk], bf1[kk], acc[1], 0, 0, 1);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[kk], bf0[kk], acc[2], 0, 0, 1);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(af1[kk], bf1[kk], acc[3], 0, 0, 1);
      }
      CH_PH(1);
      {
        const BT tb = Ab + ((size_t)rb * ldu + cb) * 8;
        const uint32_t vc = here(vC);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const BT r0_ = ch_uni(tb + (size_t)(4 * r) * ld8), r1_ = ch_uni(tb + (size_t)(4 * r + 16) * ld8);
          at(r0_, vc, 0) = acc[0][r]; at(r0_, vc, 128) = acc[1][r]; at(r1_, vc, 0) = acc[2][r]; at(r1_, vc, 128) = acc[3][r];
        }
      }
      CH_PH(2);
#ifdef CH_STAMPS
      if (b == 0 && k0 == 0) ph_[3] += 1;
#endif
      if (tc == 0) {
        after_tile0();
        draw();
        if (fast) load_c(rbase, cbase, cbuf);
      }
    }
    CH_PH_OUT;
    __threadfence_block();
    __syncthreads();
    if (tid == 0) CH_STAMP(k0 / CH_NB, 2);
  }
  // zero the strictly-upper triangle so L can be used as a dense operand (a row per wavefront: no index division)
  for (int i = wave; i < M; i += CH_WAVES)
    for (int j = i + 1 + lane; j < M; j += 64) A[(unsigned)i * ldu + (unsigned)j] = 0.0;
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
#define CHS_OFF (2 * CH_NB * CH_LDP + (CHS_M - CH_NB) * CH_LDP)     // chol_body's own D | P region
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
  const size_t dbytes = (size_t)2 * CH_NB * CH_LDP * sizeof(double);
  const int rows = maxM > CH_NB ? maxM - CH_NB : 0;
  const size_t pbytes = (size_t)rows * CH_LDP * sizeof(double);
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
  { gp_status st = GP_OK; if (launch_cholesky_cluster_single(h, A, W, M, ld, 0, &st)) return st; }
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
  { gp_status st = GP_OK; if (launch_cholesky_cluster_single(h, A, nullptr, M, ld, pivot_base, &st)) return st; }
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
    const bool stalled = (st[0] == 2);      // chol_cluster.hip: a wavefront gave up waiting for the cluster's exchange
    if (stalled) snprintf(buf, sizeof(buf), "Cholesky failed: the workgroup cluster stalled (GPITCH_AMD_SWITCHES=chol_cluster=0 selects the one-workgroup kernels)");
    else snprintf(buf, sizeof(buf), "Cholesky failed: matrix %d is not positive definite (pivot %d)", st[2], st[1]);
    h->last_error = buf;
    GP_HIP_CHECK(h, hipMemsetAsync(h->d_status, 0, sizeof(st), h->stream));
    return stalled ? GP_ERR_HIP : GP_ERR_NOT_PD;
  }
  return GP_OK;
}
