// gemm_wave.hip — the float64 strip products with NO LDS and NO barrier: every wavefront owns a 64 x 64 output tile and
// fetches its MFMA operand fragments straight from global memory (gfx950, round 4).
//
//   role 1  A   = W Kuf      tf.matrix_triangular_solve(Lm, Kmn) + reduce_sum(A^2), A^T q_mu   (GPflow conditional,
//   role 2  LTA = Lq^T A     tf.matmul(Lq^T, A) -> reduce_sum(LTA^2) only                        gpitch/pdgp.py:147-155)
//   role 3  G   = R (A D)    backward: Kuf_bar (TF reverse mode of the same lines)
//
// Why a third form.  gemm_strip.hip's tiles (128 x 128 per workgroup, operands staged through LDS, one barrier per K-tile,
// two workgroups per CU) run the dense product at 0.83 of the matrix peak but the triangular ones at 0.65-0.75
// (profiles/r03/strip_stamps.txt): a K-tile inside the diagonal block carries 8-56 of 64 MFMAs per wavefront against
// the same barrier, LDS round trip and global-load latency, and with two wavefronts per SIMD nothing else is there to
// fill the gap.  Here a wavefront never waits for another one:
//  * v_mfma_f64_16x16x4_f64 wants  A: lane (kq, lc) = op(A)[row lc][k kq],  B: lane (kq, lc) = B[k kq][column lc].  The
//    contraction index may be visited in any order as long as both operands agree, and so may the 16 columns (rows) a
//    tile covers as long as the epilogue knows.  So one 16-byte load per lane fills TWO fragments:
//      op(A) k-contiguous (W, R):  lane (kq, lc) <- rows 16 a + lc, k0 + 2 kq, k0 + 2 kq + 1      (k-steps s = 0, 1)
//      op(A) row-contiguous (Lq^T): lane (kq, lc) <- k0 + 2 kq + s, rows 32 a' + 2 lc, + 1        (row tiles 2 a', 2 a' + 1)
//      B (strip, frames contiguous): lane (kq, lc) <- k0 + 2 kq + s, columns 32 h + 2 lc, + 1     (column tiles 2 h, 2 h + 1)
//    (a wavefront's B load is 4 k-rows x 256 contiguous bytes; its A load 16 rows x 64 bytes of an L2-resident matrix);
//  * the accumulators then hold columns j0 + 32 h + 2 lc + {0, 1} side by side: the epilogue stores 16 bytes per lane,
//    4 rows x 256 bytes per instruction, straight from the registers — no LDS transposition;
//  * operands are requested two 8-deep chunks ahead into a ring of three register sets (32 VGPRs each beside the 128
//    accumulator registers), addresses = scalar base (advanced by the scalar ALU) + one constant lane offset: the K loop
//    issues MFMAs, loads and scalar adds only;
//  * a workgroup is four wavefronts on four ADJACENT 64-column strips of the same row tile(s): they read the same rows
//    of op(A) at the same time (one fetch into the CU's L1), each its own columns of B;
//  * triangular op(A): row tiles are dealt in pairs (t, last - t) to one wavefront, so every wavefront of a launch walks
//    the same number of chunks; inside the diagonal block 16-row MFMA tiles that are structurally zero are skipped
//    (wave-uniform), and the matrices themselves carry exact zeros above (below) the diagonal — no masks.
// Column scaling of role 3 (D = 2 gv, indexed by the frame = output column) commutes with the product: it is applied to
// the finished tile (64 multiplies per tile instead of 4 per k-step).
// Whole, aligned problems only (M a multiple of 64, N of 256, K-structure = M, even leading dimensions): the launcher
// returns false otherwise and gemm_strip.hip / gemm.hip run.
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef const char __attribute__((address_space(1))) * gcbytes;
typedef char __attribute__((address_space(1))) * gbytes;
typedef const double __attribute__((address_space(1))) * gcptr;
typedef double __attribute__((address_space(1))) * gptr;
typedef const dbl2 __attribute__((address_space(1))) * gcptr2;
typedef dbl2 __attribute__((address_space(1))) * gptr2;
typedef unsigned int gw_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int gw_u2 __attribute__((ext_vector_type(2)));

#define GW_T 64                 // rows and columns of a wavefront's tile
#define GW_CH 8                 // k per chunk (two MFMA k-steps)

struct WaveFlags {
  int t0, t1;         // 64-row tiles [t0, t1) of every problem belong to this launch
  int nunits;         // work units per 256-column group: (t1 - t0 + 1) / 2 pairs (triangular) or t1 - t0 tiles (dense)
  int epi;            // EPI_* bitmask
  double alpha;       // role 3: a power of two, folded into the column scales
  int N;              // frames (row stride of the partial-sum arrays)
  int pad_;
  const double* xcols; // role 5: the frames x (the B strip's columns; not part of any descriptor: pdgp.hip pdgp_bind)
};

// One 64 x 64 tile: rows [i0, i0 + 64) of op(A) x columns [j0, j0 + 64) of B over k in [kbeg, kend) (multiples of 8).
// TAG 1: op(A) = A lower triangular (k < i0 + 64; zeros above the diagonal are IN the matrix).  TAG 2: op(A) = A^T with A
// lower triangular (k >= i0), walked downwards.  TAG 3: dense.  TAG 5 (KT = a stationary kernel type): TAG 3's product with the
// Kuf-side hyper-gradient contraction of that kernel as its epilogue (gemm_strip.hip role 5; bwd.hip hyper_contract_kernel:
// sum_ij (Kuf_bar_ij + alpha_i gm_j) dK_ij / d(variance, lengthscale)), nothing stored: one partial record per wavefront tile.
template <int TAG, int KT = -1>
__device__ __forceinline__ void gw_tile(const GemmProblem& p, const WaveFlags& f, const int i0, const int j0, const int lane,
                                        double* etab = nullptr) {
  constexpr bool DENSE = (TAG == 3 || TAG == 5);
  constexpr bool TA = (TAG == 2);
  constexpr bool KDOWN = (TAG == 2);
  const int lc = lane & 15, kq = lane >> 4;
  int kbeg = 0, kend = p.K;
  if (TAG == 1) kend = i0 + GW_T;
  if (TAG == 2) kbeg = i0;
  const int nch = (kend - kbeg) / GW_CH;
  const int kfirst = KDOWN ? kend - GW_CH : kbeg;

  // ---- operand addresses: BUFFER loads — resource descriptor (SGPRs) + one constant 32-bit lane offset (VGPR) + a scalar
  // offset that the scalar ALU advances: no vector instruction per load (global_load with a 64-bit address costs a
  // v_lshl_add_u64 per distinct scalar base, and next to float64 MFMAs every vector instruction costs matrix time)
  // !TA: soff[a] -> row i0 + 16 a, column kfirst; lane offset (lc * lda + 2 kq) * 8; a chunk further = +- 64 bytes
  //  TA: soff[2 s + a'] -> row kfirst + s, column i0 + 32 a'; lane offset (2 kq * lda + 2 lc) * 8; a chunk = +- 8 lda * 8
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7fffffff, 0x00020000);
  int soA[4], soB[2];
  int voffA, stepA;
  if (!TA) {
#pragma unroll
    for (int a = 0; a < 4; a++) soA[a] = (int)(((int64_t)(i0 + 16 * a) * p.lda + kfirst) * 8);
    voffA = (int)(((int64_t)lc * p.lda + 2 * kq) * 8);
    stepA = (KDOWN ? -1 : 1) * GW_CH * 8;
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) soA[q] = (int)(((int64_t)(kfirst + (q >> 1)) * p.lda + i0 + 32 * (q & 1)) * 8);
    voffA = (int)(((int64_t)2 * kq * p.lda + 2 * lc) * 8);
    stepA = (KDOWN ? -1 : 1) * GW_CH * (int)p.lda * 8;
  }
  // B: soff[s] -> row kfirst + s, column j0; lane offset (2 kq * ldb + 2 lc) * 8; column half h = + 256 bytes (immediate)
#pragma unroll
  for (int s = 0; s < 2; s++) soB[s] = (int)(((int64_t)(kfirst + s) * p.ldb + j0) * 8);
  const int voffB = (int)(((int64_t)2 * kq * p.ldb + 2 * lc) * 8);
  const int stepB = (KDOWN ? -1 : 1) * GW_CH * (int)p.ldb * 8;

  int nlA = 0, nlB = 0;                       // chunks requested so far (scalar)
  // raw loads only: nothing here consumes a loaded value.  A request past the last chunk re-reads the last one (L1 hit).
  // op(A) (M x M, L2-resident) is requested ONE chunk ahead into two register sets, the B strip (streamed from HBM) THREE
  // chunks ahead into four: both ring lengths divide the 8 chunks of a diagonal block, so the loop below (4 chunks per
  // trip) meets the diagonal block in a fixed phase and its MFMA-tile skipping is compile-time.
  auto load_A = [&](dbl2 (&Ar)[4]) {
    const int da = (nlA > 0 && nlA < nch) ? stepA : 0;    // (scalar)
#pragma unroll
    for (int q = 0; q < 4; q++) soA[q] += da;
    nlA++;
#pragma unroll
    for (int q = 0; q < 4; q++) Ar[q] = __builtin_bit_cast(dbl2, __builtin_amdgcn_raw_buffer_load_b128(rA, voffA, soA[q], 0));
  };
  auto load_B = [&](dbl2 (&Br)[4]) {
    const int db = (nlB > 0 && nlB < nch) ? stepB : 0;
#pragma unroll
    for (int s = 0; s < 2; s++) soB[s] += db;
    nlB++;
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int hh = 0; hh < 2; hh++)
        Br[2 * s + hh] = __builtin_bit_cast(dbl2, __builtin_amdgcn_raw_buffer_load_b128(rB, voffB + hh * 256, soB[s], 0));
  };

  d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

  // MFMAs of one chunk over row tiles [A_LO, A_HI) (compile-time: straight-line code, no control flow around an MFMA).
  // !TA: Ar[a] = (k-step 0, k-step 1) of row tile a.  TA: Ar[2 s + a'] = row tiles (2 a', 2 a' + 1) of k-step s
  auto mfma_chunk = [&](const dbl2 (&Ar)[4], const dbl2 (&Br)[4], auto lo_tag, auto hi_tag) {
    constexpr int A_LO = decltype(lo_tag)::value, A_HI = decltype(hi_tag)::value;
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int a = A_LO; a < A_HI; a++) {
        const double af = TA ? ((a & 1) ? Ar[2 * s + (a >> 1)].y : Ar[2 * s + (a >> 1)].x) : (s ? Ar[a].y : Ar[a].x);
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const double bf = (b & 1) ? Br[2 * s + (b >> 1)].y : Br[2 * s + (b >> 1)].x;
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[a][b], 0, 0, 0);
        }
      }
  };
  // TAG 2 inside the diagonal block: the strict upper triangle of q_sqrt is a free parameter the reference masks away
  // (matrix_band_part, GPflow conditional) — entries Lq[k][i] with k < i are cleared in the registers
  const int dki = 2 * kq - 2 * lc;            // (k - k0 - s) - (i - i0 - 32 a')
  auto mask_ta = [&](dbl2 (&Ar)[4], const int krel) {      // krel = k0 - i0
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int d = krel + dki + (q >> 1) - 32 * (q & 1);  // k - i for the lane's first row; its second row is one further
      if (d < 0) Ar[q].x = 0.0;
      if (d < 1) Ar[q].y = 0.0;
    }
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 4> c4{};
#define GW_SB __builtin_amdgcn_sched_barrier(0)
  // one chunk: request what comes next into the register sets that are free, then this chunk's 32 (or fewer) MFMAs
#define GW_CHUNK(AR, BR, LO, HI, LOADS) do { LOADS; GW_SB; mfma_chunk(AR, BR, LO, HI); GW_SB; } while (0)
  // chunk D (0 .. 7, in walking order) of the diagonal block of a triangular op(A): which row tiles have anything in it
  //   TAG 1 (k0 = i0 + 8 D): row tile a (rows i0 + 16 a ..+15) is all-zero when k0 > i0 + 16 a + 15 -> tiles [D / 2, 4)
  //   TAG 2 (k0 = i0 + 56 - 8 D, walking down): with the row permutation of TA a tile PAIR spans 32 rows; pair a' (rows
  //     i0 + 32 a' ..+31) is all-zero when k0 + 7 < i0 + 32 a' -> tiles [0, 4) for D < 4, [0, 2) from there on
#define GW_DIAG(AR, BR, D, LOADS) \
  do { LOADS; GW_SB; if (TAG == 2) mask_ta(AR, 56 - 8 * (D)); \
       mfma_chunk(AR, BR, (std::integral_constant<int, (TAG == 1) ? (D) / 2 : 0>{}), \
                  (std::integral_constant<int, (TAG == 1 || (D) < 4) ? 4 : 2>{})); GW_SB; } while (0)
  // "A0 has landed; the loads issued behind it may pend" — stated on both ways into the loop head (left to itself the
  // compiler's counter merge there asks for vmcnt(0), i.e. drains the B prefetch every fourth chunk)
#define GW_WAIT(N) do { GW_SB; __builtin_amdgcn_s_waitcnt(0x0F70 | (N)); GW_SB; } while (0)

  dbl2 A0[4], A1[4], B0[4], B1[4], B2[4], B3[4];
  load_B(B0); load_B(B1); load_A(A0); load_B(B2);
  GW_WAIT(4);
  const int nplain = DENSE ? nch : nch - 8;          // chunks outside the diagonal block: a multiple of 8 (of 4: dense)
  for (int c = 0; c < nplain; c += 4) {
    GW_CHUNK(A0, B0, c0, c4, (load_A(A1), load_B(B3)));
    GW_CHUNK(A1, B1, c0, c4, (load_A(A0), load_B(B0)));
    GW_CHUNK(A0, B2, c0, c4, (load_A(A1), load_B(B1)));
    GW_CHUNK(A1, B3, c0, c4, (load_A(A0), load_B(B2)));
    GW_WAIT(4);
  }
  if (!DENSE) {
    GW_DIAG(A0, B0, 0, (load_A(A1), load_B(B3)));
    GW_DIAG(A1, B1, 1, (load_A(A0), load_B(B0)));
    GW_DIAG(A0, B2, 2, (load_A(A1), load_B(B1)));
    GW_DIAG(A1, B3, 3, (load_A(A0), load_B(B2)));
    GW_DIAG(A0, B0, 4, (load_A(A1), load_B(B3)));
    GW_DIAG(A1, B1, 5, (load_A(A0)));
    GW_DIAG(A0, B2, 6, (load_A(A1)));
    GW_DIAG(A1, B3, 7, (void)0);
  }
#undef GW_DIAG
#undef GW_WAIT
#undef GW_CHUNK

  // ---- epilogue -------------------------------------------------------------------------------------------------------
  // acc[a][b][r] = C(i0 + ROW(a, r), j0 + 32 (b >> 1) + 2 lc + (b & 1)),  ROW = 16 a + 4 r + kq  (TA: the row permutation
  // is irrelevant — role 2 stores nothing and its column sums run over all 64 rows)
  if (DENSE) {
    const gcptr2 gs = (gcptr2)((gcbytes)p.v1 + (int64_t)(j0 + 2 * lc) * 8);
    const dbl2 s0 = gs[0], s1 = gs[16];
    const double sc[4] = {f.alpha * s0.x, f.alpha * s0.y, f.alpha * s1.x, f.alpha * s1.y};
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[a][b][r] *= sc[b];
  }
  if (TAG == 5) {
    // per entry the arithmetic of hyper_contract_kernel<1, false, false, KT> (Stationary.euclid_dist expansion included);
    // the wavefront's own copy of the exp table (LDS operations of one wavefront complete in order: no barrier)
    etab[lane] = exp2((double)lane * (1.0 / 64.0));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const gcptr th = (gcptr)p.kern.theta;
    const double var = th[0], ls = th[1], inv_ls = 1.0 / ls;
    const gcptr gz = (gcptr)p.xa + i0 + kq, gal = (gcptr)p.v0 + i0 + kq;
    const gcptr2 gx = (gcptr2)((gcptr)f.xcols + j0 + 2 * lc), ggm = (gcptr2)((gcptr)p.v2 + j0 + 2 * lc);
    const dbl2 x0 = gx[0], x1 = gx[16], g0 = ggm[0], g1 = ggm[16];
    const double xc[4] = {x0.x, x0.y, x1.x, x1.y}, gmc[4] = {g0.x, g0.y, g1.x, g1.y};
    double acc_v = 0.0, acc_l = 0.0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
      double ra[4], ral[4];
#pragma unroll
      for (int r = 0; r < 4; r++) { ra[r] = gz[16 * a + 4 * r] / ls; ral[r] = gal[16 * a + 4 * r]; }
#pragma unroll
      for (int b = 0; b < 4; b++) {
        __builtin_amdgcn_sched_barrier(0);       // one 16-row tile at a time (the compiler hoists every table read otherwise)
        const double bcol = xc[b] / ls, bb = __dmul_rn(bcol, bcol), gmj = gmc[b];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double av = ra[r], aa = __dmul_rn(av, av);
          const double w = fma(ral[r], gmj, acc[a][b][r]);
          const double r2 = __dadd_rn(__dadd_rn(-2.0 * __dmul_rn(av, bcol), aa), bb);
          if (KT == GP_KERN_RBF) {
            const double e = gp_exp_neg(-0.5 * r2, etab);
            acc_v = fma(w, e, acc_v);
            acc_l = fma(w, var * e * r2 * inv_ls, acc_l);
          } else {
            double rr, rinv;
            gp_sqrt_rsqrt_pos(__dadd_rn(r2, 1e-12), rr, rinv);
            double phi, dphi;
            if (KT == GP_KERN_MATERN12) { phi = gp_exp_neg(-rr, etab); dphi = -phi; }
            else if (KT == GP_KERN_MATERN32) {
              const double s3 = 1.7320508075688772, e = gp_exp_neg(-s3 * rr, etab);
              phi = (1.0 + s3 * rr) * e; dphi = -3.0 * rr * e;
            } else {
              const double s5 = 2.23606797749979, e = gp_exp_neg(-s5 * rr, etab);
              phi = (1.0 + s5 * rr + (5.0 / 3.0) * rr * rr) * e; dphi = -(5.0 / 3.0) * rr * (1.0 + s5 * rr) * e;
            }
            acc_v = fma(w, phi, acc_v);
            acc_l = fma(w * var * dphi, -r2 * rinv * inv_ls, acc_l);
          }
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) { acc_v += __shfl_down(acc_v, o, 64); acc_l += __shfl_down(acc_l, o, 64); }
    if (lane == 0) {
      const gptr out = (gptr)p.o0 + ((int64_t)(i0 / GW_T) * (f.N / GW_T) + j0 / GW_T) * 2;
      out[0] = acc_v; out[1] = acc_l;
    }
    return;
  }
  dbl2 s2[2], sd[2];
  if (f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) {
    // per-column sums over this tile's 64 rows, one partial row per 64-row tile: o0 / o1 [i0 / 64][N]
    double v0r[16];
    if (f.epi & EPI_COLDOT) {
      const gcptr gv0 = (gcptr)p.v0 + i0 + kq;
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) v0r[4 * a + r] = gv0[16 * a + 4 * r];
    }
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      s2[hh] = dbl2{0.0, 0.0}; sd[hh] = dbl2{0.0, 0.0};
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double x0 = acc[a][2 * hh][r], x1 = acc[a][2 * hh + 1][r];
          s2[hh].x = fma(x0, x0, s2[hh].x); s2[hh].y = fma(x1, x1, s2[hh].y);
          if (f.epi & EPI_COLDOT) { sd[hh].x = fma(x0, v0r[4 * a + r], sd[hh].x); sd[hh].y = fma(x1, v0r[4 * a + r], sd[hh].y); }
        }
      s2[hh].x += __shfl_xor(s2[hh].x, 16, 64); s2[hh].x += __shfl_xor(s2[hh].x, 32, 64);
      s2[hh].y += __shfl_xor(s2[hh].y, 16, 64); s2[hh].y += __shfl_xor(s2[hh].y, 32, 64);
      if (f.epi & EPI_COLDOT) {
        sd[hh].x += __shfl_xor(sd[hh].x, 16, 64); sd[hh].x += __shfl_xor(sd[hh].x, 32, 64);
        sd[hh].y += __shfl_xor(sd[hh].y, 16, 64); sd[hh].y += __shfl_xor(sd[hh].y, 32, 64);
      }
    }
  }
  // STORES LAST, and no vector instruction behind a store writes a register the store reads.  A 16-byte buffer store with
  // a scalar offset reads its data registers well after it has issued; the compiler pads FLAT stores and buffer stores
  // WITHOUT scalar offset with wait states and — following the SI-era rule — not this form, and it refills one temporary
  // register quad for store after store.  Measured on MI355X with exact integers (tools/probe_wave.hip): lanes 12-15 /
  // 44-47 of a store now and then carried the NEXT store's low dwords (the neighbouring column's: 1e-7 relative in a
  // smooth strip), 300-800 torn values per 67 M stored, more when other wavefronts compete for the SIMD; s_nop 3 behind
  // every store did not cure it.  So the 32 payloads (the two column parities of a row side by side) are built into 32
  // DIFFERENT register quads — the ring's registers and those of the accumulators already copied — kept alive by the
  // empty asm below, and the wavefront waits for its stores before its next tile (or the next wavefront) reuses them.
  if (f.epi & EPI_STORE) {
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0x7fffffff, 0x00020000);
    const int voffC = (int)(((int64_t)kq * p.ldc + 2 * lc) * 8);
    dbl2 pay[4][4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          const d4 t0 = acc[a][2 * hh], t1 = acc[a][2 * hh + 1];
          pay[a][r][hh].x = (r == 0) ? t0.x : (r == 1) ? t0.y : (r == 2) ? t0.z : t0.w;
          pay[a][r][hh].y = (r == 0) ? t1.x : (r == 1) ? t1.y : (r == 2) ? t1.z : t1.w;
        }
    GW_SB;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int soC = (int)(((int64_t)(i0 + 16 * a + 4 * r) * p.ldc + j0) * 8);      // (scalar)
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gw_u4, pay[a][r][hh]), rC, voffC + hh * 256, soC, 0);
      }
    GW_SB;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) asm volatile("" :: "v"(pay[a][r][0]), "v"(pay[a][r][1]));
  }
  if ((f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) && kq == 0) {
    const int64_t prow = (int64_t)(i0 / GW_T) * f.N + j0 + 2 * lc;
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      if (f.epi & EPI_COLSUMSQ) *(gptr2)((gptr)p.o0 + prow + 32 * hh) = s2[hh];
      if (f.epi & EPI_COLDOT) *(gptr2)((gptr)p.o1 + prow + 32 * hh) = sd[hh];
    }
  }
  asm volatile("" :: "v"(s2[0]), "v"(s2[1]), "v"(sd[0]), "v"(sd[1]));
  // the tile's stores have read their registers (and landed) before anything — the next tile of this wavefront — writes them
  GW_SB;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  GW_SB;
#undef GW_SB
}

template <int TAG, int KT = -1>
__global__ void __launch_bounds__(256, 2) gemm_wave_kernel(const GemmProblem* __restrict__ probs, WaveFlags f) {
  __shared__ double etabs[(TAG == 5) ? 4 * 64 : 1];
  // XCD-aware renumbering of the flattened (unit, batch) grid (gemm_strip.hip): blocks b and b + 8 share an XCD and its L2
  int bid = blockIdx.x, bz = blockIdx.z;
  {
    const int nx = gridDim.x, total = nx * (int)gridDim.z;
    if ((total & 7) == 0) {
      const int lin = bz * nx + bid;
      const int log = (lin & 7) * (total >> 3) + (lin >> 3);
      bz = log / nx; bid = log - bz * nx;
    }
  }
  const GemmProblem p = probs[bz];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // units of a 256-column group adjacent (they share the group's B strip in the XCD's L2), rotated over the shader engines
  const int cg = bid / f.nunits;
  const int u = (bid % f.nunits + cg) % f.nunits;
  const int j0 = cg * 256 + wv * GW_T;
  if (TAG == 3 || TAG == 5) {
    gw_tile<TAG, KT>(p, f, (f.t0 + u) * GW_T, j0, lane, etabs + ((TAG == 5) ? 64 * wv : 0));
  } else {
    // pair (t0 + u, t1 - 1 - u): the longer K range first (every pair starts on the strip's common end: k = 0 for the
    // lower-triangular product, k = K for the upper one), then the shorter one
    const int ta = f.t0 + u, tb = f.t1 - 1 - u;
    const int first = (TAG == 1) ? tb : ta, second = (TAG == 1) ? ta : tb;
    const int npass = (second != first) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; pass++)          // (one copy of the tile body)
      gw_tile<TAG>(p, f, __builtin_amdgcn_readfirstlane((pass ? second : first) * GW_T), j0, lane);
  }
}

static bool gw_enabled() { return gp_switches().strip_wave != 0; }
static int gw_roles() { return gp_switches().strip_wave_roles; }      // bitmask 1 << role (switches.h)

// would a launch of that role and shape take the wave form?  (engine.hip sizes the partial-sum rows by it: 64-row tiles)
bool gemm_wave_takes(int role, int maxM, int maxN, int uniform_aligned) {
  if (!gw_enabled() || !uniform_aligned || role < 1 || (role > 3 && role != 5) || !((gw_roles() >> role) & 1)) return false;
  return maxM > 0 && (maxM % GW_T) == 0 && (maxN % 256) == 0;
}

template <int TAG, int KT = -1>
static gp_status gw_launch(gp_handle h, const GemmProblem* d_probs, int batch, int M, int N, const GemmFlags& f) {
  WaveFlags wf;
  const int tiles = M / GW_T;
  wf.t0 = 0; wf.t1 = tiles;
  if (f.tile_m0 > 0 || f.tile_mcount > 0) {      // row-blocks of 128 = pairs of 64-row tiles
    wf.t0 = 2 * f.tile_m0;
    if (wf.t0 >= tiles) return GP_OK;
    if (f.tile_mcount > 0 && 2 * (f.tile_m0 + f.tile_mcount) < tiles) wf.t1 = 2 * (f.tile_m0 + f.tile_mcount);
  }
  const int nt = wf.t1 - wf.t0;
  wf.nunits = (TAG == 3 || TAG == 5) ? nt : (nt + 1) / 2;
  wf.epi = f.epilogue; wf.alpha = f.alpha; wf.N = N; wf.pad_ = 0; wf.xcols = f.aux_x;
  dim3 grid(wf.nunits * (N / 256), 1, batch);
  hipLaunchKernelGGL((gemm_wave_kernel<TAG, KT>), grid, dim3(256), 0, h->stream, d_probs, wf);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// Returns true when the wave form took the launch (*st = its status).  `uniform_aligned` = the caller vouches that every
// problem of the batch has M = K-structure maxM, N = maxN, 16-byte aligned operands, even leading dimensions.
bool launch_gemm_wave(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f, gp_status* st) {
  if (!gemm_wave_takes(f.role, maxM, maxN, f.uniform_aligned)) return false;
  if (f.beta != 0.0 || f.triC != TRI_NONE) return false;
  if (f.role >= 3 ? !(f.scale_mode == 1 && (f.alpha == 1.0 || f.alpha == 2.0 || f.alpha == 0.5 || f.alpha == 4.0)) : (f.alpha != 1.0)) return false;
  if (f.role == 5) {       // (the caller has asked gemm_fused_contraction_records first: it cannot fall back from here)
    if (!f.aux_x || f.tile_m0 || f.tile_mcount) { *st = gp_fail(h, GP_ERR_BAD_ARG, "fused Kuf_bar contraction: bad launch"); return true; }
    switch (f.aux_ktype) {
      case GP_KERN_MATERN12: *st = gw_launch<5, GP_KERN_MATERN12>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_MATERN32: *st = gw_launch<5, GP_KERN_MATERN32>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_MATERN52: *st = gw_launch<5, GP_KERN_MATERN52>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_RBF: *st = gw_launch<5, GP_KERN_RBF>(h, d_probs, batch, maxM, maxN, f); break;
      default: *st = gp_fail(h, GP_ERR_BAD_ARG, "fused Kuf_bar contraction: not a stationary kernel");
    }
    return true;
  }
  if (f.role == 1) *st = gw_launch<1>(h, d_probs, batch, maxM, maxN, f);
  else if (f.role == 2) *st = gw_launch<2>(h, d_probs, batch, maxM, maxN, f);
  else *st = gw_launch<3>(h, d_probs, batch, maxM, maxN, f);
  return true;
}

// partial records (2 sums each) the fused Kuf_bar contraction of a float64 family leaves per latent GP — 0: no fused form
// for that shape / kernel type (the separate contraction kernel runs).  One per wavefront tile here, one per 128 x 128 tile
// in gemm_strip.hip's form.
int gemm_fused_contraction_records(int maxM, int maxN, int ktype) {
  if (!gemm_strip_fused_contraction_ok(maxM, maxN, ktype)) return 0;
  if (gemm_wave_takes(5, maxM, maxN, 1)) return (maxM / GW_T) * (maxN / GW_T);
  return (maxM / 128) * (maxN / 128);
}
