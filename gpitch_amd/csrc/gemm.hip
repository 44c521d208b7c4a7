// gemm.hip — batched fp64 matrix-core GEMM family for the sparse-GP conditional (gfx950).
//
// Replaces the dense TF ops inside GPflow `conditional` and `SGPR` as called from
// gpitch/pdgp.py:147-155 and gpitch/sgpr_ss.py:48-53:
//   tf.matrix_triangular_solve(Lm, Kmn)   ->  A   = W * Kuf        (W = Lm^-1 lower, tri-aware)
//   tf.matmul(L_q^T, A)                   ->  LTA = Lq^T * A       (upper-from-lower, tri-aware)
//   reduce_sum(square(.), 0), A^T q_mu    ->  fused column reductions in the epilogue
//   tf.matmul(A, A, transpose_b=True)     ->  split-K "NT" product over the frame dimension
// plus every M x M x M product of the backward pass (Cholesky adjoint etc.).
//
// Design for CDNA4: 256-thread workgroups (4 wavefronts: 1 x 4 for the 128-tiles, 2 x 2 for the 64-tiles),
// v_mfma_f64_16x16x4_f64, 128 x 128 (strip) or 64 x 64 (small) output tiles, BK = 16, operands staged global -> registers ->
// LDS with one barrier per K-tile (double-buffered LDS, next tile's global loads in flight during the
// MFMAs), LDS strides padded so that every fragment read is bank-conflict free for ds_read_b64
// (k-contiguous tiles use the odd stride BK+1, row-contiguous tiles use stride B+16).
// Triangular operands skip whole K-tiles that are structurally zero and mask inside the diagonal
// tile; workgroups are renumbered so that the row-blocks sharing one operand strip sit on the same
// XCD (shared L2).  Two workgroups are resident per CU (<= 256 VGPRs, 2 x 72 KiB LDS).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <atomic>

typedef double d4 __attribute__((ext_vector_type(4)));
// Pointers fetched from the problem descriptor are generic to the compiler, which then emits flat_load:
// FLAT counts on lgkmcnt as well as vmcnt, so every LDS wait in the K loop would also wait for the
// in-flight operand prefetch.  Typing them as address-space-1 gives global_load / global_store.
typedef const double __attribute__((address_space(1))) * gcptr;
typedef double __attribute__((address_space(1))) * gptr;
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef const dbl2 __attribute__((address_space(1))) * gcptr2;

// Threads per workgroup: 4 wavefronts (1 x 4 on the 128-tiles, each 128 x 32; 2 x 2 on the 64-tiles), except the
// Lq^T A strip product (TAG 2), which measures 5 % faster with 8 (2 x 4, each 64 x 32, <= 128 VGPRs: four
// wavefronts per SIMD); the other roles lose more to the extra LDS reads and staging than they gain.
#ifndef GP_GEMM_NOSKIP
#define GP_GEMM_NOSKIP 0
#endif
#ifndef GP_GEMM_NT_STORE
#define GP_GEMM_NT_STORE 0
#endif
#ifndef GP_GEMM_8W_TAG
#define GP_GEMM_8W_TAG 2      // a second role to build with 8 wavefronts (same-box A/B builds only)
#endif
constexpr int gemm_threads(int BM, int TAG) { return (BM == 128 && (TAG == 2 || TAG == GP_GEMM_8W_TAG)) ? 512 : 256; }
#define GEMM_BK 16
#ifndef GP_MFMA_PRIO
#define GP_MFMA_PRIO 2
#endif

struct GemmDevFlags {
  int triA, triB, triC;
  double alpha, beta;
  int epi;         // bitmask: 1 store, 2 colsumsq -> o0, 4 coldot(v0) -> o1
  int scale_mode;  // 0 none, 1: B(k,n) *= v1[n], 2: B(k,n) *= v1[k]
  int ksplit;      // >1: split-K, slabs written to o2 + s*M*N (ldc = N), epi forced to plain store
  int tilesM, tilesN;
  int tm0;         // first row-block of this launch (tilesM counts the launch's row-blocks)
  int tilesM_req;  // host side only: requested number of row-blocks (0 = all)
};

template <int BM, int BN, bool TA, bool TB>
struct GemmSmem {
  // k-contiguous tiles: odd stride (BK+1) so the 16-lane groups of the fused ds_read2_b64 fragment reads
  // (32-bank rule) hit 16 distinct bank pairs; row-contiguous tiles: consecutive lanes read consecutive doubles.
  static constexpr int SA = TA ? (BM + 16) : (GEMM_BK + 1);
  static constexpr int SB = TB ? (GEMM_BK + 1) : (BN + 16);
  static constexpr int A_ELEMS = TA ? GEMM_BK * SA : BM * SA;
  static constexpr int B_ELEMS = TB ? BN * SB : GEMM_BK * SB;
  static constexpr int STAGE = A_ELEMS + B_ELEMS;
  static constexpr size_t BYTES = (size_t)2 * STAGE * sizeof(double);
};

// TAG names the instantiation (one symbol per role, so profiles attribute time to the right product) and
// fixes that role's operand structure at compile time:
//   0 generic (structure from runtime flags), 1 cond_A (A = W Kuf, W lower), 2 cond_LTA (Lq^T A, upper),
//   3 kuf_bar (R (A D), columns of B scaled by v1[n]), 4 nt (X [D] Y^T split-K, B optionally scaled by v1[k])
template <int TAG> struct RoleCfg { static constexpr int triA = -1, triB = -1, scale = -1; };
template <> struct RoleCfg<1> { static constexpr int triA = TRI_LOWER, triB = TRI_NONE, scale = 0; };
template <> struct RoleCfg<2> { static constexpr int triA = TRI_UPPER, triB = TRI_NONE, scale = 0; };
template <> struct RoleCfg<3> { static constexpr int triA = TRI_NONE, triB = TRI_NONE, scale = 1; };
template <> struct RoleCfg<4> { static constexpr int triA = TRI_NONE, triB = TRI_NONE, scale = -1; };  // scale: runtime (v1 may be absent)

template <int BM, int BN, bool TA, bool TB, int TAG>
__global__ void __launch_bounds__(gemm_threads(BM, TAG), gemm_threads(BM, TAG) / 128) gemm_f64_kernel(const GemmProblem* __restrict__ probs,
                                                                    GemmDevFlags f) {
  using S = GemmSmem<BM, BN, TA, TB>;
  // wave layout: 128-tiles use 1 x 4 (each wave owns all 128 rows of a 32-column slice, so triangular
  // skipping inside the diagonal block is identical for every wave: no barrier imbalance), or 2 x 4 with 8
  // wavefronts; 64-tiles 2 x 2.
  constexpr int GEMM_THREADS = gemm_threads(BM, TAG);
  constexpr int WAVES_M = (BM == 128) ? GEMM_THREADS / 256 : 2, WAVES_N = (GEMM_THREADS / 64) / WAVES_M;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int EA = BM * GEMM_BK / GEMM_THREADS;  // elements per thread, A tile
  constexpr int EB = BN * GEMM_BK / GEMM_THREADS;
  extern __shared__ __attribute__((aligned(16))) double smem[];

  const int triA = RoleCfg<TAG>::triA >= 0 ? RoleCfg<TAG>::triA : f.triA;
  const int triB = RoleCfg<TAG>::triB >= 0 ? RoleCfg<TAG>::triB : f.triB;
  const int scale_mode = RoleCfg<TAG>::scale >= 0 ? RoleCfg<TAG>::scale : f.scale_mode;

  // XCD-aware renumbering.  The hardware deals workgroups to the 8 XCDs round-robin in dispatch order (x fastest, then
  // the batch index z), so blocks b and b + 8 of the FLATTENED grid share an XCD (and its L2).  Each XCD is given a
  // contiguous range of logical (tile, batch) indices.  The split-K grid (tiles x slices = 90 per GP at the bench
  // shape) is not a multiple of 8 per batch entry, and renumbering x alone left its ten tiles of a K-slice on eight
  // different L2s: every tile fetched both its operand strips from HBM (13 GB per launch for 3.2 GB of operands).
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
  const gcptr gA = (gcptr)p.A, gB = (gcptr)p.B, gv0 = (gcptr)p.v0, gv1 = (gcptr)p.v1;
  const gptr gC = (gptr)p.C, go0 = (gptr)p.o0, go1 = (gptr)p.o1, go2 = (gptr)p.o2;
  int tm, tn, ksl = 0;
  if (f.ksplit > 1) {
    // split-K.  With a symmetric (lower) output only the tilesM(tilesM+1)/2 tiles on or below the diagonal are
    // enumerated, so every launched workgroup has work.  Inside an XCD's range the OUTPUT TILE varies fastest, so the
    // workgroups resident together on one XCD work on the same K-slice of different tiles and share its operand
    // strips in that XCD's L2 (each strip is used by tilesM+1 tiles).
    const int ntl = (f.triC == TRI_LOWER) ? f.tilesM * (f.tilesM + 1) / 2 : f.tilesM * f.tilesN;
    ksl = bid / ntl;
    const int t = bid % ntl;
    if (f.triC == TRI_LOWER) {
      tm = (int)((__dsqrt_rn(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
      while (tm * (tm + 1) / 2 > t) tm--;
      tn = t - tm * (tm + 1) / 2;
    } else {
      tm = t % f.tilesM; tn = t / f.tilesM;
    }
  } else {
    // (XCD-contiguous logical order from above: the row-blocks that read the same B strip hit the same L2)
    // Row-blocks of a triangular product differ ~7x in work and the hardware deals consecutive workgroups
    // to the shader engines round-robin, so a fixed (bid % tilesM) would pin all the heavy row-blocks on
    // one engine; rotating by the strip index keeps a strip's row-blocks adjacent but cycles who gets which.
    tn = bid / f.tilesM;
    tm = f.tm0 + (bid % f.tilesM + tn) % f.tilesM;
  }
  const int i0 = tm * BM, j0 = tn * BN;
  if (i0 >= p.M || j0 >= p.N) return;
  if (f.triC == TRI_LOWER && j0 > i0 + BM - 1) return;

  int kbeg = 0, kend = p.K;
  if (triA == TRI_LOWER) kend = min(kend, i0 + BM);
  if (triA == TRI_UPPER) kbeg = max(kbeg, i0);
  if (triB == TRI_LOWER) kbeg = max(kbeg, j0);
  if (triB == TRI_UPPER) kend = min(kend, j0 + BN);
  kbeg = (kbeg / GEMM_BK) * GEMM_BK;
  if (f.ksplit > 1) {
    int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
    int per = (nk + f.ksplit - 1) / f.ksplit;
    int b0 = kbeg + ksl * per * GEMM_BK;
    kend = min(kend, b0 + per * GEMM_BK);
    kbeg = b0;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WAVES_N, wc = wave % WAVES_N;
  const int lc = lane & 15, kq = lane >> 4;
  // Column tiles (16 wide) owned by this wave.  The symmetric split-K product (TAG 4) pairs tile wc with tile
  // 7 - wc: on a diagonal output tile only the MFMA tiles with row-tile >= column-tile are needed, and this
  // pairing gives every wave the same number of them (9 of 16), so the skip costs no barrier imbalance.
  constexpr bool PERM = (TAG == 4 && TN == 2 && WAVES_N == 4);
  int ctile[TN];
#pragma unroll
  for (int b = 0; b < TN; b++) ctile[b] = PERM ? (b == 0 ? wc : 2 * WAVES_N - 1 - wc) : (wc * TN + b);
  const bool diag_sym = PERM && (f.triC == TRI_LOWER) && (tm == tn);
  // first row-tile each of this wave's column tiles needs (wave-uniform scalars; 0 = everything)
  int cmin[TN];
#pragma unroll
  for (int b = 0; b < TN; b++)
    cmin[b] = diag_sym ? max(0, __builtin_amdgcn_readfirstlane(ctile[b]) - __builtin_amdgcn_readfirstlane(wr) * TM) : 0;

  // ---- global -> register staging ---------------------------------------------------------------
  // Row-contiguous tiles (op(A) transposed / op(B) plain) are dealt to threads in 16-byte pairs so that
  // consecutive lanes hold consecutive pairs: coalesced 256-B row segments per load and conflict-free
  // ds_write_b128.  k-contiguous tiles give each thread EA (EB) consecutive k of one row.
  double ra[EA], rb[EB], rs[EB];
  // split-K product only: the tiles of the first tile-column also form u = X v2 (row dot products with a
  // frame vector) from the A tiles they stage anyway — one pass over X instead of two (bwd: u = A gm).
  const bool rowdot = (TAG == 4) && !TA && (p.v2 != nullptr) && (tn == 0);
  const gcptr gv2 = (gcptr)p.v2;
  double udot = 0.0, rg[EA];
  int a_i, a_k;
  constexpr int TPR_A = TA ? (BM / EA) : (GEMM_BK / EA);
  if (TA) { a_k = tid / TPR_A; a_i = (tid % TPR_A) * 2; }   // pairs at a_i + e/2 * (2*TPR_A)
  else    { a_i = tid / TPR_A; a_k = (tid % TPR_A) * EA; }
  int b_k, b_n;
  constexpr int TPR_B = TB ? (GEMM_BK / EB) : (BN / EB);
  if (TB) { b_n = tid / TPR_B; b_k = (tid % TPR_B) * EB; }
  else    { b_k = tid / TPR_B; b_n = (tid % TPR_B) * 2; }   // pairs at b_n + e/2 * (2*TPR_B)
  const bool a_vec = ((p.lda & 1) == 0) && ((((uintptr_t)p.A) & 15) == 0);
  const bool b_vec = ((p.ldb & 1) == 0) && ((((uintptr_t)p.B) & 15) == 0);
  auto a_off = [&](int e) { return TA ? (a_i + (e >> 1) * (2 * TPR_A) + (e & 1)) : (a_k + e); };  // tile-local i (TA) or k
  auto b_off = [&](int e) { return TB ? (b_k + e) : (b_n + (e >> 1) * (2 * TPR_B) + (e & 1)); };  // tile-local k (TB) or n

  // per-column scale is invariant over the K loop: fetch it once
  if (scale_mode == 1) {
#pragma unroll
    for (int e = 0; e < EB; e++) {
      const int n = j0 + (TB ? b_n : b_off(e));
      rs[e] = (n < p.N) ? gv1[n] : 0.0;
    }
  }

  // raw loads only: nothing here may consume a loaded value (the waits must sit after the MFMAs)
  auto load_tiles = [&](int kt, auto fast_tag) {
    if constexpr (decltype(fast_tag)::value) {
      // interior tile, aligned operands, whole K-tiles: unconditional 16-byte loads, no exec-mask regions
      if (TA) {
        const gcptr src = gA + (int64_t)(kt + a_k) * p.lda + i0;
#pragma unroll
        for (int e = 0; e < EA; e += 2) { dbl2 v = *(gcptr2)(src + a_off(e)); ra[e] = v.x; ra[e + 1] = v.y; }
      } else {
        const gcptr src = gA + (int64_t)(i0 + a_i) * p.lda + kt + a_k;
#pragma unroll
        for (int e = 0; e < EA; e += 2) { dbl2 v = *(gcptr2)(src + e); ra[e] = v.x; ra[e + 1] = v.y; }
        if (TAG == 4 && rowdot) {
#pragma unroll
          for (int e = 0; e < EA; e++) rg[e] = gv2[kt + a_k + e];
        }
      }
      if (TB) {
        const gcptr src = gB + (int64_t)(j0 + b_n) * p.ldb + kt + b_k;
#pragma unroll
        for (int e = 0; e < EB; e += 2) { dbl2 v = *(gcptr2)(src + e); rb[e] = v.x; rb[e + 1] = v.y; }
        if (scale_mode == 2) {
#pragma unroll
          for (int e = 0; e < EB; e++) rs[e] = gv1[kt + b_k + e];
        }
      } else {
        const gcptr src = gB + (int64_t)(kt + b_k) * p.ldb + j0;
#pragma unroll
        for (int e = 0; e < EB; e += 2) { dbl2 v = *(gcptr2)(src + b_off(e)); rb[e] = v.x; rb[e + 1] = v.y; }
        if (scale_mode == 2) {
          const double sv = gv1[kt + b_k];
#pragma unroll
          for (int e = 0; e < EB; e++) rs[e] = sv;
        }
      }
      return;
    }
    if (TA) {
      const int k = kt + a_k;
#pragma unroll
      for (int e = 0; e < EA; e += 2) {
        const int i = i0 + a_off(e);
        const gcptr src = gA + (int64_t)k * p.lda + i;
        if (k < kend && i + 1 < p.M && a_vec) { dbl2 v = *(gcptr2)(src); ra[e] = v.x; ra[e + 1] = v.y; }
        else { ra[e] = (k < kend && i < p.M) ? src[0] : 0.0; ra[e + 1] = (k < kend && i + 1 < p.M) ? src[1] : 0.0; }
      }
    } else {
      const int i = i0 + a_i, k = kt + a_k;
      const gcptr src = gA + (int64_t)i * p.lda + k;
      if (i < p.M && k + EA <= kend && a_vec) {
#pragma unroll
        for (int e = 0; e < EA; e += 2) { dbl2 v = *(gcptr2)(src + e); ra[e] = v.x; ra[e + 1] = v.y; }
      } else {
#pragma unroll
        for (int e = 0; e < EA; e++) ra[e] = (i < p.M && k + e < kend) ? src[e] : 0.0;
      }
      if (TAG == 4 && rowdot) {
#pragma unroll
        for (int e = 0; e < EA; e++) rg[e] = (k + e < kend) ? gv2[k + e] : 0.0;
      }
    }
    if (TB) {
      const int n = j0 + b_n, k = kt + b_k;
      const gcptr src = gB + (int64_t)n * p.ldb + k;
      if (n < p.N && k + EB <= kend && b_vec) {
#pragma unroll
        for (int e = 0; e < EB; e += 2) { dbl2 v = *(gcptr2)(src + e); rb[e] = v.x; rb[e + 1] = v.y; }
      } else {
#pragma unroll
        for (int e = 0; e < EB; e++) rb[e] = (n < p.N && k + e < kend) ? src[e] : 0.0;
      }
      if (scale_mode == 2) {
#pragma unroll
        for (int e = 0; e < EB; e++) rs[e] = (k + e < kend) ? gv1[k + e] : 0.0;
      }
    } else {
      const int k = kt + b_k;
#pragma unroll
      for (int e = 0; e < EB; e += 2) {
        const int n = j0 + b_off(e);
        const gcptr src = gB + (int64_t)k * p.ldb + n;
        if (k < kend && n + 1 < p.N && b_vec) { dbl2 v = *(gcptr2)(src); rb[e] = v.x; rb[e + 1] = v.y; }
        else { rb[e] = (k < kend && n < p.N) ? src[0] : 0.0; rb[e + 1] = (k < kend && n + 1 < p.N) ? src[1] : 0.0; }
      }
      if (scale_mode == 2) {
        const double sv = (k < kend) ? gv1[k] : 0.0;
#pragma unroll
        for (int e = 0; e < EB; e++) rs[e] = sv;
      }
    }
  };

  // structural masks + scaling, then registers -> LDS (runs after the MFMAs of the current tile)
  auto store_tiles = [&](int buf, int kt) {
    double* As = smem + buf * S::STAGE;
    double* Bs = As + S::A_ELEMS;
    if (triA != TRI_NONE) {
#pragma unroll
      for (int e = 0; e < EA; e++) {
        const int i = i0 + (TA ? a_off(e) : a_i), k = kt + (TA ? a_k : a_k + e);
        const bool z = (triA == TRI_LOWER) ? (k > i) : (k < i);
        if (z) ra[e] = 0.0;
      }
    }
    if (triB != TRI_NONE) {
#pragma unroll
      for (int e = 0; e < EB; e++) {
        const int n = j0 + (TB ? b_n : b_off(e)), k = kt + (TB ? b_k + e : b_k);
        const bool z = (triB == TRI_LOWER) ? (n > k) : (k > n);
        if (z) rb[e] = 0.0;
      }
    }
    if (scale_mode != 0) {
#pragma unroll
      for (int e = 0; e < EB; e++) rb[e] *= rs[e];
    }
    if (TAG == 4 && rowdot) {
#pragma unroll
      for (int e = 0; e < EA; e++) udot = fma(ra[e], rg[e], udot);
    }
    if (TA) {
#pragma unroll
      for (int e = 0; e < EA; e += 2)
        *reinterpret_cast<double2*>(As + a_k * S::SA + a_off(e)) = make_double2(ra[e], ra[e + 1]);
    } else {
#pragma unroll
      for (int e = 0; e < EA; e++) As[a_i * S::SA + a_k + e] = ra[e];   // odd stride: 8-byte stores
    }
    if (TB) {
#pragma unroll
      for (int e = 0; e < EB; e++) Bs[b_n * S::SB + b_k + e] = rb[e];
    } else {
#pragma unroll
      for (int e = 0; e < EB; e += 2)
        *reinterpret_cast<double2*>(Bs + b_k * S::SB + b_off(e)) = make_double2(rb[e], rb[e + 1]);
    }
  };

  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

  const int rowbase = i0 + wr * WM;
  // K order.  Lq^T A (TAG 2, upper-triangular op(A)): row-block r needs k in [128 r, M).  Walking k UPWARDS, the four
  // row-blocks of a column strip start at four different places and never meet: each fetched its own copy of the B
  // strip (8.5 GB per launch for 3.2 GB of operands, exactly (4+3+2+1)/4 x).  Walking DOWN from the common end they
  // run in step through the part they share and the strip leaves HBM once.
  constexpr bool KDOWN = (TAG == 2);
  auto run_k_loop = [&](auto fast_tag) {
    const int nkt = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
    const int kfirst = KDOWN ? kbeg + (nkt - 1) * GEMM_BK : kbeg;
    const int kstep = KDOWN ? -GEMM_BK : GEMM_BK;
    load_tiles(kfirst, fast_tag);
    store_tiles(0, kfirst);
    __syncthreads();
    int buf = 0;
    for (int it = 0, kt = kfirst; it < nkt; it++, kt += kstep) {
      const bool more = (it + 1 < nkt);
      if (more) load_tiles(kt + kstep, fast_tag);
      const double* As = smem + buf * S::STAGE;
      const double* Bs = As + S::A_ELEMS;
      // K-tiles strictly inside the non-zero part of a triangular op(A) run the straight-line body; only
      // the K-tiles that cross the diagonal pay for the (wave-uniform) per-tile tests.
      bool full = true;
      if (triA == TRI_LOWER) full = (kt <= rowbase);
      if (triA == TRI_UPPER) full = (kt >= rowbase + 16 * TM - 16);
      if (GP_GEMM_NOSKIP) full = true;       // experiment: no skipping inside the diagonal block (operands are masked)
      if (full) {
        __builtin_amdgcn_s_setprio(GP_MFMA_PRIO);
#pragma unroll
        for (int ks = 0; ks < GEMM_BK / 4; ks++) {
          const int k = ks * 4 + kq;
          double af[TM], bf[TN];
#pragma unroll
          for (int b = 0; b < TN; b++) {
            const int n = ctile[b] * 16 + lc;
            bf[b] = TB ? Bs[n * S::SB + k] : Bs[k * S::SB + n];
          }
#pragma unroll
          for (int a = 0; a < TM; a++) {
            const int i = wr * WM + a * 16 + lc;
            af[a] = TA ? As[k * S::SA + i] : As[i * S::SA + k];
          }
#pragma unroll
          for (int a = 0; a < TM; a++)
#pragma unroll
            for (int b = 0; b < TN; b++)
              if (!PERM || a >= cmin[b])   // symmetric diagonal tile: upper MFMA tiles are never read
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
      } else {
        __builtin_amdgcn_s_setprio(GP_MFMA_PRIO);
#pragma unroll
        for (int ks = 0; ks < GEMM_BK / 4; ks++) {
          const int k = ks * 4 + kq;
          const int kg = kt + ks * 4;  // first global k of this MFMA step
          // 16-row MFMA tiles of op(A) that are structurally zero for this k-step are skipped (wave-uniform)
          int a_lo = 0, a_hi = TM;
          if (triA == TRI_LOWER) a_lo = max(0, (kg - rowbase) >> 4);             // need rowbase+16a+15 >= kg
          if (triA == TRI_UPPER) a_hi = min(TM, ((kg + 3 - rowbase) >> 4) + 1);  // need rowbase+16a <= kg+3
          double af[TM], bf[TN];
#pragma unroll
          for (int b = 0; b < TN; b++) {
            const int n = ctile[b] * 16 + lc;
            bf[b] = TB ? Bs[n * S::SB + k] : Bs[k * S::SB + n];
          }
#pragma unroll
          for (int a = 0; a < TM; a++) {
            const int i = wr * WM + a * 16 + lc;
            af[a] = TA ? As[k * S::SA + i] : As[i * S::SA + k];
          }
#pragma unroll
          for (int a = 0; a < TM; a++) {
            if (a >= a_lo && a < a_hi) {
#pragma unroll
              for (int b = 0; b < TN; b++)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
      if (more) store_tiles(buf ^ 1, kt + kstep);
      __syncthreads();
      buf ^= 1;
    }
  };
  if (kbeg < kend) {
    // wave-uniform: every operand access of this tile is in range and 16-byte aligned
    const bool fastpath = a_vec && b_vec && (i0 + BM <= p.M) && (j0 + BN <= p.N) && (((kend - kbeg) % GEMM_BK) == 0);
    if (fastpath) run_k_loop(std::true_type{}); else run_k_loop(std::false_type{});
  }

  // ---- epilogue -------------------------------------------------------------------------------
  if (f.ksplit > 1) {
    gptr slab = go2 + (int64_t)ksl * p.M * p.N;
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = rowbase + a * 16 + kq + 4 * r, j = j0 + ctile[b] * 16 + lc;
          if (i < p.M && j < p.N) slab[(int64_t)i * p.N + j] = acc[a][b][r];
        }
    if (TAG == 4 && rowdot) {
      // the GEMM_BK / EA threads that share a row hold disjoint k-ranges: combine, then one partial per K-slice
      constexpr int TPR = GEMM_BK / EA;
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) udot += __shfl_xor(udot, o, 64);
      if ((tid % TPR) == 0 && i0 + a_i < p.M) go1[(int64_t)ksl * p.M + i0 + a_i] = udot;
    }
    return;
  }
  if (f.epi & 1) {
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = rowbase + a * 16 + kq + 4 * r, j = j0 + ctile[b] * 16 + lc;
          if (i < p.M && j < p.N) {
            double v = f.alpha * acc[a][b][r];
            gptr c = gC + (int64_t)i * p.ldc + j;
            if (f.beta != 0.0) v += f.beta * (*c);
            if (f.triC == TRI_LOWER && j > i) v = 0.0;
            if (GP_GEMM_NT_STORE && (TAG == 1 || TAG == 3)) __builtin_nontemporal_store(v, c);   // strips: read again only after the whole launch
            else *c = v;
          }
        }
  }
  if (f.epi & 6) {
    // per-column reductions over this tile's rows: sum acc^2 and sum acc * v0[row]
    double* red = smem;  // [2 kinds][WAVES_M][BN]  (only needed when two wave-rows share a column)
    if (WAVES_M > 1) __syncthreads();  // all waves are past their last LDS read
#pragma unroll
    for (int b = 0; b < TN; b++) {
      double s2 = 0.0, sd = 0.0;
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = rowbase + a * 16 + kq + 4 * r;
          const double v = f.alpha * acc[a][b][r];  // rows >= M hold exact zeros
          s2 = fma(v, v, s2);
          if (f.epi & 4) sd = fma(v, (i < p.M) ? gv0[i] : 0.0, sd);
        }
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
      const int n = ctile[b] * 16 + lc;
      if (WAVES_M == 1) {
        const int j = j0 + n;
        if (kq == 0 && j < p.N) {
          if (f.epi & 2) go0[(int64_t)tm * p.N + j] = s2;
          if (f.epi & 4) go1[(int64_t)tm * p.N + j] = sd;
        }
      } else if (kq == 0) {
        red[(0 * WAVES_M + wr) * BN + n] = s2;
        red[(1 * WAVES_M + wr) * BN + n] = sd;
      }
    }
    if (WAVES_M > 1) {
      __syncthreads();
      if (tid < BN) {
        const int j = j0 + tid;
        if (j < p.N) {
          if (f.epi & 2) go0[(int64_t)tm * p.N + j] = red[(0 * 2 + 0) * BN + tid] + red[(0 * 2 + 1) * BN + tid];
          if (f.epi & 4) go1[(int64_t)tm * p.N + j] = red[(1 * 2 + 0) * BN + tid] + red[(1 * 2 + 1) * BN + tid];
        }
      }
    }
  }
}

// Sum split-K slabs in a fixed order into C.  With sym != 0 only elements j <= i are summed (coalesced
// reads of the lower triangle) and each result is also written to its mirror position.
__global__ void __launch_bounds__(256) slab_reduce_kernel(const GemmProblem* __restrict__ probs, int nsplit, int sym,
                                                          double alpha) {
  const GemmProblem p = probs[blockIdx.z];
  const int64_t total = (int64_t)p.M * p.N;
  // two adjacent elements per thread (16-byte loads) when the rows allow it, four slabs in flight per accumulator set: the
  // slabs are read once, in a fixed order per element (the sum's association is k = 0, 4, 8, .. | 1, 5, .. | .. then the four)
  const bool pair = ((p.N & 1) == 0) && ((((uintptr_t)p.o2) & 15) == 0) && ((total & 1) == 0);
  const int64_t step = pair ? 2 : 1;
  for (int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * step; idx < total; idx += (int64_t)gridDim.x * blockDim.x * step) {
    const int i = (int)(idx / p.N), j = (int)(idx % p.N);
    if (sym && j > i) continue;
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    const double* src = p.o2 + idx;
    int k = 0;
    if (pair) {
      for (; k + 4 <= nsplit; k += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) { const dbl2 v = *(const dbl2*)(src + (int64_t)(k + q) * total); s0[q] += v.x; s1[q] += v.y; }
      }
      for (; k < nsplit; k++) { const dbl2 v = *(const dbl2*)(src + (int64_t)k * total); s0[0] += v.x; s1[0] += v.y; }
    } else {
      for (; k + 4 <= nsplit; k += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) s0[q] += src[(int64_t)(k + q) * total];
      }
      for (; k < nsplit; k++) s0[0] += src[(int64_t)k * total];
    }
    const double a = ((s0[0] + s0[1]) + (s0[2] + s0[3])) * alpha;
    p.C[(int64_t)i * p.ldc + j] = a;
    if (sym && j < i) p.C[(int64_t)j * p.ldc + i] = a;
    if (pair) {
      const double b = ((s1[0] + s1[1]) + (s1[2] + s1[3])) * alpha;
      if (!(sym && j + 1 > i)) {
        p.C[(int64_t)i * p.ldc + j + 1] = b;
        if (sym && j + 1 < i) p.C[(int64_t)(j + 1) * p.ldc + i] = b;
      }
    }
  }
  if (p.v2 && p.o1 && blockIdx.x == 0) {   // u = sum over K-slices of the fused row-dot partials; o0 = u, v0 += u
    for (int i = threadIdx.x; i < p.M; i += blockDim.x) {
      double s = 0.0;
      for (int k = 0; k < nsplit; k++) s += p.o1[(int64_t)k * p.M + i];
      p.o0[i] = s;
      if (p.xa) const_cast<double*>(p.xa)[i] += s;
    }
  }
}

int gemm_rowblocks(int M, int big_tiles) { int bm = big_tiles ? 128 : 64; return (M + bm - 1) / bm; }

template <int BM, int BN, bool TA, bool TB, int TAG>
static gp_status launch_one(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, GemmDevFlags df,
                            int ksplit) {
  using S = GemmSmem<BM, BN, TA, TB>;
  df.tilesM = (maxM + BM - 1) / BM;
  if (df.tm0 > 0 || df.tilesM_req > 0) {       // a row-block range (never with split-K)
    const int all = df.tilesM;
    if (ksplit > 1 || df.tm0 >= all) return GP_OK;
    df.tilesM = (df.tilesM_req > 0 && df.tm0 + df.tilesM_req < all) ? df.tilesM_req : all - df.tm0;
  }
  df.tilesN = (maxN + BN - 1) / BN;
  df.ksplit = ksplit;
  int ntiles = df.tilesM * df.tilesN;
  if (ksplit > 1 && df.triC == TRI_LOWER) ntiles = df.tilesM * (df.tilesM + 1) / 2;
  dim3 grid(ntiles * (ksplit > 1 ? ksplit : 1), 1, batch);
  // the dynamic-LDS limit is a per-DEVICE attribute of the function: one bit per device, per instantiation (a
  // repeated call from another thread is harmless)
  static std::atomic<uint32_t> attr_devs{0};
  const uint32_t bit = 1u << (h->device & 31);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)gemm_f64_kernel<BM, BN, TA, TB, TAG>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES));
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, TA, TB, TAG>), grid, dim3(gemm_threads(BM, TAG)), S::BYTES, h->stream, d_probs, df);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

template <int B>
static gp_status dispatch_trans(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN,
                                const GemmFlags& f, GemmDevFlags df, int ksplit) {
  if (!f.transA && !f.transB) return launch_one<B, B, false, false, 0>(h, d_probs, batch, maxM, maxN, df, ksplit);
  if (f.transA && !f.transB) return launch_one<B, B, true, false, 0>(h, d_probs, batch, maxM, maxN, df, ksplit);
  if (!f.transA && f.transB) return launch_one<B, B, false, true, 0>(h, d_probs, batch, maxM, maxN, df, ksplit);
  return launch_one<B, B, true, true, 0>(h, d_probs, batch, maxM, maxN, df, ksplit);
}

static GemmDevFlags to_dev(const GemmFlags& f) {
  GemmDevFlags df;
  df.triA = f.triA; df.triB = f.triB; df.triC = f.triC;
  df.alpha = f.alpha; df.beta = f.beta;
  df.epi = f.epilogue;
  df.scale_mode = f.scale_mode;
  df.ksplit = 1; df.tilesM = df.tilesN = 1;
  df.tm0 = f.tile_m0; df.tilesM_req = f.tile_mcount;
  return df;
}

gp_status launch_gemm_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN,
                              const GemmFlags& f) {
  if (batch <= 0 || maxM <= 0 || maxN <= 0) return GP_OK;
  GpTimerScope ts(h, f.timer);
  {   // whole aligned strips: a 64 x 64 tile per wavefront, no LDS, no barrier (gemm_wave.hip) ...
    gp_status st = GP_OK;
    if ((f.role == 3 || f.role == 5 || f.rows64_ok) && launch_gemm_wave(h, d_probs, batch, maxM, maxN, f, &st)) return st;
  }
  {   // ... or the form whose K loop issues no vector-ALU instructions (gemm_strip.hip)
    gp_status st = GP_OK;
    if (launch_gemm_strip_lean(h, d_probs, batch, maxM, maxN, f, &st)) return st;
  }
  GemmDevFlags df = to_dev(f);
  // the three frame-strip products of the hot path get their own symbols
  if (f.role == 1) return launch_one<128, 128, false, false, 1>(h, d_probs, batch, maxM, maxN, df, 1);
  if (f.role == 2) return launch_one<128, 128, true, false, 2>(h, d_probs, batch, maxM, maxN, df, 1);
  if (f.role == 3) return launch_one<128, 128, false, false, 3>(h, d_probs, batch, maxM, maxN, df, 1);
  if (f.big_tiles) return dispatch_trans<128>(h, d_probs, batch, maxM, maxN, f, df, 1);
  // a launch that would not put a 64 x 64 workgroup on every CU (one or two Kuu-sized problems: the dependent M x M chains of an
  // SGPRSS evaluation or of a one-pitch model) takes 32 x 32 tiles: four times the workgroups, a quarter of the K loop's work each
  // (plain products only: the column-sum epilogues' partial rows are laid out per 64-row block)
  if (gp_switches().gemm_tile32 && f.role == 0 && f.epilogue == EPI_STORE && f.tile_m0 == 0 && f.tile_mcount == 0 && maxM >= 64 && maxN >= 64 &&
      (int64_t)batch * ((maxM + 63) / 64) * ((maxN + 63) / 64) < (int64_t)gp_switches().gemm_tile32)
    return dispatch_trans<32>(h, d_probs, batch, maxM, maxN, f, df, 1);
  return dispatch_trans<64>(h, d_probs, batch, maxM, maxN, f, df, 1);
}

int gemm_nt_nsplit(int M, int Nlong, int batch) {
  // enough K-slices that (tiles x slices x batch) covers the chip ~4 times over (2 workgroups per CU), each
  // slice >= 512 deep; fewer slices = less slab traffic
  int tiles = ((M + 127) / 128);
  tiles = tiles * (tiles + 1) / 2 * (batch > 0 ? batch : 1);
  // (coverage target: 2048 workgroups for the long batches; short ones — a few latent GPs, cfg3's 72 tiles — sit on the step's
  // critical chain with their slab reduction behind them and come out 0.5 - 1.5 % ahead with half the slices: nt_cover = 0 picks)
  const int cover = gp_switches().nt_cover > 0 ? gp_switches().nt_cover : (tiles < 128 ? 1024 : 2048);
  int want = (cover + tiles - 1) / tiles;
  int maxs = (Nlong + 511) / 512;
  // a window-sized problem (one tile, a few thousand frames) would be 4 workgroups of 32 K-tiles each: latency, not
  // throughput — slices down to 128 deep then
  if ((int64_t)tiles * maxs < 256) maxs = (Nlong + 127) / 128;
  int s = want < maxs ? want : maxs;
  if (s < 2) s = 2;
  if (s > 64) s = 64;
  return s;
}

gp_status launch_gemm_nt_reduce_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxNlong,
                                        int nsplit, int sym, int scale_by_k, double alpha, int uniform_aligned) {
  if (batch <= 0 || maxM <= 0) return GP_OK;
  bool lean = false;
  if (uniform_aligned) {
    GpTimerScope ts(h, GP_TIMER_NT_GEMM);
    gp_status st = GP_OK;
    lean = launch_gemm_strip_nt_lean(h, d_probs, batch, maxM, maxNlong, nsplit > 1 ? nsplit : 2, sym, scale_by_k, &st);
    if (lean) GP_CHECK(st);
  }
  if (!lean) {
    GpTimerScope ts(h, GP_TIMER_NT_GEMM);
    GemmFlags f;
    f.transA = 0; f.transB = 1;
    f.triC = sym ? TRI_LOWER : TRI_NONE;
    f.scale_mode = scale_by_k ? 2 : 0;
    GemmDevFlags df = to_dev(f);
    df.epi = 1;
    GP_CHECK((launch_one<128, 128, false, true, 4>(h, d_probs, batch, maxM, maxM, df, nsplit > 1 ? nsplit : 2)));
  }
  return launch_slab_reduce(h, d_probs, batch, maxM, nsplit > 1 ? nsplit : 2, sym, alpha);
}

gp_status launch_slab_reduce(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int nsplit, int sym, double alpha) {
  GpTimerScope ts(h, GP_TIMER_SMALL_GEMM);
  int blocks = (int)(((int64_t)maxM * maxM / 2 + 255) / 256);      // (two elements per thread where rows are even)
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks, 1, batch), dim3(256), 0, h->stream, d_probs, nsplit, sym, alpha);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
