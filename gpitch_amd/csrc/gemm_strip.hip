// gemm_strip.hip — the three M x N frame-strip products of the conditional / its backward pass with a K loop that
// issues (almost) no vector-ALU instructions (gfx950, round 3).
//
//   role 1  A   = W Kuf      tf.matrix_triangular_solve(Lm, Kmn)  + reduce_sum(A^2), A^T q_mu   (GPflow conditional,
//   role 2  LTA = Lq^T A     tf.matmul(Lq^T, A) -> reduce_sum(LTA^2) only                         gpitch/pdgp.py:147-155)
//   role 3  G   = R (A D)    backward: Kuf_bar (columns of A scaled by D = 2 gv)
//
// Why a second form of gemm.hip's kernel.  Measured on MI355X (profiles/r03): a v_mfma_f64_16x16x4_f64 holds the SIMD's
// vector ALU for its 64 cycles — the float64 "matrix core" is the float64 vector lanes — so every vector instruction of
// either wavefront on that SIMD, integer address arithmetic included, ADDS ~6 cycles to the matrix time instead of hiding
// under it: 48 dummy v_add_u32 per K-tile in gemm.hip's loop cost the dense product +5.7 %, 96 cost +10.7 %.  gemm.hip's
// loop has ~45 per K-tile (LDS fragment addresses recomputed because the stage index is a run-time value, 64-bit global
// addresses, structural masks on every tile) and ~830 per output tile in the epilogue (a 64-bit address, bounds and
// alpha/beta handling per 8-byte store).  Here:
//  - the K loop is unrolled over the two LDS stages, so every LDS address is one per-lane base register + an immediate;
//  - global operand tiles are fetched as  scalar base (advanced by the scalar ALU) + one constant 32-bit lane offset;
//  - structural masks run only on the K-tiles that cross the diagonal block;
//  - the epilogue transposes the accumulators through the (then idle) staging LDS and stores 16 bytes per lane, four rows
//    x 256 bytes per instruction, again scalar base + constant lane offset: no vector arithmetic per store.
// Tiling, wave layout (1 x 4 wavefronts of 128 x 32), MFMA order and the order of every reduction are gemm.hip's: results
// are bit-identical.  Whole, aligned tiles only (every problem of the batch M = K-compatible multiples of 128, N a
// multiple of 128, even leading dimensions, alpha = 1, beta = 0): the launcher returns false otherwise and gemm.hip runs.
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

#define GS_BK 16
#define GS_BM 128
#define GS_BN 128
// LDS tiles are FRAGMENT-MAJOR: the 64 doubles one MFMA operand fragment consists of (lane = 16 (k & 3) + (row or column & 15))
// are contiguous, fragment (ks = k / 4, t = row or column tile) at double offset (8 ks + t) * 64.  Every fragment read of
// the K loop is then  ONE per-lane base register (8 * lane) + a multiple of 512 bytes  — ds_read2st64_b64 reaches all of a
// stage's 32 + 8 fragments from it, no address arithmetic — and conflict-free by construction (a wavefront reads 512
// contiguous bytes), with no padding: 2 x 16 KiB per stage.
#define GS_FRAG 64               // doubles per fragment
#ifndef GS_MFMA_PRIO
#define GS_MFMA_PRIO 2
#endif

#ifdef GS_STAMPS
// diagnostic build (never shipped): per-workgroup s_memtime stamps [entry, prologue done, K loop done, epilogue done] + (tm, nkt)
__device__ unsigned long long gs_stamps[6 * 65536];
__device__ unsigned int gs_stamp_count;
extern "C" int gp_debug_strip_stamps(unsigned long long* host, int max_records) {
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(gs_stamp_count), sizeof(n)) != hipSuccess) return -1;
  if ((int)n > max_records) n = max_records;
  if (n > 65536) n = 65536;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(gs_stamps), (size_t)n * 6 * sizeof(unsigned long long)) != hipSuccess) return -1;
  unsigned int z = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(gs_stamp_count), &z, sizeof(z));
  return (int)n;
}
#define GS_STAMP(i) do { if (threadIdx.x == 0) st_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GS_STAMP(i) do { } while (0)
#endif

#ifndef GS_PAD_BYTES
#define GS_PAD_BYTES 0      // occupancy probe: extra dynamic LDS per workgroup (one workgroup per CU from ~16 KiB on)
#endif
struct StripFlags {
  int tilesM, tilesN, tm0;
  int epi;            // EPI_* bitmask
  double alpha;       // role 3 only, a power of two: folded into the column scales (exact)
  const double* xcols; // role 5 only: the frames x (the B strip's columns; not part of any descriptor: pdgp.hip pdgp_bind)
};

template <bool TA> struct StripSmem {
  static constexpr int A_ELEMS = GS_BM * GS_BK;
  static constexpr int B_ELEMS = GS_BK * GS_BN;
  static constexpr int STAGE = A_ELEMS + B_ELEMS;
  static constexpr int V0_OFF = 2 * STAGE;               // 128 doubles behind the stages: v0[i0 .. i0 + 127] (EPI_COLDOT)
  static constexpr int SINK_OFF = V0_OFF + GS_BM;        // 4 x 32 doubles: the L2-warming loads' sink, one per wavefront
  static constexpr size_t BYTES = (size_t)(2 * STAGE + GS_BM + 4 * 32) * sizeof(double) + GS_PAD_BYTES;
};

__device__ __forceinline__ gcbytes gs_uniform(gcbytes p) {
  const uint64_t b = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (gcbytes)(((uint64_t)hi << 32) | lo);
}

// L2 warming: the strips are streamed from HBM once per product and the K loop fetches only one K-tile ahead — with the
// HBM round trip (5-8 k cycles under load) longer than a K-tile whose MFMA count is small (the diagonal tiles of the
// symmetric product: 8.0 k cycles per K-tile measured for 4.6 k of MFMA work) the loop runs at memory latency.  One
// dword per 128-byte line, GS_PF K-tiles ahead, as an LDS-DMA load (global_load_lds_dword: no destination register — an
// inline-asm load into a scratch VGPR would land long after the compiler has reused that register) into a 256-byte sink
// per wavefront that nobody reads: the real loads then hit the XCD's L2.
// MEASURED (same box, GS_PF = 3 and 6 against 0): the dense product 6.34 -> 6.24 ms, but the symmetric split-K product
// 4.18 -> 4.49 ms, A = W Kuf 3.97 -> 4.16, Lq^T A +2.5 %: the extra requests cost more than the latency they hide.  Off.
#ifndef GS_PF
#define GS_PF 0
#endif
#ifndef GS_EXTRA_VALU
#define GS_EXTRA_VALU 0
#endif
typedef __attribute__((address_space(3))) void* gs_ldsptr;
__device__ __forceinline__ void gs_touch(gcbytes base, uint32_t voff, double* sink) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff), (gs_ldsptr)sink, 4, 0, 0);
}

// TAG 1: op(A) = W (lower), NN.  TAG 2: op(A) = Lq^T (upper; A read transposed), K walked downwards.  TAG 3: dense, NN,
// B(k, n) *= v1[n].  TAG 5 (KT = a stationary kernel type): TAG 3's product with the Kuf-side hyper-gradient contraction of
// a stationary kernel as its epilogue — Kuf_bar = R (A D) of a latent GP whose inducing inputs are fixed is wanted for
// nothing but  sum_ij (Kuf_bar_ij + alpha_i gm_j) dK_ij/d(variance, lengthscale)  (bwd.hip: hyper_contract_kernel), so the
// tile is contracted straight out of the accumulators and never stored: one partial record (2 sums) per tile instead of
// a 134-MB strip written and read back per GP.  Same per-entry arithmetic as hyper_contract_kernel<1, false, false, KT>.
template <int TAG, int KT = -1>
__global__ void __launch_bounds__(256, 2) gemm_strip_kernel(const GemmProblem* __restrict__ probs, StripFlags f) {
  constexpr bool SCALE = (TAG == 3 || TAG == 5);
  constexpr bool TA = (TAG == 2);
  constexpr bool KDOWN = (TAG == 2);
  constexpr int TRI = (TAG == 1) ? TRI_LOWER : (TAG == 2) ? TRI_UPPER : TRI_NONE;
  using S = StripSmem<TA>;
  constexpr int TM = 8, TN = 2;           // 16 x 16 MFMA tiles per wavefront: 128 rows x 32 columns
  constexpr int EA = 8, EB = 8;           // staged elements per thread and K-tile
  extern __shared__ __attribute__((aligned(16))) double smem[];

  // XCD-aware renumbering of the flattened (tile, batch) grid (gemm.hip): blocks b and b + 8 share an XCD and its L2
  int bid = blockIdx.x, bz = blockIdx.z;
  {
    const int nx = gridDim.x, total = nx * (int)gridDim.z;
    if ((total & 7) == 0) {
      const int lin = bz * nx + bid;
      const int log = (lin & 7) * (total >> 3) + (lin >> 3);
      bz = log / nx; bid = log - bz * nx;
    }
  }
#ifdef GS_STAMPS
  unsigned long long st_[4] = {0, 0, 0, 0};
#endif
  GS_STAMP(0);
  const GemmProblem p = probs[bz];
  const int tn = bid / f.tilesM;
  const int tm = f.tm0 + (bid % f.tilesM + tn) % f.tilesM;     // row-blocks of a strip adjacent, rotated over the shader engines
  const int i0 = tm * GS_BM, j0 = tn * GS_BN;
  if (i0 >= p.M || j0 >= p.N) return;
  int kbeg = 0, kend = p.K;
  if (TRI == TRI_LOWER) kend = min(kend, i0 + GS_BM);
  if (TRI == TRI_UPPER) kbeg = max(kbeg, i0);
  const int nkt = (kend - kbeg) / GS_BK;

  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;

  // the tile's 128 entries of v0 (q_mu for A^T q_mu) go to LDS now: the epilogue's 32 reads per lane are then immediates off
  // one base instead of 64 global loads with 64-bit addresses (measured: 18 k of the 27 k cycles of role 1's epilogue)
  if ((f.epi & EPI_COLDOT) && tid < GS_BM) smem[S::V0_OFF + tid] = ((gcptr)p.v0)[i0 + tid];
  // ---- operand staging: constant per-lane byte offsets, scalar bases that walk along K --------------------------
  // A, k-contiguous (TAG 1, 3): thread -> row a_i = tid / 2, eight consecutive k from a_k = 8 (tid & 1)
  // A, transposed   (TAG 2)   : thread -> k row a_k = tid / 16, row pairs a_i + 32 q, a_i = 2 (tid & 15)
  // B (row-contiguous)        : thread -> k row b_k = tid / 16, column pairs b_n + 32 q, b_n = 2 (tid & 15)
  const int a_i = TA ? (tid & 15) * 2 : (tid >> 1);
  const int a_k = TA ? (tid >> 4) : (tid & 1) * 8;
  const int b_k = tid >> 4, b_n = (tid & 15) * 2;
  const uint32_t voffA = TA ? (uint32_t)(((int64_t)a_k * p.lda + i0 + a_i) * 8) : (uint32_t)(((int64_t)(i0 + a_i) * p.lda + a_k) * 8);
  const uint32_t voffB = (uint32_t)(((int64_t)b_k * p.ldb + j0 + b_n) * 8);
  const int kfirst = KDOWN ? kend - GS_BK : kbeg;
  const int64_t stepA = (KDOWN ? -1 : 1) * (TA ? (int64_t)GS_BK * p.lda * 8 : (int64_t)GS_BK * 8);
  const int64_t stepB = (KDOWN ? -1 : 1) * (int64_t)GS_BK * p.ldb * 8;
  gcbytes sA = gs_uniform((gcbytes)p.A + (TA ? (int64_t)kfirst * p.lda * 8 : (int64_t)kfirst * 8));
  gcbytes sB = gs_uniform((gcbytes)p.B + (int64_t)kfirst * p.ldb * 8);
  double ra[EA], rbX[EB], rbY[EB], rs[EB];
  if (SCALE) {
    const gcptr gv1 = (gcptr)p.v1;
#pragma unroll
    for (int e = 0; e < EB; e++) rs[e] = f.alpha * gv1[j0 + b_n + (e >> 1) * 32 + (e & 1)];   // (alpha = 2^k: exact)
  }
  // B strip rows of a K-tile: 16 rows x 8 lines of 128 bytes; wavefront 0 touches rows 0-7, wavefront 1 rows 8-15
  const uint32_t voffP = (uint32_t)(((int64_t)((lane >> 3) + 8 * (wc & 1)) * p.ldb + j0 + (lane & 7) * 16) * 8);
  int nld = 0;                              // B K-tiles requested so far (scalar)
  gcbytes sP = gs_uniform((gcbytes)((int64_t)sB + GS_PF * stepB));
  const bool toucher = __builtin_amdgcn_readfirstlane(wc) < 2;
  // raw loads only: nothing here consumes a loaded value.  op(A) (M x M, L2-resident) is requested one K-tile ahead; the
  // B strip — streamed from HBM once per product, every row-block of a strip waiting on the same fetch — TWO ahead,
  // into two register sets that alternate with the (unrolled) LDS stages: with one K-tile of lookahead the K-tiles whose
  // MFMA count is small (the diagonal block: 7.1 k cycles per K-tile measured for 2.3 k of MFMA work) ran at HBM latency.
  auto load_A = [&]() {
#pragma unroll
    for (int e = 0; e < EA; e += 2) {
      const dbl2 v = *(gcptr2)(sA + voffA + (TA ? (e >> 1) * 256 : e * 8));
      ra[e] = v.x; ra[e + 1] = v.y;
    }
    sA = (gcbytes)((int64_t)sA + stepA);    // (scalar ALU)
  };
  auto load_B = [&](double (&rb)[EB]) {
    if (GS_PF > 0 && toucher && nld + GS_PF < nkt) gs_touch(sP, voffP, smem + S::SINK_OFF + 32 * wc);
    sP = (gcbytes)((int64_t)sP + stepB);
    nld++;
#pragma unroll
    for (int e = 0; e < EB; e += 2) {
      const dbl2 v = *(gcptr2)(sB + voffB + (e >> 1) * 256);
      rb[e] = v.x; rb[e + 1] = v.y;
    }
    sB = (gcbytes)((int64_t)sB + stepB);
  };
  // LDS: per-lane bases of stage 0; stage 1 = + STAGE * 8 bytes (an immediate).  Element (row i, k) of op(A) sits at
  // (8 (k / 4) + i / 16) * 64 + 16 (k & 3) + (i & 15); element (k, column n) of B at A_ELEMS + (8 (k / 4) + n / 16) * 64 + ...
  const int wA = TA ? ((8 * (a_k >> 2) + (a_i >> 4)) * GS_FRAG + 16 * (a_k & 3) + (a_i & 15))
                    : ((8 * (a_k >> 2) + (a_i >> 4)) * GS_FRAG + (a_i & 15));          // (a_k = 0 or 8: k & 3 = 0)
  const int wB = S::A_ELEMS + (8 * (b_k >> 2) + (b_n >> 4)) * GS_FRAG + 16 * (b_k & 3) + (b_n & 15);
  // mask_tag = false_type: the K-tile lies wholly outside the diagonal block (no structural zeros to write)
  auto store_tiles = [&](const int stage_off, int kt, auto mask_tag, double (&rb)[EB]) {       // stage_off: 0 or STAGE (literals)
    constexpr bool MASK = decltype(mask_tag)::value;
    double* As = smem + stage_off + wA;
    double* Bs = smem + stage_off + wB;
    // structural zeros of op(A): only K-tiles inside the diagonal block have any
    if (MASK && TRI == TRI_LOWER && kt + GS_BK > i0) {
#pragma unroll
      for (int e = 0; e < EA; e++) if (kt + a_k + e > i0 + a_i) ra[e] = 0.0;
    }
    if (MASK && TRI == TRI_UPPER && kt < i0 + GS_BM) {
#pragma unroll
      for (int e = 0; e < EA; e++) if (kt + a_k < i0 + a_i + (e >> 1) * 32 + (e & 1)) ra[e] = 0.0;
    }
    if (SCALE) {
#pragma unroll
      for (int e = 0; e < EB; e++) rb[e] *= rs[e];
    }
    if (TA) {        // row pairs i, i + 1 (same fragment); pair q is 32 rows = two row tiles further on
#pragma unroll
      for (int e = 0; e < EA; e += 2) *reinterpret_cast<double2*>(As + (e >> 1) * 2 * GS_FRAG) = make_double2(ra[e], ra[e + 1]);
    } else {         // eight consecutive k of one row: k-step e / 4 (8 fragments further on), k-slot e & 3 (16 doubles)
#pragma unroll
      for (int e = 0; e < EA; e++) As[(e >> 2) * 8 * GS_FRAG + (e & 3) * 16] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < EB; e += 2) *reinterpret_cast<double2*>(Bs + (e >> 1) * 2 * GS_FRAG) = make_double2(rb[e], rb[e + 1]);
  };

  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

  // fragment reads: per-lane base + immediates (a * 16 rows, ks * 4 k, b * 16 columns, stage)
  const int rA = lane;                                        // fragment (ks, a): + (8 ks + a) * 64
  const int rB = S::A_ELEMS + 2 * wc * GS_FRAG + lane;        // fragment (ks, 2 wc + b): + (8 ks + b) * 64
  // straight-line K-tile: every MFMA tile of op(A) is (treated as) dense; stage offset a literal
  auto mfma_full = [&](const int stage_off) {
    const double* As = smem + stage_off + rA;
    const double* Bs = smem + stage_off + rB;
#if GS_EXTRA_VALU > 0
    // measurement probe (tools/valu_probe.sh): N dependent-free integer vector adds per K-tile.  They cost matrix time
    // one for one — v_mfma_f64_16x16x4_f64 holds the SIMD's vector ALU for its 64 cycles
    {
      int dummy = lane;
#pragma unroll
      for (int e = 0; e < GS_EXTRA_VALU; e++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
      asm volatile("" :: "v"(dummy));
    }
#endif
    __builtin_amdgcn_s_setprio(GS_MFMA_PRIO);
#pragma unroll
    for (int ks = 0; ks < GS_BK / 4; ks++) {
      double af[TM], bf[TN];
#pragma unroll
      for (int b = 0; b < TN; b++) bf[b] = Bs[(8 * ks + b) * GS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) af[a] = As[(8 * ks + a) * GS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // K-tile inside the diagonal block of a triangular op(A): 16-row MFMA tiles that are structurally zero for a k-step
  // are skipped (wave-uniform); stage offset a run-time value (only the eight K-tiles of the diagonal block come here)
  auto mfma_diag = [&](const int stage_off, int kt) {
    const double* As = smem + stage_off + rA;
    const double* Bs = smem + stage_off + rB;
    __builtin_amdgcn_s_setprio(GS_MFMA_PRIO);
#pragma unroll
    for (int ks = 0; ks < GS_BK / 4; ks++) {
      const int kg = kt + ks * 4;      // first global k of this MFMA step
      int a_lo = 0, a_hi = TM;
      if (TRI == TRI_LOWER) a_lo = max(0, (kg - i0) >> 4);             // need i0 + 16 a + 15 >= kg
      if (TRI == TRI_UPPER) a_hi = min(TM, ((kg + 3 - i0) >> 4) + 1);  // need i0 + 16 a <= kg + 3
      double af[TM], bf[TN];
#pragma unroll
      for (int b = 0; b < TN; b++) bf[b] = Bs[(8 * ks + b) * GS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) af[a] = As[(8 * ks + a) * GS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) {
        if (a >= a_lo && a < a_hi) {
#pragma unroll
          for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- K loop ----------------------------------------------------------------------------------------------------------
  // The K-tiles that lie wholly outside the diagonal block of a triangular op(A) (all of them for the dense product) run
  // in pairs with the two LDS stages as literals and no structural masks; the remaining ones (eight, plus an odd one)
  // follow with a run-time stage, masks and MFMA-tile skipping.
  {
    const int kstep = KDOWN ? -GS_BK : GS_BK;
    int nplain = nkt;                                                   // leading K-tiles with no structural zero at all
    if (TRI == TRI_LOWER) nplain = i0 / GS_BK;                          // kt + 16 <= i0
    if (TRI == TRI_UPPER) nplain = (kend - (i0 + GS_BM)) / GS_BK;       // kt >= i0 + 128 (walking down from kend - 16)
    const int npair = (max(0, min(nplain, nkt)) / 2) * 2;
    const std::true_type masked{};
    const std::false_type plain{};
    int kt = kfirst;
    load_A();
    load_B(rbX);
    if (npair > 0) store_tiles(0, kt, plain, rbX); else store_tiles(0, kt, masked, rbX);
    __syncthreads();
    if (nkt > 1) load_B(rbY);                 // B of K-tile 1: in flight over the whole first K-tile
    GS_STAMP(1);
    // (nkt is a multiple of 8: both loops walk pairs of K-tiles, stages and B register sets alternating as literals.  The
    //  last plain pair goes to the second loop: its second store is the first masked tile, and a masked store inside the
    //  first loop gets if-converted into per-element selects on EVERY iteration)
    int it = 0;
    for (; it + 2 < npair; it += 2) {
      load_A();                               // A(it + 1)
      load_B(rbX);                            // B(it + 2)
      mfma_full(0);
      store_tiles(S::STAGE, kt + kstep, plain, rbY);
      __syncthreads();
      kt += kstep;
      load_A();                               // A(it + 2)
      if (it + 3 < nkt) load_B(rbY);          // B(it + 3)
      mfma_full(S::STAGE);
      store_tiles(0, kt + kstep, plain, rbX);
      __syncthreads();
      kt += kstep;
    }
    for (; it < nkt; it += 2) {
      load_A();
      if (it + 2 < nkt) load_B(rbX);
      mfma_diag(0, kt);
      store_tiles(S::STAGE, kt + kstep, masked, rbY);
      __syncthreads();
      kt += kstep;
      const bool more = (it + 2 < nkt);
      if (more) load_A();
      if (it + 3 < nkt) load_B(rbY);
      mfma_diag(S::STAGE, kt);
      if (more) store_tiles(0, kt + kstep, masked, rbX);
      __syncthreads();
      kt += kstep;
    }
  }

  GS_STAMP(2);
  // ---- epilogue ---------------------------------------------------------------------------------------------------
  // (every wavefront is past its last fragment read: the final barrier of the K loop)
  if (f.epi & EPI_STORE) {
    // through LDS: the accumulator layout puts 16 columns of FOUR rows on a wavefront's lanes (4 x 128-byte pieces per
    // 8-byte store); 32 rows at a time go into a per-wavefront [32][34] tile and leave as 16 bytes per lane, four rows
    // x 256 bytes per instruction.
    constexpr int TS = 34;
    double* tw = smem + wc * (32 * TS);                       // 4 x 8704 bytes <= 2 stages
    const int srow = lane >> 4, scol = (lane & 15) * 2;
    const uint32_t voffC = (uint32_t)(((int64_t)(i0 + srow) * p.ldc + j0 + wc * 32 + scol) * 8);
    const int64_t rowstride = p.ldc * 8;
#pragma unroll
    for (int half = 0; half < 4; half++) {
#pragma unroll
      for (int a4 = 0; a4 < 2; a4++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) tw[(a4 * 16 + kq + 4 * r) * TS + b * 16 + lc] = acc[half * 2 + a4][b][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();          // LDS operations of one wavefront complete in order
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const dbl2 v = *reinterpret_cast<const dbl2*>(tw + (4 * q + srow) * TS + scol);
        const gbytes cb = (gbytes)gs_uniform((gcbytes)p.C + (int64_t)(half * 32 + 4 * q) * rowstride);
        *(gptr2)(cb + voffC) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();          // the tile is rewritten by the next 32 rows
    }
  }
  if (f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) {
    // per-column reductions over this tile's rows: sum acc^2 and sum acc * v0[row]  (gemm.hip's order)
    const double* v0s = smem + S::V0_OFF + kq;          // (written in the prologue; every barrier since orders it)
    const gptr go0 = (gptr)p.o0, go1 = (gptr)p.o1;
#pragma unroll
    for (int b = 0; b < TN; b++) {
      double s2 = 0.0, sd = 0.0;
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double v = acc[a][b][r];
          s2 = fma(v, v, s2);
          if (f.epi & EPI_COLDOT) sd = fma(v, v0s[a * 16 + 4 * r], sd);
        }
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
      const int j = j0 + wc * 32 + b * 16 + lc;
      if (kq == 0) {
        if (f.epi & EPI_COLSUMSQ) go0[(int64_t)tm * p.N + j] = s2;
        if (f.epi & EPI_COLDOT) go1[(int64_t)tm * p.N + j] = sd;
      }
    }
  }
  if (TAG == 5) {
    // acc[a][b][r] = Kuf_bar(i0 + 16 a + kq + 4 r, j0 + 32 wc + 16 b + lc).  Row tables (z_i / l, alpha_i) and the exp table go
    // to the (now free) LDS stages; sums over the tile in a fixed order: lanes by shuffles, wavefronts through LDS.
    double* etab = smem;                 // GP_EXP_TAB = 64
    double* row_a = smem + 64;           // 128
    double* row_al = smem + 192;         // 128
    double* red = smem + 320;            // 4 x 2
    const gcptr th = (gcptr)p.kern.theta;
    const double var = th[0], ls = th[1];
    gp_exp_tab_init(etab);
    if (tid < GS_BM) { row_a[tid] = ((gcptr)p.xa)[i0 + tid] / ls; row_al[tid] = ((gcptr)p.v0)[i0 + tid]; }
    __syncthreads();
    const double inv_ls = 1.0 / ls;
    double acc_v = 0.0, acc_l = 0.0;
    const gcptr gx = (gcptr)f.xcols, ggm = (gcptr)p.v2;
#pragma unroll
    for (int b = 0; b < TN; b++) {
      const int j = j0 + wc * 32 + b * 16 + lc;
      const double bcol = gx[j] / ls, bb = __dmul_rn(bcol, bcol), gmj = ggm[j];
#pragma unroll
      for (int a = 0; a < TM; a++) {
        __builtin_amdgcn_sched_barrier(0);       // one 16-row tile at a time: hoisting all 64 table reads spilled 36 VGPRs
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int ii = a * 16 + kq + 4 * r;
          const double av = row_a[ii], aa = __dmul_rn(av, av);
          const double w = fma(row_al[ii], gmj, acc[a][b][r]);
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
    if (lane == 0) { red[wc * 2 + 0] = acc_v; red[wc * 2 + 1] = acc_l; }
    __syncthreads();
    if (tid < 2) {
      const double sacc = (red[0 * 2 + tid] + red[1 * 2 + tid]) + (red[2 * 2 + tid] + red[3 * 2 + tid]);
      ((gptr)p.o0)[((int64_t)tm * f.tilesN + tn) * 2 + tid] = sacc;
    }
  }
#ifdef GS_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  GS_STAMP(3);
  if (threadIdx.x == 0) {
    const unsigned int slot = atomicAdd(&gs_stamp_count, 1u);
    if (slot < 65536) {
      unsigned long long* o = gs_stamps + 6 * (size_t)slot;
      o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3]; o[4] = (unsigned long long)(TAG * 100 + tm); o[5] = (unsigned long long)nkt;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// role 4: the symmetric split-K product over the frames, H = X diag(d) X^T (backward: X = A, d = 2 gv; sgpr_ss.py:49
// tf.matmul(A, A, transpose_b=True): d absent), with u = X v2 fused into the tiles of the first tile column.  Both
// operand tiles are k-contiguous rows of X; same fragment-major LDS stages, scalar-base staging and unrolled K loop as
// above (gemm.hip's TAG 4 is the reference for tile enumeration, K-slicing, the (wc, 7 - wc) column-tile pairing that
// balances the skipped upper MFMA tiles of a diagonal output tile, and the slab layout).  Bit-identical slabs.
struct StripNtFlags {
  int tilesM, ksplit, sym;
  int scale;          // B(k, n) *= v1[k]
};

__global__ void __launch_bounds__(256, 2) gemm_strip_nt_kernel(const GemmProblem* __restrict__ probs, StripNtFlags f) {
  using S = StripSmem<false>;
  constexpr int TM = 8, TN = 2, EA = 8, EB = 8;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int bid = blockIdx.x, bz = blockIdx.z;
  {
    const int nx = gridDim.x, total = nx * (int)gridDim.z;
    if ((total & 7) == 0) {
      const int lin = bz * nx + bid;
      const int log = (lin & 7) * (total >> 3) + (lin >> 3);
      bz = log / nx; bid = log - bz * nx;
    }
  }
#ifdef GS_STAMPS
  unsigned long long st_[4] = {0, 0, 0, 0};
#endif
  GS_STAMP(0);
  const GemmProblem p = probs[bz];
  // lower tiles only (sym) — inside an XCD's range the output tile varies fastest: the workgroups resident together work
  // on the same K-slice of different tiles and share its operand strips in that XCD's L2
  const int ntl = f.sym ? f.tilesM * (f.tilesM + 1) / 2 : f.tilesM * f.tilesM;
  const int ksl = bid / ntl, t = bid % ntl;
  int tm, tn;
  if (f.sym) {
    tm = (int)((__dsqrt_rn(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
    while (tm * (tm + 1) / 2 > t) tm--;
    tn = t - tm * (tm + 1) / 2;
  } else { tm = t % f.tilesM; tn = t / f.tilesM; }
  const int i0 = tm * GS_BM, j0 = tn * GS_BN;
  if (i0 >= p.M || j0 >= p.N) return;
  int kbeg, kend;
  {
    const int nk = p.K / GS_BK, per = (nk + f.ksplit - 1) / f.ksplit;
    kbeg = ksl * per * GS_BK;
    kend = min(p.K, kbeg + per * GS_BK);
  }
  const int nkt = (kend > kbeg) ? (kend - kbeg) / GS_BK : 0;
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  const bool diag_sym = f.sym && (tm == tn);
  // column tiles of this wavefront: wc and 7 - wc; on a diagonal output tile row tiles above the column tile are skipped
  const int ct0 = wc, ct1 = 7 - wc;
  const int cmin0 = diag_sym ? ct0 : 0, cmin1 = diag_sym ? ct1 : 0;      // (wave-uniform)
  const bool rowdot = (p.v2 != nullptr) && (tn == 0);

  // staging: thread -> row r = tid / 2 of the A tile (i0 + r) and of the B tile (j0 + r), eight consecutive k from 8 (tid & 1)
  const int s_r = tid >> 1, s_k = (tid & 1) * 8;
  const uint32_t voffA = (uint32_t)(((int64_t)(i0 + s_r) * p.lda + s_k) * 8);
  const uint32_t voffB = (uint32_t)(((int64_t)(j0 + s_r) * p.ldb + s_k) * 8);
  const uint32_t voffK = (uint32_t)(s_k * 8);
  gcbytes sA = gs_uniform((gcbytes)p.A + (int64_t)kbeg * 8);
  gcbytes sB = gs_uniform((gcbytes)p.B + (int64_t)kbeg * 8);
  gcbytes sV1 = gs_uniform((gcbytes)p.v1 + (int64_t)kbeg * 8);
  gcbytes sV2 = gs_uniform((gcbytes)p.v2 + (int64_t)kbeg * 8);
  const bool scale = f.scale && (p.v1 != nullptr);
  double ra[EA], rb[EB], rs[EB], rg[EA], udot = 0.0;
  // a K-tile of either operand: 128 rows x one 128-byte line; wavefronts 0, 1 touch the A rows, 2, 3 the B rows
  const uint32_t voffP = (wc < 2) ? (uint32_t)(((int64_t)(i0 + 64 * (wc & 1) + lane) * p.lda) * 8)
                                  : (uint32_t)(((int64_t)(j0 + 64 * (wc & 1) + lane) * p.ldb) * 8);
  int nld = 0;
  gcbytes sP = gs_uniform(((wc < 2) ? (gcbytes)p.A : (gcbytes)p.B) + (int64_t)(kbeg + GS_PF * GS_BK) * 8);
  auto load_tiles = [&]() {
    if (GS_PF > 0 && nld + GS_PF < nkt) gs_touch(sP, voffP, smem + S::SINK_OFF + 32 * wc);
    sP += GS_BK * 8;
    nld++;
#pragma unroll
    for (int e = 0; e < EA; e += 2) { const dbl2 v = *(gcptr2)(sA + voffA + e * 8); ra[e] = v.x; ra[e + 1] = v.y; }
#pragma unroll
    for (int e = 0; e < EB; e += 2) { const dbl2 v = *(gcptr2)(sB + voffB + e * 8); rb[e] = v.x; rb[e + 1] = v.y; }
    if (scale) {
#pragma unroll
      for (int e = 0; e < EB; e += 2) { const dbl2 v = *(gcptr2)(sV1 + voffK + e * 8); rs[e] = v.x; rs[e + 1] = v.y; }
    }
    if (rowdot) {
#pragma unroll
      for (int e = 0; e < EA; e += 2) { const dbl2 v = *(gcptr2)(sV2 + voffK + e * 8); rg[e] = v.x; rg[e + 1] = v.y; }
    }
    sA += GS_BK * 8; sB += GS_BK * 8; sV1 += GS_BK * 8; sV2 += GS_BK * 8;       // (scalar ALU)
  };
  const int wA = (8 * (s_k >> 2) + (s_r >> 4)) * GS_FRAG + (s_r & 15);
  const int wB = S::A_ELEMS + wA;
  auto store_tiles = [&](const int stage_off) {
    double* As = smem + stage_off + wA;
    double* Bs = smem + stage_off + wB;
    if (scale) {
#pragma unroll
      for (int e = 0; e < EB; e++) rb[e] *= rs[e];
    }
    if (rowdot) {
#pragma unroll
      for (int e = 0; e < EA; e++) udot = fma(ra[e], rg[e], udot);
    }
#pragma unroll
    for (int e = 0; e < EA; e++) As[(e >> 2) * 8 * GS_FRAG + (e & 3) * 16] = ra[e];
#pragma unroll
    for (int e = 0; e < EB; e++) Bs[(e >> 2) * 8 * GS_FRAG + (e & 3) * 16] = rb[e];
  };
  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
  const int rA = lane;
  const int rB0 = S::A_ELEMS + ct0 * GS_FRAG + lane, rB1 = S::A_ELEMS + ct1 * GS_FRAG + lane;
  // sym_tag = true_type: a diagonal output tile of the symmetric product — row tiles above a column tile are never read back
  // and are skipped (wave-uniform scalar branches); false_type: every MFMA tile, straight-line (6 of the 10 lower tiles)
  auto mfma_tile = [&](const int stage_off, auto sym_tag) {
    constexpr bool SYM = decltype(sym_tag)::value;
    const double* As = smem + stage_off + rA;
    const double* B0 = smem + stage_off + rB0;
    const double* B1 = smem + stage_off + rB1;
    __builtin_amdgcn_s_setprio(GS_MFMA_PRIO);
#pragma unroll
    for (int ks = 0; ks < GS_BK / 4; ks++) {
      double af[TM];
      const double bf0 = B0[8 * ks * GS_FRAG], bf1 = B1[8 * ks * GS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) af[a] = As[(8 * ks + a) * GS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) {
        if (!SYM || a >= cmin0) acc[a][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf0, acc[a][0], 0, 0, 0);
        if (!SYM || a >= cmin1) acc[a][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf1, acc[a][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto k_loop = [&](auto sym_tag) {
    load_tiles();
    store_tiles(0);
    __syncthreads();
    int it = 0;
    for (; it + 2 <= nkt; it += 2) {
      load_tiles();
      mfma_tile(0, sym_tag);
      store_tiles(S::STAGE);
      __syncthreads();
      const bool more = (it + 2 < nkt);
      if (more) load_tiles();
      mfma_tile(S::STAGE, sym_tag);
      if (more) store_tiles(0);
      __syncthreads();
    }
    if (it < nkt) {                        // odd count: the last K-tile sits in stage 0
      mfma_tile(0, sym_tag);
      __syncthreads();
    }
  };
  GS_STAMP(1);
  // (one instantiation: per-MFMA scalar branches on every tile measured the same 4.14 ms as separate plain / symmetric
  //  loops, which needed 17 spilled VGPRs)
  if (nkt > 0) k_loop(std::true_type{});
  GS_STAMP(2);
  // ---- epilogue: the K-slice's slab [ksl][M][N] --------------------------------------------------------------------
  {
    constexpr int TS = 34;
    double* tw = smem + wc * (32 * TS);
    const int srow = lane >> 4, sc = lane & 15;
    // lane's column pair inside the wavefront's two (non-adjacent) column tiles
    const int gcol = ((sc < 8) ? ct0 : ct1) * 16 + (sc & 7) * 2;
    const int lcol = ((sc < 8) ? 0 : 16) + (sc & 7) * 2;
    const uint32_t voffC = (uint32_t)(((int64_t)(i0 + srow) * p.N + j0 + gcol) * 8);
    const gcbytes slab = (gcbytes)p.o2 + (int64_t)ksl * p.M * p.N * 8;
    const int64_t rowstride = (int64_t)p.N * 8;
#pragma unroll
    for (int part = 0; part < 4; part++) {
#pragma unroll
      for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) tw[(a2 * 16 + kq + 4 * r) * TS + b * 16 + lc] = acc[part * 2 + a2][b][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const dbl2 v = *reinterpret_cast<const dbl2*>(tw + (4 * q + srow) * TS + lcol);
        const gbytes cb = (gbytes)gs_uniform(slab + (int64_t)(part * 32 + 4 * q) * rowstride);
        *(gptr2)(cb + voffC) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (rowdot) {      // the two threads that share a row hold disjoint k-ranges: combine, one partial per K-slice
      udot += __shfl_xor(udot, 1, 64);
      if ((tid & 1) == 0) ((gptr)p.o1)[(int64_t)ksl * p.M + i0 + s_r] = udot;
    }
  }
#ifdef GS_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  GS_STAMP(3);
  if (threadIdx.x == 0) {
    const unsigned int slot = atomicAdd(&gs_stamp_count, 1u);
    if (slot < 65536) {
      unsigned long long* o = gs_stamps + 6 * (size_t)slot;
      o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3]; o[4] = (unsigned long long)(400 + (diag_sym ? 1 : 0) + (tn == 0 ? 10 : 0)); o[5] = (unsigned long long)nkt;
    }
  }
#endif
}

bool launch_gemm_strip_nt_lean(gp_handle h, const GemmProblem* d_probs, int batch, int M, int Nlong, int nsplit, int sym,
                               int scale_by_k, gp_status* st) {
  const bool enabled = gp_switches().strip_lean != 0;
  if (!enabled || (M % GS_BM) != 0 || (Nlong % GS_BK) != 0 || nsplit < 2) return false;
  using S = StripSmem<false>;
  StripNtFlags nf;
  nf.tilesM = M / GS_BM; nf.ksplit = nsplit; nf.sym = sym; nf.scale = scale_by_k;
  const int ntl = sym ? nf.tilesM * (nf.tilesM + 1) / 2 : nf.tilesM * nf.tilesM;
  static std::atomic<uint32_t> attr_devs{0};
  const uint32_t bit = 1u << (h->device & 31);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_strip_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES);
    if (e != hipSuccess) { *st = gp_fail(h, GP_ERR_HIP, "hipFuncSetAttribute failed"); return true; }
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(gemm_strip_nt_kernel, dim3(ntl * nsplit, 1, batch), dim3(256), S::BYTES, h->stream, d_probs, nf);
  *st = (hipGetLastError() == hipSuccess) ? GP_OK : gp_fail(h, GP_ERR_HIP, "gemm_strip_nt_kernel launch failed");
  return true;
}

template <int TAG, int KT = -1>
static gp_status launch_strip(gp_handle h, const GemmProblem* d_probs, int batch, int M, int N, const GemmFlags& f) {
  using S = StripSmem<TAG == 2>;
  StripFlags sf;
  sf.tilesM = M / GS_BM; sf.tilesN = N / GS_BN; sf.tm0 = f.tile_m0; sf.epi = f.epilogue; sf.alpha = f.alpha; sf.xcols = f.aux_x;
  if (f.tile_m0 > 0 || f.tile_mcount > 0) {
    const int all = sf.tilesM;
    if (f.tile_m0 >= all) return GP_OK;
    sf.tilesM = (f.tile_mcount > 0 && f.tile_m0 + f.tile_mcount < all) ? f.tile_mcount : all - f.tile_m0;
  }
  static std::atomic<uint32_t> attr_devs{0};
  const uint32_t bit = 1u << (h->device & 31);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)gemm_strip_kernel<TAG, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::BYTES));
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  dim3 grid(sf.tilesM * sf.tilesN, 1, batch);
  hipLaunchKernelGGL((gemm_strip_kernel<TAG, KT>), grid, dim3(256), S::BYTES, h->stream, d_probs, sf);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// role 5 (Kuf_bar with the stationary hyper-gradient contraction as its epilogue) exists in the lean form only: whether a
// launch of that shape would take it (pdgp_backward asks before it decides to skip the separate contraction)
bool gemm_strip_fused_contraction_ok(int maxM, int maxN, int ktype) {
  const bool enabled = gp_switches().strip_lean != 0, fuse = gp_switches().hyper_fuse != 0;
  const bool stat = (ktype == GP_KERN_MATERN12 || ktype == GP_KERN_MATERN32 || ktype == GP_KERN_MATERN52 || ktype == GP_KERN_RBF);
  return enabled && fuse && stat && maxM > 0 && (maxM % GS_BM) == 0 && (maxN % GS_BN) == 0;
}

// Returns true when the lean form took the launch (*st = its status); false: run gemm.hip's kernel.  `uniform` = the caller
// vouches that every problem of the batch has M = K-structure maxM, N = maxN, 16-byte aligned operands, even leading dimensions.
bool launch_gemm_strip_lean(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f,
                            gp_status* st) {
  const bool enabled = gp_switches().strip_lean != 0;
  if (!enabled || !f.uniform_aligned || f.role < 1 || (f.role > 3 && f.role != 5)) return false;
  if ((maxM % GS_BM) != 0 || (maxN % GS_BN) != 0 || f.beta != 0.0 || f.triC != TRI_NONE) return false;
  if (f.role >= 3 ? !(f.alpha == 1.0 || f.alpha == 2.0 || f.alpha == 0.5 || f.alpha == 4.0) : (f.alpha != 1.0)) return false;
  if (f.role >= 3 && f.scale_mode != 1) return false;
  if (f.role == 5) {       // (the caller has asked gemm_strip_fused_contraction_ok first: it cannot fall back from here)
    if (!f.aux_x || f.tile_m0 || f.tile_mcount) { *st = gp_fail(h, GP_ERR_BAD_ARG, "fused Kuf_bar contraction: bad launch"); return true; }
    switch (f.aux_ktype) {
      case GP_KERN_MATERN12: *st = launch_strip<5, GP_KERN_MATERN12>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_MATERN32: *st = launch_strip<5, GP_KERN_MATERN32>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_MATERN52: *st = launch_strip<5, GP_KERN_MATERN52>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_RBF: *st = launch_strip<5, GP_KERN_RBF>(h, d_probs, batch, maxM, maxN, f); break;
      default: *st = gp_fail(h, GP_ERR_BAD_ARG, "fused Kuf_bar contraction: not a stationary kernel");
    }
    return true;
  }
  if (f.role == 1) *st = launch_strip<1>(h, d_probs, batch, maxM, maxN, f);
  else if (f.role == 2) *st = launch_strip<2>(h, d_probs, batch, maxM, maxN, f);
  else *st = launch_strip<3>(h, d_probs, batch, maxM, maxN, f);
  return true;
}
