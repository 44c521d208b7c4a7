// gemm_strip_f32.hip — float32 forms of gemm_strip.hip's lean strip products (gfx950, round 3): the four O(M^2 N) products
// of the conditional and its backward pass with the M x N strips STORED as float32 and v_mfma_f32_16x16x4_f32 doing the
// arithmetic (BASELINE configs 3 and 5 are quoted at fp32; the reference's dtype is a setting, gpitch/pdgp.py:13).
//
// What is float32 / what stays float64 is gemm_f32.hip's contract, unchanged: the M x M operands (W = Lm^-1, Lq, R) are
// float64 in memory and rounded to float32 as they are staged; every reduction over the inducing index (sum A^2, A^T q_mu,
// sum LTA^2) and over the frames (split-K slabs, A gm) is accumulated in float64.  Results are bit-identical to
// gemm_f32.hip's (same K-tile of 32, same MFMA order, same reduction order).
//
// Why a second form: a float32 MFMA runs at the float32 VECTOR rate — like the float64 one it occupies the SIMD's vector
// ALU while it executes — and a float32 K-tile is half as long as a float64 one against the same staging work, so the
// ~150 vector instructions per K-tile of gemm_f32.hip's loop (run-time LDS stage, 64-bit addresses, masks on every tile)
// cost it twice what they cost the float64 kernel (0.34-0.37 of the float32 matrix peak in round 2).  Structure as
// gemm_strip.hip: fragment-major LDS stages (64 floats per MFMA operand fragment, every fragment read = one per-lane base
// + an immediate), K loop unrolled over the two stages, scalar-base + constant-lane-offset operand loads, masks only in
// the diagonal block, LDS-transposed 16-byte epilogue stores.  Whole aligned tiles only; gemm_f32.hip runs otherwise.
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef const char __attribute__((address_space(1))) * gcbytes;
typedef char __attribute__((address_space(1))) * gbytes;
typedef const double __attribute__((address_space(1))) * gcptr;
typedef double __attribute__((address_space(1))) * gptr;
typedef const dbl2 __attribute__((address_space(1))) * gcptr2;
typedef dbl2 __attribute__((address_space(1))) * gptr2;
typedef const f4 __attribute__((address_space(1))) * gcfptr4;
typedef f4 __attribute__((address_space(1))) * gfptr4;

#define FS_BK 32
#define FS_BT 128
#define FS_FRAG 64                       // floats per MFMA operand fragment: lane = 16 (k & 3) + (row or column & 15)
#define FS_TILE (FS_BT * FS_BK)          // floats per operand tile: fragment (ks = k / 4, t = row or column tile) at (8 ks + t) * 64
#define FS_STAGE (2 * FS_TILE)
#define FS_V0_OFF (2 * FS_STAGE)         // (floats) 128 doubles behind the stages: v0[i0 .. i0 + 127]
#ifndef FS_PAD_BYTES
#define FS_PAD_BYTES 0                // occupancy probe: extra dynamic LDS per workgroup (one workgroup per CU above 16 KiB)
#endif
#define FS_BYTES ((size_t)(2 * FS_STAGE) * sizeof(float) + FS_BT * sizeof(double) + FS_PAD_BYTES)

struct Strip32Flags {
  int tilesM, tilesN, tm0;
  int epi;
  float alpha;        // role 3 only, a power of two: folded into the column scales (exact)
};

__device__ __forceinline__ gcbytes fs_uniform(gcbytes p) {
  const uint64_t b = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (gcbytes)(((uint64_t)hi << 32) | lo);
}

// TAG 1: op(A) = W (lower), k-contiguous float64.  TAG 2: op(A) = Lq^T (upper; Lq read row-wise), K walked downwards.
// TAG 3: op(A) = R dense, B(k, n) *= v1[n].  B, C: float32 strips.
// BK: K-tile depth, 32 (two workgroups per CU: 64 KiB of LDS stages each) or 16 (32 KiB and <= 168 VGPRs: three per CU — the
// short K loops of the triangular products at M = 256 / 512 are latency, and a third resident workgroup hides more of it
// than the shorter K-tile costs.  Same k-step order for roles 1 and 3 (bit-identical results); role 2 walks its K-tiles
// downwards, so its sums are taken in another order: float32 rounding, inside the stated tolerance)
// (The contracting form of the dense product — gemm_f32.hip's — was also built on this kernel and measured: its launch is
// shorter, 1.51 vs 1.86 ms on cfg3, and the step LONGER, 4.40 vs 4.11 ms: that launch shares the device with the split-K product
// and the M x M chain on the helper stream, and three resident workgroups of it per CU leave them no room.  Not kept.)
template <int TAG, int BK>
__global__ void __launch_bounds__(256, (BK == 16) ? 3 : 2) gemm_strip_f32_kernel(const GemmProblem* __restrict__ probs, Strip32Flags f) {
  constexpr int NH = BK / 16;                // 16-row halves of a K-tile
  constexpr int EA = BK / 2;                 // k-contiguous A: elements per thread
  constexpr int KTILE = FS_BT * BK, KSTAGE = 2 * KTILE, KV0_OFF = 2 * KSTAGE;
  constexpr bool TA = (TAG == 2);
  constexpr bool KDOWN = (TAG == 2);
  constexpr int TRI = (TAG == 1) ? TRI_LOWER : (TAG == 2) ? TRI_UPPER : TRI_NONE;
  constexpr int TM = 8, TN = 2, NKS = BK / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
  const int tn = bid / f.tilesM;
  const int tm = f.tm0 + (bid % f.tilesM + tn) % f.tilesM;
  const int i0 = tm * FS_BT, j0 = tn * FS_BT;
  if (i0 >= p.M || j0 >= p.N) return;
  int kbeg = 0, kend = p.K;
  if (TRI == TRI_LOWER) kend = min(kend, i0 + FS_BT);
  if (TRI == TRI_UPPER) kbeg = max(kbeg, i0);
  const int nkt = (kend - kbeg) / BK;
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  double* v0s = reinterpret_cast<double*>(smem + KV0_OFF);
  if ((f.epi & EPI_COLDOT) && tid < FS_BT) v0s[tid] = ((gcptr)p.v0)[i0 + tid];

  // ---- staging maps ------------------------------------------------------------------------------------------------
  // A, k-contiguous (TAG 1, 3): thread -> row a_i = tid / 2, sixteen consecutive k from a_k = 16 (tid & 1)
  // A, row-wise     (TAG 2)   : thread -> k rows a_k = tid / 16 and a_k + 16, row pairs a_i + 32 q, a_i = 2 (tid & 15)
  // B (float32, row-contiguous): thread -> k rows b_k = tid / 16 and b_k + 16, column quads b_n + 64 q, b_n = 4 (tid & 15)
  const int a_i = TA ? (tid & 15) * 2 : (tid >> 1);
  const int a_k = TA ? (tid >> 4) : (tid & 1) * EA;
  const int b_k = tid >> 4, b_n = (tid & 15) * 4;
  const uint32_t voffA = TA ? (uint32_t)(((int64_t)a_k * p.lda + i0 + a_i) * 8) : (uint32_t)(((int64_t)(i0 + a_i) * p.lda + a_k) * 8);
  const uint32_t voffB = (uint32_t)(((int64_t)b_k * p.ldb + j0 + b_n) * 4);
  const int kfirst = KDOWN ? kend - BK : kbeg;
  const int64_t stepA = (KDOWN ? -1 : 1) * (TA ? (int64_t)BK * p.lda * 8 : (int64_t)BK * 8);
  const int64_t stepB = (KDOWN ? -1 : 1) * (int64_t)BK * p.ldb * 4;
  gcbytes sA = fs_uniform((gcbytes)p.A + (TA ? (int64_t)kfirst * p.lda * 8 : (int64_t)kfirst * 8));
  gcbytes sA2 = fs_uniform((gcbytes)p.A + (TA ? (int64_t)(kfirst + 16) * p.lda * 8 : 0));     // second k row (TAG 2)
  gcbytes sB = fs_uniform((gcbytes)p.B + (int64_t)kfirst * p.ldb * 4);
  gcbytes sB2 = fs_uniform((gcbytes)p.B + (int64_t)(kfirst + 16) * p.ldb * 4);
  double ra[8 * NH];
  f4 rbX[2 * NH], rbY[2 * NH];
  float rs[8];
  if (TAG == 3) {
    const gcptr gv1 = (gcptr)p.v1;
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int c = 0; c < 4; c++) rs[q * 4 + c] = f.alpha * (float)gv1[j0 + b_n + 64 * q + c];     // (alpha = 2^k: exact)
  }
  auto load_A = [&]() {
    if (TA) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const dbl2 u = *(gcptr2)(sA + voffA + q * 256);
        ra[2 * q] = u.x; ra[2 * q + 1] = u.y;
        if (NH == 2) { const dbl2 v = *(gcptr2)(sA2 + voffA + q * 256); ra[8 * (NH - 1) + 2 * q] = v.x; ra[8 * (NH - 1) + 2 * q + 1] = v.y; }
      }
      sA2 = (gcbytes)((int64_t)sA2 + stepA);
    } else {
#pragma unroll
      for (int e = 0; e < EA; e += 2) { const dbl2 v = *(gcptr2)(sA + voffA + e * 8); ra[e] = v.x; ra[e + 1] = v.y; }
    }
    sA = (gcbytes)((int64_t)sA + stepA);
  };
  auto load_B = [&](f4 (&rb)[2 * NH]) {
#pragma unroll
    for (int q = 0; q < 2; q++) { rb[q] = *(gcfptr4)(sB + voffB + q * 256); if (NH == 2) rb[2 * (NH - 1) + q] = *(gcfptr4)(sB2 + voffB + q * 256); }
    sB = (gcbytes)((int64_t)sB + stepB);
    sB2 = (gcbytes)((int64_t)sB2 + stepB);
  };
  // LDS write bases (floats) of stage 0
  const int wA = TA ? ((8 * (a_k >> 2) + (a_i >> 4)) * FS_FRAG + 16 * (a_k & 3) + (a_i & 15))
                    : ((8 * (a_k >> 2) + (a_i >> 4)) * FS_FRAG + (a_i & 15));
  const int wB = KTILE + (8 * (b_k >> 2) + (b_n >> 4)) * FS_FRAG + 16 * (b_k & 3) + (b_n & 15);
  auto store_tiles = [&](const int stage_off, int kt, auto mask_tag, f4 (&rb)[2 * NH]) {
    constexpr bool MASK = decltype(mask_tag)::value;
    float* As = smem + stage_off + wA;
    float* Bs = smem + stage_off + wB;
    if (TA) {
      // element (row pair i = a_i + 32 q, k row a_k [+ 16]): non-zero iff k >= i
#pragma unroll
      for (int h = 0; h < NH; h++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float v0 = (float)ra[8 * h + 2 * q], v1 = (float)ra[8 * h + 2 * q + 1];
          if (MASK && kt < i0 + FS_BT) {
            const int k = kt + a_k + 16 * h, i = i0 + a_i + 32 * q;
            if (k < i) v0 = 0.f;
            if (k < i + 1) v1 = 0.f;
          }
          *reinterpret_cast<float2*>(As + h * 4 * 8 * FS_FRAG + q * 2 * FS_FRAG) = make_float2(v0, v1);
        }
    } else {
#pragma unroll
      for (int e = 0; e < EA; e++) {
        float v = (float)ra[e];
        if (MASK && TRI == TRI_LOWER && kt + BK > i0) { if (kt + a_k + e > i0 + a_i) v = 0.f; }
        As[(e >> 2) * 8 * FS_FRAG + (e & 3) * 16] = v;
      }
    }
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        f4 v = rb[2 * h + q];
        if (TAG == 3) { v.x *= rs[q * 4]; v.y *= rs[q * 4 + 1]; v.z *= rs[q * 4 + 2]; v.w *= rs[q * 4 + 3]; }
        *reinterpret_cast<f4*>(Bs + h * 4 * 8 * FS_FRAG + q * 4 * FS_FRAG) = v;
      }
  };
  f4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int rA = lane, rB = KTILE + 2 * wc * FS_FRAG + lane;
  auto mfma_full = [&](const int stage_off) {
    const float* As = smem + stage_off + rA;
    const float* Bs = smem + stage_off + rB;
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      float af[TM], bf[TN];
#pragma unroll
      for (int b = 0; b < TN; b++) bf[b] = Bs[(8 * ks + b) * FS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) af[a] = As[(8 * ks + a) * FS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto mfma_diag = [&](const int stage_off, int kt) {
    const float* As = smem + stage_off + rA;
    const float* Bs = smem + stage_off + rB;
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      const int kg = kt + ks * 4;
      int a_lo = 0, a_hi = TM;
      if (TRI == TRI_LOWER) a_lo = max(0, (kg - i0) >> 4);
      if (TRI == TRI_UPPER) a_hi = min(TM, ((kg + 3 - i0) >> 4) + 1);
      float af[TM], bf[TN];
#pragma unroll
      for (int b = 0; b < TN; b++) bf[b] = Bs[(8 * ks + b) * FS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) af[a] = As[(8 * ks + a) * FS_FRAG];
#pragma unroll
      for (int a = 0; a < TM; a++) {
        if (a >= a_lo && a < a_hi) {
#pragma unroll
          for (int b = 0; b < TN; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // ---- K loop (gemm_strip.hip's): plain pairs with literal stages and no masks, then the diagonal block ------------------
  {
    const int kstep = KDOWN ? -BK : BK;
    int nplain = nkt;
    if (TRI == TRI_LOWER) nplain = i0 / BK;
    if (TRI == TRI_UPPER) nplain = (kend - (i0 + FS_BT)) / BK;
    const int npair = (max(0, min(nplain, nkt)) / 2) * 2;
    const std::true_type masked{};
    const std::false_type plain{};
    int kt = kfirst;
    load_A();
    load_B(rbX);
    if (npair > 0) store_tiles(0, kt, plain, rbX); else store_tiles(0, kt, masked, rbX);
    __syncthreads();
    if (nkt > 1) load_B(rbY);
    int it = 0;
    for (; it + 2 < npair; it += 2) {
      load_A();
      load_B(rbX);
      mfma_full(0);
      store_tiles(KSTAGE, kt + kstep, plain, rbY);
      __syncthreads();
      kt += kstep;
      load_A();
      if (it + 3 < nkt) load_B(rbY);
      mfma_full(KSTAGE);
      store_tiles(0, kt + kstep, plain, rbX);
      __syncthreads();
      kt += kstep;
    }
    for (; it < nkt; it += 2) {          // (nkt is a multiple of 4: M, K multiples of 128)
      load_A();
      if (it + 2 < nkt) load_B(rbX);
      mfma_diag(0, kt);
      store_tiles(KSTAGE, kt + kstep, masked, rbY);
      __syncthreads();
      kt += kstep;
      const bool more = (it + 2 < nkt);
      if (more) load_A();
      if (it + 3 < nkt) load_B(rbY);
      mfma_diag(KSTAGE, kt);
      if (more) store_tiles(0, kt + kstep, masked, rbX);
      __syncthreads();
      kt += kstep;
    }
  }
  // ---- epilogue: accumulator element r of tile (a, b): row 16 a + 4 kq + r, column 16 b + lc of the wavefront's 128 x 32 ----
  if (f.epi & EPI_STORE) {
    constexpr int TS = 36;
    float* tw = smem + wc * (32 * TS);
    const int srow = lane >> 3, scol = (lane & 7) * 4;
    const uint32_t voffC = (uint32_t)(((int64_t)(i0 + srow) * p.ldc + j0 + wc * 32 + scol) * 4);
    const int64_t rowstride = p.ldc * 4;
#pragma unroll
    for (int part = 0; part < 4; part++) {
#pragma unroll
      for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) tw[(a2 * 16 + 4 * kq + r) * TS + b * 16 + lc] = acc[part * 2 + a2][b][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f4 v = *reinterpret_cast<const f4*>(tw + (8 * q + srow) * TS + scol);
        const gbytes cb = (gbytes)fs_uniform((gcbytes)p.C + (int64_t)(part * 32 + 8 * q) * rowstride);
        *(gfptr4)(cb + voffC) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) {
    const double* vv = v0s + 4 * kq;
    const gptr go0 = (gptr)p.o0, go1 = (gptr)p.o1;
#pragma unroll
    for (int b = 0; b < TN; b++) {
      double s2 = 0.0, sd = 0.0;
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double v = (double)acc[a][b][r];
          s2 = fma(v, v, s2);
          if (f.epi & EPI_COLDOT) sd = fma(v, vv[a * 16 + r], sd);
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
}

template <int TAG, int BK>
static gp_status launch_strip32(gp_handle h, const GemmProblem* d_probs, int batch, int M, int N, const GemmFlags& f) {
  constexpr size_t BYTES = (size_t)(4 * FS_BT * BK) * sizeof(float) + FS_BT * sizeof(double) + FS_PAD_BYTES;
  Strip32Flags sf;
  sf.tilesM = M / FS_BT; sf.tilesN = N / FS_BT; sf.tm0 = f.tile_m0; sf.epi = f.epilogue; sf.alpha = (float)f.alpha;
  if (f.tile_m0 > 0 || f.tile_mcount > 0) {
    const int all = sf.tilesM;
    if (f.tile_m0 >= all) return GP_OK;
    sf.tilesM = (f.tile_mcount > 0 && f.tile_m0 + f.tile_mcount < all) ? f.tile_mcount : all - f.tile_m0;
  }
  static std::atomic<uint32_t> attr_devs{0};
  const uint32_t bit = 1u << (h->device & 31);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)gemm_strip_f32_kernel<TAG, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BYTES));
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_strip_f32_kernel<TAG, BK>), dim3(sf.tilesM * sf.tilesN, 1, batch), dim3(256), BYTES, h->stream, d_probs, sf);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

bool launch_gemm_strip_f32_lean(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f,
                                gp_status* st) {
  const bool enabled = gp_switches().strip_lean != 0;
  // MEASURED (same box, headline shape, overlap 0, against gemm_f32.hip's 8-wavefront kernels): A = W Kuf 2.46 -> 2.41 ms,
  // Lq^T A 2.13 -> 2.00, Kuf_bar 3.59 -> 3.73 (slower), split-K product 2.57 -> 2.58: the float32 products are not bound by
  // the K loop's vector instructions the way the float64 ones are.  With 16-deep K-tiles and three workgroups per CU (below)
  // the dense product's lean form passes the 8-wavefront kernel too (cfg3 4.12 -> 4.10 ms, headline shape 12.75 -> 12.63):
  // roles 1-3 take the lean form, the split-K product stays.
  if (!enabled || !f.uniform_aligned || f.role < 1 || f.role > 3) return false;
  if ((maxM % FS_BT) != 0 || (maxN % FS_BT) != 0 || f.beta != 0.0 || f.triC != TRI_NONE) return false;
  if (f.role == 3 ? !(f.alpha == 1.0 || f.alpha == 2.0 || f.alpha == 0.5 || f.alpha == 4.0) : (f.alpha != 1.0)) return false;
  if (f.role == 3 && f.scale_mode != 1) return false;
  // K-tile depth 16: three workgroups per CU (32-deep tiles, two per CU, measured 5 % behind on cfg3: DESIGN.md section 3b)
  if (f.role == 1) *st = launch_strip32<1, 16>(h, d_probs, batch, maxM, maxN, f);
  else if (f.role == 2) *st = launch_strip32<2, 16>(h, d_probs, batch, maxM, maxN, f);
  else *st = launch_strip32<3, 16>(h, d_probs, batch, maxM, maxN, f);
  return true;
}
