// gemm_res_f32.hip — float32 strip products for SMALL inducing sets (M <= 256) with the M x M operand RESIDENT in LDS
// (gfx950, round 3).  BASELINE configs[2] (12 pitches, M = 256 per pitch, fp32) is the case: with K = M = 256 a
// 128 x 128 output tile of gemm_f32.hip / gemm_strip_f32.hip lives for only 8 K-tiles, its prologue (first operand round
// trip) and epilogue are a third of its life, and the products ran at 0.31-0.43 of the float32 matrix peak.
//
// Here a workgroup (8 wavefronts, ONE per CU: the LDS is the resource) takes a 128-row block of op(A) — W = Lm^-1
// (gpitch/pdgp.py:147 conditional: matrix_triangular_solve as a product), Lq^T, or R = W^T (Lq Lq^T - I) of the backward
// pass — converts it to float32 ONCE into LDS (128 rows x K <= 256: 128 KiB, already in MFMA fragment order) and then
// streams the float32 strip B through a 3-slot ring of 16-row K-tiles (24 KiB) for a whole run of column tiles: the
// stream never stops at a tile boundary (the next column tile's first K-tiles are in flight while the current one
// finishes), there is no A staging at all, and the epilogue of a column tile is the only thing between two K loops.
//   wavefront w owns all 128 rows x columns [16 w, 16 w + 16) of the 128-column tile: 8 accumulators of
//   v_mfma_f32_16x16x4_f32, so the column reductions (sum A^2, A^T q_mu, sum LTA^2) finish inside one wavefront exactly
//   as in the other strip kernels (row-block partials, fixed order, float64).
// LDS fragment order: A: [k-step][row-tile half][lane][row tile & 3] so one ds_read_b128 fetches four row tiles' operands;
//                     B slot: [column tile][lane][k-step & 3] so one ds_read_b128 fetches a K-tile's four k-steps.
// What is float32 / float64 is gemm_f32.hip's contract; the role-3 column scale (2 gv, folded alpha) is applied to the
// OUTPUT columns here ((R A) D instead of R (A D): same product, one float32 rounding placed differently).
#include "common.h"
#include <stdlib.h>
#include <atomic>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const char __attribute__((address_space(1))) * gcbytes;
typedef char __attribute__((address_space(1))) * gbytes;
typedef const double __attribute__((address_space(1))) * gcptr;
typedef double __attribute__((address_space(1))) * gptr;
typedef const f4 __attribute__((address_space(1))) * gcfptr4;
typedef float __attribute__((address_space(1))) * gfptr;

#ifndef RS_EXP
#define RS_EXP 0      // measurement variants (tools/build_variant.sh): 1 no stream loads, 2 no barrier, 3 no MFMAs
#endif
#define RS_BT 128
#define RS_BK 16
#define RS_RING 3
#define RS_THREADS 512
#define RS_MAXK 256
#define RS_SLOT 2048                                  // floats per ring slot: 8 column tiles x 64 lanes x 4 k-steps
#define RS_A_FLOATS (RS_MAXK * RS_BT)                 // 32768
#define RS_BYTES ((size_t)(RS_A_FLOATS + RS_RING * RS_SLOT) * sizeof(float) + RS_BT * sizeof(double))

struct ResFlags {
  int tilesM, tm0;          // row-blocks of this launch: [tm0, tm0 + tilesM)
  int tilesN, chunk_tiles;  // column tiles of the strip, column tiles per workgroup
  int chunks;
  int epi;
  float alpha;              // role 3: folded into the column scales
};

__device__ __forceinline__ gcbytes rs_uniform(gcbytes p) {
  const uint64_t b = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (gcbytes)(((uint64_t)hi << 32) | lo);
}

// TAG 1: op(A) = W (lower).  TAG 2: op(A) = Lq^T (upper; Lq read row-wise).  TAG 3: op(A) = R dense, C(i, n) *= alpha v1[n].
template <int TAG>
__global__ void __launch_bounds__(RS_THREADS) gemm_res_f32_kernel(const GemmProblem* __restrict__ probs, ResFlags f) {
  constexpr int TRI = (TAG == 1) ? TRI_LOWER : (TAG == 2) ? TRI_UPPER : TRI_NONE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ares = smem;
  float* Bring = smem + RS_A_FLOATS;
  double* v0s = reinterpret_cast<double*>(Bring + RS_RING * RS_SLOT);
  const GemmProblem p = probs[blockIdx.z];
  // item = (row-block, run of column tiles).  Workgroups are dealt round-robin to the 8 XCDs: the row-blocks of one run
  // (same B columns) are given ids 8 apart, i.e. the same L2
  int bid = blockIdx.x, rbi, chunk;
  if ((f.chunks & 7) == 0) { rbi = (bid >> 3) % f.tilesM; chunk = (bid & 7) + 8 * (bid / (8 * f.tilesM)); }
  else { rbi = bid % f.tilesM; chunk = bid / f.tilesM; }
  const int tm = f.tm0 + rbi, i0 = tm * RS_BT;
  const int ct0 = chunk * f.chunk_tiles;
  const int nct = min(f.chunk_tiles, f.tilesN - ct0);
  if (i0 >= p.M || nct <= 0) return;
  int kbeg = 0, kend = p.K;
  if (TRI == TRI_LOWER) kend = min(kend, i0 + RS_BT);
  if (TRI == TRI_UPPER) kbeg = max(kbeg, i0);
  const int KR = kend - kbeg;
  const int nk = KR / RS_BK;                  // K-tiles per column tile
  const int G = nct * nk;                     // K-tiles of this workgroup's stream
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kq = lane >> 4;

  // ---- the resident operand: rows [i0, i0 + 128) of op(A), k in [kbeg, kend), float64 -> float32, structural zeros made
  //      explicit (q_sqrt's upper triangle is free parameter space: GPflow takes band_part(-1, 0)) -----------------------
  {
    const gcptr gA = (gcptr)p.A;
    auto put = [&](int r, int kr, float v) {
      const int ks = kr >> 2, a = r >> 4, ln = 16 * (kr & 3) + (r & 15);
      Ares[((ks * 2 + (a >> 2)) * 64 + ln) * 4 + (a & 3)] = v;
    };
    // KR is 128 or 256 (M in {128, 256}): index arithmetic by shifts; sixteen loads in flight per thread (one dependent
    // round trip per element made this fill as long as the item's whole stream)
    const int lg = (KR == 256) ? 8 : 7;
    const int per = (KR * RS_BT) / RS_THREADS;          // 32 or 64 elements per thread
    if (TAG == 2) {      // op(A)(r, k) = Lq(k, i0 + r): consecutive threads -> consecutive r
      const int r = tid & 127, kr0 = tid >> 7;
      for (int i = 0; i < per; i += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = gA[(int64_t)(kbeg + kr0 + 4 * (i + u)) * p.lda + i0 + r];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int kr = kr0 + 4 * (i + u);
          put(r, kr, (kbeg + kr < i0 + r) ? 0.f : (float)v[u]);
        }
      }
    } else {             // row-major, k contiguous: consecutive threads -> consecutive k
      const int kr = tid & (KR - 1), r0 = tid >> lg, rstep = RS_THREADS >> lg;
      for (int i = 0; i < per; i += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = gA[(int64_t)(i0 + r0 + rstep * (i + u)) * p.lda + kbeg + kr];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const int r = r0 + rstep * (i + u);
          put(r, kr, (TRI == TRI_LOWER && kbeg + kr > i0 + r) ? 0.f : (float)v[u]);
        }
      }
    }
    if ((f.epi & EPI_COLDOT) && tid < RS_BT) v0s[tid] = ((gcptr)p.v0)[i0 + tid];
  }

  // ---- B stream: thread -> column bc of the tile and the four k rows kk, kk + 4, kk + 8, kk + 12 of a K-tile: four dword
  //      loads (a wavefront reads 256 contiguous bytes of a row) and ONE 16-byte LDS write — the ring slot is
  //      [column tile][lane = 16 (k & 3) + (column & 15)][k-step], so this thread's four values are adjacent ---------------
  const int bc = tid & 127, kk = tid >> 7;
  const uint32_t voffB = (uint32_t)(((int64_t)kk * p.ldb + bc) * 4);
  const int wB = ((bc >> 4) * 64 + 16 * kk + (bc & 15)) * 4;                               // floats inside a slot
  const int64_t rowB4 = (int64_t)4 * p.ldb * 4;                                          // four k rows further
  const int64_t stepB = (int64_t)RS_BK * p.ldb * 4;
  const int64_t wrapB = (int64_t)RS_BT * 4 - (int64_t)nk * stepB;                       // back to the first K-tile, next column tile
  gcbytes sB = rs_uniform((gcbytes)p.B + ((int64_t)kbeg * p.ldb + (int64_t)ct0 * RS_BT) * 4);
  int gl = 0, gl_kt = 0;                       // next K-tile of the stream to request, its index inside the column tile
  // (an unconditional load — past the end of the stream the last K-tile is simply read again: a load inside a branch
  // makes every later wait a wait for ALL outstanding loads, and the stream's latency budget is three iterations)
  auto request = [&](f4& rb) {
    uint32_t vo = voffB;
    asm volatile("" : "+v"(vo));
    typedef const float __attribute__((address_space(1))) * gcfptr;
#if RS_EXP == 1
    gl++; return;      // measurement variant: no stream loads (results are garbage)
#endif
    rb.x = *(gcfptr)(sB + vo);
    rb.y = *(gcfptr)(rs_uniform(sB + rowB4) + vo);
    rb.z = *(gcfptr)(rs_uniform(sB + 2 * rowB4) + vo);
    rb.w = *(gcfptr)(rs_uniform(sB + 3 * rowB4) + vo);
    gl++;
    if (gl < G) {
      sB = (gcbytes)((int64_t)sB + stepB);
      gl_kt++;
      if (gl_kt == nk) { gl_kt = 0; sB = (gcbytes)((int64_t)sB + wrapB); }
    }
  };
  auto deposit = [&](const f4& rb, int slot) {
    *reinterpret_cast<f4*>(Bring + slot * RS_SLOT + wB) = rb;
  };
  // role 3: the sixteen column scales of this wavefront's columns come through the scalar cache (constant address space:
  // 2 gv is not written while this kernel runs), one column tile ahead, and are dealt to the lanes with selects
  typedef const double __attribute__((address_space(4))) * ccptr;
  auto column_scale = [&](int ctile) -> float {
    float sc = 1.f;
    if (TAG == 3) {
      const ccptr cv = (ccptr)p.v1 + ((int64_t)min(ctile, f.tilesN - 1) * RS_BT + 16 * w);
      double dv[16], d = 0.0;
#pragma unroll
      for (int c = 0; c < 16; c++) dv[c] = cv[c];
#pragma unroll
      for (int c = 0; c < 16; c++) asm volatile("" ::"s"(dv[c]));     // all sixteen loaded here, not inside the selects
#pragma unroll
      for (int c = 0; c < 16; c++) d = (lc == c) ? dv[c] : d;
      sc = f.alpha * (float)d;
    }
    return sc;
  };
  f4 rb0 = {0.f, 0.f, 0.f, 0.f}, rb1 = rb0, rb2 = rb0;
  {   // tiles 0, 1 -> slots 0, 1; tiles 2, 3, 4 -> the three register buffers
    f4 t0 = rb0, t1 = rb0;
    request(t0); request(t1);
    request(rb0); request(rb1); request(rb2);
    deposit(t0, 0); deposit(t1, 1);
  }
  float sc = column_scale(ct0);
  f4 acc[8];
#pragma unroll
  for (int a = 0; a < 8; a++) acc[a] = f4{0.f, 0.f, 0.f, 0.f};
  const float* Al = Ares + lane * 4;                          // + (ks * 2 + half) * 256 floats
  const float* Bl = Bring + (w * 64 + lane) * 4;              // + slot * RS_SLOT
  int kt = 0, ct = 0, slot = 0, wslot = 2;
  // one K-tile; `rb` = the register buffer that holds K-tile t + 2 (deposited now) and receives K-tile t + 5
  auto body = [&](f4& rb) {
#if RS_EXP != 2
    __syncthreads();
#endif
    // operands of K-tile t: four k-steps of the resident block, one fragment quad of the ring
    const f4 bq = *reinterpret_cast<const f4*>(Bl + slot * RS_SLOT);
    f4 af[4][2];
    {
      const float* Ak = Al + kt * (4 * 2 * 256);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        af[ks][0] = *reinterpret_cast<const f4*>(Ak + (ks * 2 + 0) * 256);
        af[ks][1] = *reinterpret_cast<const f4*>(Ak + (ks * 2 + 1) * 256);
      }
    }
    deposit(rb, wslot);
    request(rb);
    wslot = (wslot == RS_RING - 1) ? 0 : wslot + 1;
    // structural zeros of the triangular operands: 16-row tiles outside [alo, ahi] are zero for this whole K-tile
    int alo = 0, ahi = 7;
    {
      const int krel = kbeg + kt * RS_BK - i0;
      if (TRI == TRI_LOWER && krel >= 0) alo = krel >> 4;
      if (TRI == TRI_UPPER) ahi = min(7, (krel + 3) >> 4);
    }
    const float bqv[4] = {bq.x, bq.y, bq.z, bq.w};
    if (RS_EXP == 3) { acc[0][0] += af[0][0][0] * bqv[0] + af[3][1][3] * bqv[3]; }
    else if (TRI == TRI_NONE || (alo == 0 && ahi == 7)) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int a = 0; a < 8; a++)
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][a >> 2][a & 3], bqv[ks], acc[a], 0, 0, 0);
    } else {
#pragma unroll
      for (int a = 0; a < 8; a++)
        if (a >= alo && a <= ahi) {
#pragma unroll
          for (int ks = 0; ks < 4; ks++)
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][a >> 2][a & 3], bqv[ks], acc[a], 0, 0, 0);
        }
    }
    slot = (slot == RS_RING - 1) ? 0 : slot + 1;
    kt++;
    if (kt == nk) {
      // ---- epilogue of column tile ct0 + ct: element r of accumulator a = row 16 a + 4 kq + r, column 16 w + lc -------
      kt = 0;
      const int j0 = (ct0 + ct) * RS_BT;
      const int j = j0 + 16 * w + lc;
      const float scn = column_scale(ct0 + ct + 1);     // (requested first: consumed after the stores below)
      if (f.epi & EPI_STORE) {
        uint32_t vo = (uint32_t)(((int64_t)(4 * kq) * p.ldc + lc) * 4);
        asm volatile("" : "+v"(vo));
        const gcbytes cb0 = (gcbytes)p.C + ((int64_t)i0 * p.ldc + j0 + 16 * w) * 4;
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const gbytes cb = (gbytes)rs_uniform(cb0 + (int64_t)(16 * a + r) * p.ldc * 4);
            *(gfptr)(cb + vo) = (TAG == 3) ? acc[a][r] * sc : acc[a][r];
          }
      }
      if (f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) {
        const double* vv = v0s + 4 * kq;
        double s2 = 0.0, sd = 0.0;
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double v = (double)((TAG == 3) ? acc[a][r] * sc : acc[a][r]);
            s2 = fma(v, v, s2);
          }
        if (f.epi & EPI_COLDOT) {
#pragma unroll
          for (int a = 0; a < 8; a++)
#pragma unroll
            for (int r = 0; r < 4; r++) sd = fma((double)((TAG == 3) ? acc[a][r] * sc : acc[a][r]), vv[a * 16 + r], sd);
        }
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
        if (kq == 0) {
          if (f.epi & EPI_COLSUMSQ) ((gptr)p.o0)[(int64_t)tm * p.N + j] = s2;
          if (f.epi & EPI_COLDOT) ((gptr)p.o1)[(int64_t)tm * p.N + j] = sd;
        }
      }
#pragma unroll
      for (int a = 0; a < 8; a++) acc[a] = f4{0.f, 0.f, 0.f, 0.f};
      sc = scn;
      ct++;
    }
  };
  for (int t = 0; t < G;) {       // unrolled over the three register buffers: each keeps its identity, no copies
    body(rb0); if (++t >= G) break;
    body(rb1); if (++t >= G) break;
    body(rb2); ++t;
  }
}

template <int TAG>
static gp_status launch_res32(gp_handle h, const GemmProblem* d_probs, int batch, int M, int N, const GemmFlags& f) {
  ResFlags rf;
  rf.tilesM = M / RS_BT; rf.tm0 = f.tile_m0; rf.tilesN = N / RS_BT; rf.epi = f.epilogue; rf.alpha = (float)f.alpha;
  if (f.tile_m0 > 0 || f.tile_mcount > 0) {
    const int all = rf.tilesM;
    if (f.tile_m0 >= all) return GP_OK;
    rf.tilesM = (f.tile_mcount > 0 && f.tile_m0 + f.tile_mcount < all) ? f.tile_mcount : all - f.tile_m0;
  }
  // column tiles per workgroup: long runs amortise the resident operand's load (256 KiB of float64 from L2), but the
  // launch wants a few workgroups per CU for balance
  int ctl = 16;
  while (ctl > 2 && (int64_t)batch * rf.tilesM * ((rf.tilesN + ctl - 1) / ctl) < 3 * 256) ctl >>= 1;
  rf.chunk_tiles = ctl;
  rf.chunks = (rf.tilesN + ctl - 1) / ctl;
  static std::atomic<uint32_t> attr_devs{0};
  const uint32_t bit = 1u << (h->device & 31);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)gemm_res_f32_kernel<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS_BYTES));
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_res_f32_kernel<TAG>), dim3(rf.tilesM * rf.chunks, 1, batch), dim3(RS_THREADS), RS_BYTES, h->stream, d_probs, rf);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// true = taken (status in *st).  Whole aligned strips, M a multiple of 128 and at most 256, every problem M = maxM.
bool launch_gemm_res_f32(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f, gp_status* st) {
  // OFF by default (GP_RES32=1 selects it).  MEASURED on cfg3 (N = 32768, M = 256, P = 12, same box): alone on the device it
  // ties gemm_f32.hip's kernels — Kuf_bar 1.00 vs 0.98 ms per step (0.66 of the float32 matrix peak both), A = W Kuf 0.97 vs
  // 0.96, Lq^T A 0.86 vs 0.75 — because at this size the products are within 1.6x of their HBM time (6.4 GB of strips per
  // step), not bound by tile prologues as their in-step timers (inflated by the kernels running beside them) suggested;
  // and in the overlapped step it LOSES (5.1 vs 4.8 ms): a workgroup that owns a CU's whole LDS keeps the helper stream's
  // kernels off that CU.  Ablations (-DRS_EXP=1/2/3): without the stream's loads 0.89, without the barrier 0.90, without
  // the MFMAs 0.46 ms.  Kept as a measured alternative; parity: tests/test_gpu_f32.py (GP_RES32=1).
  static const bool enabled = getenv("GP_RES32") && atoi(getenv("GP_RES32")) != 0;
  if (!enabled || !f.uniform_aligned || f.role < 1 || f.role > 3) return false;
  if (maxM > RS_MAXK || (maxM % RS_BT) != 0 || (maxN % RS_BT) != 0 || f.beta != 0.0 || f.triC != TRI_NONE) return false;
  if (f.role == 3 ? (f.scale_mode != 1) : (f.alpha != 1.0 || f.scale_mode != 0)) return false;
  if (f.role == 1) *st = launch_res32<1>(h, d_probs, batch, maxM, maxN, f);
  else if (f.role == 2) *st = launch_res32<2>(h, d_probs, batch, maxM, maxN, f);
  else *st = launch_res32<3>(h, d_probs, batch, maxM, maxN, f);
  return true;
}
