// gemm_res_f32.hip — float32 strip products for SMALL inducing sets (M <= 256) with the M x M operand RESIDENT in LDS
// (gfx950, round 3; opt-in, GP_RES32=1: see the measurements at launch_gemm_res_f32).
//
// A workgroup (8 wavefronts, ONE per CU: the LDS is the resource) takes a 128-row block of op(A) — W = Lm^-1
// (gpitch/pdgp.py:147 conditional: matrix_triangular_solve as a product), Lq^T, or R = W^T (Lq Lq^T - I) of the backward
// pass — converts it to float32 ONCE into LDS (128 rows x K <= 256: 128 KiB, already in MFMA fragment order) and then
// streams the float32 strip B for a whole run of column tiles.  The strip never touches LDS: wavefront (wr, wc) owns rows
// [64 wr, +64) x columns [32 wc, +32) of the 128-column tile and loads just its own B fragments straight into the MFMA
// operand layout (eight dword loads per lane and 16-row K-tile, four K-tiles in flight in four register buffers that keep
// their identity), so after the resident block is in place NOTHING is shared between the wavefronts and the K loop has no
// barrier; the stream does not stop at a column-tile boundary either.  Column reductions (sum A^2, A^T q_mu, sum LTA^2:
// float64, row-block partials as in the other strip kernels) combine the two row halves through LDS in a fixed order,
// one barrier per column tile.
// LDS fragment order of the resident block: [k-step][row-tile half][lane][row tile & 3]: one ds_read_b128 fetches a
// wavefront's four row tiles' operands of a k-step.
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
#define RS_EXP 0      // measurement variants (tools/build_variant.sh): 1 no stream loads, 3 no MFMAs
#endif
#define RS_BT 128
#define RS_BK 16
#define RS_THREADS 512
#define RS_MAXK 256
#define RS_A_FLOATS (RS_MAXK * RS_BT)                 // 32768
#define RS_BYTES ((size_t)RS_A_FLOATS * sizeof(float) + (RS_BT + 2 * 4 * 32 * 2) * sizeof(double))

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
  double* v0s = reinterpret_cast<double*>(smem + RS_A_FLOATS);
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

  // ---- B stream, per WAVEFRONT and through registers only: wavefront w needs just its own 16 columns of a K-tile, in the
  //      MFMA operand layout (lane = 16 (k & 3) + column): four dword loads per lane and K-tile, RS_NB K-tiles in flight.
  //      Nothing is shared between the wavefronts after the resident operand is in place, so the K loop has NO barrier:
  //      the wavefronts drift apart and the matrix pipe of a SIMD always has one of its two wavefronts to run.
  const int wr = w >> 2, wc = w & 3;            // wavefront tile: rows [64 wr, 64 wr + 64) x columns [32 wc, 32 wc + 32)
  const uint32_t voffB = (uint32_t)(((int64_t)kq * p.ldb + lc) * 4);
  const int64_t rowB4 = (int64_t)4 * p.ldb * 4;                                          // four k rows further
  const int64_t stepB = (int64_t)RS_BK * p.ldb * 4;
  const int64_t wrapB = (int64_t)RS_BT * 4 - (int64_t)nk * stepB;                       // back to the first K-tile, next column tile
  gcbytes sB = rs_uniform((gcbytes)p.B + ((int64_t)kbeg * p.ldb + (int64_t)ct0 * RS_BT + 32 * wc) * 4);
  int gl = 0, gl_kt = 0;                       // next K-tile of the stream to request, its index inside the column tile
  // (an unconditional load — past the end of the stream the last K-tile is simply read again: a load inside a branch
  // makes every later wait a wait for ALL outstanding loads)
  typedef const float __attribute__((address_space(1))) * gcfptr;
  struct BBuf { f4 c0, c1; };                   // B fragments of the wavefront's two column tiles, k-steps 0..3
  auto request = [&](BBuf& rb) {
    uint32_t vo = voffB;
    asm volatile("" : "+v"(vo));
#if RS_EXP == 1
    gl++; return;      // measurement variant: no stream loads (results are garbage)
#endif
    const gcbytes s1 = rs_uniform(sB + rowB4), s2 = rs_uniform(sB + 2 * rowB4), s3 = rs_uniform(sB + 3 * rowB4);
    rb.c0.x = *(gcfptr)(sB + vo); rb.c1.x = *(gcfptr)(sB + vo + 64);       // (the two halves of a 128-byte line, back to back)
    rb.c0.y = *(gcfptr)(s1 + vo); rb.c1.y = *(gcfptr)(s1 + vo + 64);
    rb.c0.z = *(gcfptr)(s2 + vo); rb.c1.z = *(gcfptr)(s2 + vo + 64);
    rb.c0.w = *(gcfptr)(s3 + vo); rb.c1.w = *(gcfptr)(s3 + vo + 64);
    gl++;
    if (gl < G) {
      sB = (gcbytes)((int64_t)sB + stepB);
      gl_kt++;
      if (gl_kt == nk) { gl_kt = 0; sB = (gcbytes)((int64_t)sB + wrapB); }
    }
  };
  // role 3: the sixteen column scales of this wavefront's columns come through the scalar cache (constant address space:
  // 2 gv is not written while this kernel runs), one column tile ahead, and are dealt to the lanes with selects
  typedef const double __attribute__((address_space(4))) * ccptr;
  auto column_scale = [&](int ctile, int b) -> float {
    float sc = 1.f;
    if (TAG == 3) {
      const ccptr cv = (ccptr)p.v1 + ((int64_t)min(ctile, f.tilesN - 1) * RS_BT + 32 * wc + 16 * b);
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
  BBuf rb0 = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, rb1 = rb0, rb2 = rb0, rb3 = rb0;
  request(rb0); request(rb1); request(rb2); request(rb3);
  float sc0 = column_scale(ct0, 0), sc1 = column_scale(ct0, 1);
  f4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; a++) { acc[a][0] = f4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = acc[a][0]; }
  const float* Al = Ares + lane * 4 + wr * 256;               // + ks * 512 floats: this wavefront's four row tiles
  double* comb = v0s + RS_BT;                                 // [2][4 wc][32 columns][2]: the lower row half's column sums
  int kt = 0, ct = 0;
  __syncthreads();                                            // the resident operand is complete
  // one K-tile; `rb` holds its B fragments (requested four K-tiles ago) and receives K-tile t + 4
  auto body = [&](BBuf& rb) {
    f4 af[4];
    {
      const float* Ak = Al + kt * (4 * 2 * 256);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) af[ks] = *reinterpret_cast<const f4*>(Ak + ks * 512);
    }
    // structural zeros of the triangular operands: 16-row tiles outside [alo, ahi] are zero for this whole K-tile
    int alo = 0, ahi = 7;
    {
      const int krel = kbeg + kt * RS_BK - i0;
      if (TRI == TRI_LOWER && krel >= 0) alo = krel >> 4;
      if (TRI == TRI_UPPER) ahi = min(7, (krel + 3) >> 4);
    }
    alo -= 4 * wr; ahi -= 4 * wr;                             // in this wavefront's row tiles 0..3
    const float b0v[4] = {rb.c0.x, rb.c0.y, rb.c0.z, rb.c0.w}, b1v[4] = {rb.c1.x, rb.c1.y, rb.c1.z, rb.c1.w};
    if (RS_EXP == 3) { acc[0][0][0] += af[0][0] * b0v[0] + af[3][3] * b1v[3]; }
    else if (TRI == TRI_NONE || (alo <= 0 && ahi >= 3)) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][a], b0v[ks], acc[a][0], 0, 0, 0);
          acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][a], b1v[ks], acc[a][1], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int a = 0; a < 4; a++)
        if (a >= alo && a <= ahi) {
#pragma unroll
          for (int ks = 0; ks < 4; ks++) {
            acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][a], b0v[ks], acc[a][0], 0, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][a], b1v[ks], acc[a][1], 0, 0, 0);
          }
        }
    }
    request(rb);
    kt++;
    if (kt == nk) {
      // ---- epilogue of column tile ct0 + ct: element r of accumulator (a, b) = row 64 wr + 16 a + 4 kq + r,
      //      column 32 wc + 16 b + lc ---------------------------------------------------------------------------------------
      kt = 0;
      const int j0 = (ct0 + ct) * RS_BT;
      const float scn0 = column_scale(ct0 + ct + 1, 0), scn1 = column_scale(ct0 + ct + 1, 1);   // (consumed after the stores)
      if (f.epi & EPI_STORE) {
        uint32_t vo = (uint32_t)(((int64_t)(4 * kq) * p.ldc + lc) * 4);
        asm volatile("" : "+v"(vo));
        const gcbytes cb0 = (gcbytes)p.C + ((int64_t)(i0 + 64 * wr) * p.ldc + j0 + 32 * wc) * 4;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const gbytes cb = (gbytes)rs_uniform(cb0 + (int64_t)(16 * a + r) * p.ldc * 4);
            *(gfptr)(cb + vo) = (TAG == 3) ? acc[a][0][r] * sc0 : acc[a][0][r];
            *(gfptr)(cb + vo + 64) = (TAG == 3) ? acc[a][1][r] * sc1 : acc[a][1][r];
          }
      }
      if (f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) {
        const double* vv = v0s + 64 * wr + 4 * kq;
        double s2[2] = {0.0, 0.0}, sd[2] = {0.0, 0.0};
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const float scb = b ? sc1 : sc0;
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const double v = (double)((TAG == 3) ? acc[a][b][r] * scb : acc[a][b][r]);
              s2[b] = fma(v, v, s2[b]);
            }
          if (f.epi & EPI_COLDOT) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
              for (int r = 0; r < 4; r++)
                sd[b] = fma((double)((TAG == 3) ? acc[a][b][r] * scb : acc[a][b][r]), vv[a * 16 + r], sd[b]);
          }
          s2[b] += __shfl_xor(s2[b], 16, 64); s2[b] += __shfl_xor(s2[b], 32, 64);
          sd[b] += __shfl_xor(sd[b], 16, 64); sd[b] += __shfl_xor(sd[b], 32, 64);
        }
        // the two row halves of a column are summed in a fixed order (upper half + lower half) through LDS: one barrier
        // per column tile, two alternating buffers
        double* cbuf = comb + (ct & 1) * (4 * 32 * 2) + wc * (32 * 2);
        if (wr == 1 && kq == 0) {
#pragma unroll
          for (int b = 0; b < 2; b++) { cbuf[(16 * b + lc) * 2] = s2[b]; cbuf[(16 * b + lc) * 2 + 1] = sd[b]; }
        }
        __syncthreads();
        if (wr == 0 && kq == 0) {
#pragma unroll
          for (int b = 0; b < 2; b++) {
            const int j = j0 + 32 * wc + 16 * b + lc;
            if (f.epi & EPI_COLSUMSQ) ((gptr)p.o0)[(int64_t)tm * p.N + j] = s2[b] + cbuf[(16 * b + lc) * 2];
            if (f.epi & EPI_COLDOT) ((gptr)p.o1)[(int64_t)tm * p.N + j] = sd[b] + cbuf[(16 * b + lc) * 2 + 1];
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 4; a++) { acc[a][0] = f4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = acc[a][0]; }
      sc0 = scn0; sc1 = scn1;
      ct++;
    }
  };
  // unrolled over the four register buffers (each keeps its identity: no copies); a single-exit main loop, then the
  // remainder as straight-line code
  int t = 0;
  for (; t + 4 <= G; t += 4) { body(rb0); body(rb1); body(rb2); body(rb3); }
  if (t < G) body(rb0);
  if (t + 1 < G) body(rb1);
  if (t + 2 < G) body(rb2);
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
  // OFF by default (GP_RES32=1 selects it).  MEASURED on cfg3 (N = 32768, M = 256, P = 12, same box), ms per step, this
  // kernel / gemm_f32.hip's tiled kernels: ALONE on the device A = W Kuf 0.79 / 1.00, Lq^T A 0.78 / 0.75, Kuf_bar 0.90 /
  // 0.98 (0.73 / 0.67 of the float32 matrix peak), the whole step with everything serial 5.18 / 5.44; in the OVERLAPPED step
  // 4.74 / 4.74 — a workgroup that owns a CU's whole LDS keeps the helper stream's kernels (split-K product, M x M chain)
  // off that CU, and what it gains alone it loses there.  First version (strip through a 3-slot LDS ring with a barrier
  // per K-tile): 0.97 / 0.86 / 1.00 alone, 5.1 ms overlapped.  Ablations of that version (-DRS_EXP=1/2/3): without the
  // stream's loads 0.89, without the barrier 0.90, without the MFMAs 0.46 ms.  Parity: tests/test_gpu_f32.py in a
  // GP_RES32=1 child process.
  static const bool enabled = getenv("GP_RES32") && atoi(getenv("GP_RES32")) != 0;
  if (!enabled || !f.uniform_aligned || f.role < 1 || f.role > 3) return false;
  if (maxM > RS_MAXK || (maxM % RS_BT) != 0 || (maxN % RS_BT) != 0 || f.beta != 0.0 || f.triC != TRI_NONE) return false;
  if (f.role == 3 ? (f.scale_mode != 1) : (f.alpha != 1.0 || f.scale_mode != 0)) return false;
  if (f.role == 1) *st = launch_res32<1>(h, d_probs, batch, maxM, maxN, f);
  else if (f.role == 2) *st = launch_res32<2>(h, d_probs, batch, maxM, maxN, f);
  else *st = launch_res32<3>(h, d_probs, batch, maxM, maxN, f);
  return true;
}
