// gemm_f32.hip — float32 matrix-core forms of the four frame-strip products (gfx950), for the configurations
// BASELINE.json quotes at fp32 (12-pitch transcription model, sgpr_ss source separation).
//
// The reference's dtype is a setting (gpitch/pdgp.py:13 `float_type = settings.dtypes.float_type`,
// gpitch/matern12_spectral_mixture.py:8-11).  What runs in float32 here is exactly the O(M^2 N) part —
//   role 1  A   = W Kuf          tf.matrix_triangular_solve(Lm, Kmn)    (GPflow conditional, pdgp.py:147-155)
//   role 2  LTA = Lq^T A         tf.matmul(Lq^T, A) -> column sums of squares only
//   role 3  G   = R (A D)        backward: Kuf_bar
//   role 4  H   = A D A^T        backward / sgpr_ss.py:49 tf.matmul(A, A, transpose_b=True), split-K over the frames
// — with the M x N strips (Kuf, A, G) STORED as float32 and v_mfma_f32_16x16x4_f32 (157 TFLOP/s dense, exact f32
// fma chains) doing the arithmetic.  Everything that decides conditioning stays float64: Kuu, its Cholesky factor,
// W = L^-1, Lq, R are float64 in memory (converted to float32 as they are staged into LDS), and every reduction
// over the inducing index (sum A^2, A^T q_mu, sum LTA^2) and over the frames (the split-K slabs, A gm) is
// accumulated in float64 (SURVEY section 7 "Conditioning": fvar = Kdiag - sum A^2 + sum LTA^2 is a cancellation).
//
// Tiling follows gemm.hip (128 x 128 output tile, 4 wavefronts 1 x 4, double-buffered LDS, one barrier per K-tile,
// loads for the next K-tile in flight under the MFMAs, structurally-zero MFMA tiles of the triangular operands
// skipped, XCD-contiguous tile ranges) with a K-tile of 32 so that one K-tile is the same number of matrix-core
// cycles (128 MFMAs x 32 cycles per wavefront) and the same bytes as the float64 kernel's.
// LDS layouts: k-contiguous tiles [row][k] with stride 34 floats (the 32 lanes a ds_read_b32 serves hit
// bank 2*row + k: all distinct), row-contiguous tiles [k][col] with stride 144 (bank 16*k + col).
#include "common.h"
#include <type_traits>
#include <atomic>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef const double __attribute__((address_space(1))) * gcptr;
typedef double __attribute__((address_space(1))) * gptr;
typedef const dbl2 __attribute__((address_space(1))) * gcptr2;
typedef const float __attribute__((address_space(1))) * gcfptr;
typedef float __attribute__((address_space(1))) * gfptr;
typedef const f4 __attribute__((address_space(1))) * gcfptr4;

#define F32_BK 32
#define F32_BT 128        // tile edge (rows and columns)
#define F32_THREADS 256
#define F32_SKC 34        // LDS stride of a k-contiguous tile  [128][32]
#define F32_SRC 144       // LDS stride of a row-contiguous tile [32][128]
#define F32_KC_ELEMS (F32_BT * F32_SKC)
#define F32_RC_ELEMS (F32_BK * F32_SRC)

struct Gemm32Flags {
  double alpha;
  int epi;          // EPI_STORE | EPI_COLSUMSQ | EPI_COLDOT (roles 1, 2, 3)
  int scale;        // role 3: B(k, n) *= v1[n]; role 4: B(k, n) *= v1[k] (when v1 != null)
  int sym;          // role 4: lower tiles only
  int ksplit;       // role 4: K-slices
  int tilesM, tilesN;
  int tm0, tilesM_req;   // row-block range of this launch (see GemmFlags::tile_m0)
  const double* xcols;   // KT >= 0 (role 3 with the stationary contraction as epilogue): the frames x
};

// TAG: 1 cond_A (A = W lower, float64 in memory, k-contiguous), 2 cond_LTA (A = Lq^T, upper, float64, row-contiguous),
//      3 kuf_bar (A = R dense, float64, k-contiguous), 4 nt (A, B float32 strips, both k-contiguous, split-K)
// W8: 8 wavefronts per workgroup (2 x 4: each 64 rows x 32 columns, <= 128 VGPRs, four wavefronts per SIMD) instead of
// 4 (1 x 4: each 128 x 32).  A float32 MFMA K-tile is half as long as a float64 one against the same staging work, so
// with two wavefronts per SIMD the matrix pipe idles whenever both are staging (0.68 busy on the dense product).
// KT >= 0 (TAG 3 only): a stationary kernel type — the tile of Kuf_bar is contracted with dK/d(variance, lengthscale) in the
// epilogue and not stored (gemm_strip.hip's role 5 for float32 strips: the weight is the value the strip would have held,
// float32(alpha acc), the arithmetic on it float64 as in hyper_contract_kernel).
template <int TAG, bool W8, int KT = -1>
__global__ void __launch_bounds__(W8 ? 512 : 256, W8 ? 4 : 2) gemm_f32_kernel(const GemmProblem* __restrict__ probs, Gemm32Flags f) {
  constexpr int NTHREADS = W8 ? 512 : 256;
  constexpr int WAVES_M = W8 ? 2 : 1;
  constexpr bool A_RC = (TAG == 2);          // A tile row-contiguous ([k][i]) instead of k-contiguous ([i][k])
  constexpr bool B_KC = (TAG == 4);          // B tile k-contiguous ([n][k]) instead of row-contiguous ([k][n])
  constexpr bool A_F64 = (TAG != 4);
  constexpr int TM = 8 / WAVES_M, TN = 2;    // 16 x 16 MFMA tiles per wave: 128 (or 64) rows x 32 columns
  constexpr int EPT = 4096 / NTHREADS;       // staged elements per thread and tile (16 or 8)
  constexpr int A_ELEMS = A_RC ? F32_RC_ELEMS : F32_KC_ELEMS;
  constexpr int B_ELEMS = B_KC ? F32_KC_ELEMS : F32_RC_ELEMS;
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  extern __shared__ __attribute__((aligned(16))) float smem32[];

  // XCD-contiguous ranges of logical (tile, batch) indices over the flattened grid (see gemm.hip)
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
  const gcptr gA64 = (gcptr)p.A; const gcfptr gA32 = (gcfptr)(const void*)p.A;
  const gcfptr gB = (gcfptr)(const void*)p.B;
  const gfptr gC = (gfptr)(void*)p.C;
  const gcptr gv0 = (gcptr)p.v0, gv1 = (gcptr)p.v1, gv2 = (gcptr)p.v2;
  const gptr go0 = (gptr)p.o0, go1 = (gptr)p.o1, go2 = (gptr)p.o2;

  int tm, tn, ksl = 0;
  if (TAG == 4) {
    const int ntl = f.sym ? f.tilesM * (f.tilesM + 1) / 2 : f.tilesM * f.tilesN;
    ksl = bid / ntl;
    const int t = bid % ntl;
    if (f.sym) {
      tm = (int)((__dsqrt_rn(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
      while (tm * (tm + 1) / 2 > t) tm--;
      tn = t - tm * (tm + 1) / 2;
    } else { tm = t % f.tilesM; tn = t / f.tilesM; }
  } else {
    tn = bid / f.tilesM;
    tm = f.tm0 + (bid % f.tilesM + tn) % f.tilesM;      // rotate the row-block by the strip index (see gemm.hip)
  }
  const int i0 = tm * F32_BT, j0 = tn * F32_BT;
  if (i0 >= p.M || j0 >= p.N) return;

  int kbeg = 0, kend = p.K;
  if (TAG == 1) kend = min(kend, i0 + F32_BT);
  if (TAG == 2) kbeg = (i0 / F32_BK) * F32_BK;
  if (TAG == 4) {
    const int nk = (kend - kbeg + F32_BK - 1) / F32_BK;
    const int per = (nk + f.ksplit - 1) / f.ksplit;
    const int b0 = kbeg + ksl * per * F32_BK;
    kend = min(kend, b0 + per * F32_BK);
    kbeg = b0;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave & 3, wr = wave >> 2;       // wave column (32 columns each) and wave row
  const int wrow0 = wr * (16 * TM);              // first row of this wave inside the tile
  const int lc = lane & 15, kq = lane >> 4;
  // column tiles of this wave; the symmetric product pairs tile wc with 7 - wc so that on a diagonal output tile every
  // wave skips the same number of upper MFMA tiles
  constexpr bool PERM = (TAG == 4);
  int ctile[TN];
#pragma unroll
  for (int b = 0; b < TN; b++) ctile[b] = PERM ? (b == 0 ? wc : 7 - wc) : (wc * TN + b);
  const bool diag_sym = PERM && f.sym && (tm == tn);
  int cmin[TN];
#pragma unroll
  for (int b = 0; b < TN; b++)
    cmin[b] = diag_sym ? max(0, __builtin_amdgcn_readfirstlane(ctile[b]) - __builtin_amdgcn_readfirstlane(wr) * TM) : 0;

  // ---- staging maps -----------------------------------------------------------------------------------------
  // k-contiguous tile: thread t holds 16 consecutive k of row t / 2; row-contiguous tile: thread t holds 4 consecutive
  // columns (t & 31) * 4 of the four k-rows (t >> 5) + 8 e
  // k-contiguous tile: EPT consecutive k of one row per thread; row-contiguous tile: 4 consecutive columns of EPT / 4 k-rows
  constexpr int KC_TPR = 32 / EPT;               // threads per row of a k-contiguous tile
  constexpr int RC_E = EPT / 4, RC_STRIDE = NTHREADS / 32;
  const int kc_r = tid / KC_TPR, kc_k = (tid % KC_TPR) * EPT;
  const int rc_k = tid >> 5, rc_c = (tid & 31) * 4;
  double ra64[A_F64 ? EPT : 1];
  float ra32[A_F64 ? 1 : EPT];
  float rb[EPT];
  float rs[EPT];      // per-element scale of B (role 3: by column, fetched once; role 4: by k)
  const bool rowdot = (TAG == 4) && (p.v2 != nullptr) && (tn == 0);
  double udot = 0.0, rg[(TAG == 4) ? EPT : 1];
  const bool scale = (TAG == 3) || (TAG == 4 && f.scale && p.v1 != nullptr);

  if (TAG == 3) {
#pragma unroll
    for (int e = 0; e < RC_E; e++)
#pragma unroll
      for (int c = 0; c < 4; c++) { const int n = j0 + rc_c + c; rs[e * 4 + c] = (n < p.N) ? (float)gv1[n] : 0.f; }
  }

  const bool a_vec = A_F64 ? (((p.lda & 1) == 0) && ((((uintptr_t)p.A) & 15) == 0))
                           : (((p.lda & 3) == 0) && ((((uintptr_t)p.A) & 15) == 0));
  const bool b_vec = ((p.ldb & 3) == 0) && ((((uintptr_t)p.B) & 15) == 0);

  auto load_tiles = [&](int kt, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    // ---- A ----
    if constexpr (A_RC) {           // TAG 2: A_op[i][k] = Lq[k][i]; memory rows are k
#pragma unroll
      for (int e = 0; e < RC_E; e++) {
        const int k = kt + rc_k + RC_STRIDE * e, i = i0 + rc_c;
        const gcptr src = gA64 + (int64_t)k * p.lda + i;
        if (FAST) {
          const dbl2 u = *(gcptr2)(src), v = *(gcptr2)(src + 2);
          ra64[e * 4 + 0] = u.x; ra64[e * 4 + 1] = u.y; ra64[e * 4 + 2] = v.x; ra64[e * 4 + 3] = v.y;
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++) ra64[e * 4 + c] = (k < kend && i + c < p.M) ? src[c] : 0.0;
        }
      }
    } else if constexpr (A_F64) {   // TAG 1, 3: A[i][k], k contiguous in memory
      const int i = i0 + kc_r, k = kt + kc_k;
      const gcptr src = gA64 + (int64_t)i * p.lda + k;
      if (FAST) {
#pragma unroll
        for (int e = 0; e < EPT; e += 2) { const dbl2 v = *(gcptr2)(src + e); ra64[e] = v.x; ra64[e + 1] = v.y; }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; e++) ra64[e] = (i < p.M && k + e < kend) ? src[e] : 0.0;
      }
    } else {              // TAG 4: X[i][k] float32 strip
      const int i = i0 + kc_r, k = kt + kc_k;
      const gcfptr src = gA32 + (int64_t)i * p.lda + k;
      if (FAST) {
#pragma unroll
        for (int e = 0; e < EPT; e += 4) { const f4 v = *(gcfptr4)(src + e); ra32[e] = v.x; ra32[e + 1] = v.y; ra32[e + 2] = v.z; ra32[e + 3] = v.w; }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; e++) ra32[e] = (i < p.M && k + e < kend) ? src[e] : 0.f;
      }
      if constexpr (TAG == 4) {
        if (rowdot) {
#pragma unroll
          for (int e = 0; e < EPT; e++) rg[e] = (FAST || k + e < kend) ? gv2[k + e] : 0.0;
        }
      }
    }
    // ---- B ----
    if constexpr (B_KC) {           // TAG 4: B_op[k][n] = Y[n][k]
      const int n = j0 + kc_r, k = kt + kc_k;
      const gcfptr src = gB + (int64_t)n * p.ldb + k;
      if (FAST) {
#pragma unroll
        for (int e = 0; e < EPT; e += 4) { const f4 v = *(gcfptr4)(src + e); rb[e] = v.x; rb[e + 1] = v.y; rb[e + 2] = v.z; rb[e + 3] = v.w; }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; e++) rb[e] = (n < p.N && k + e < kend) ? src[e] : 0.f;
      }
      if (scale) {
#pragma unroll
        for (int e = 0; e < EPT; e++) rs[e] = (FAST || k + e < kend) ? (float)gv1[k + e] : 0.f;
      }
    } else {              // TAG 1, 2, 3: B[k][n], n contiguous
#pragma unroll
      for (int e = 0; e < RC_E; e++) {
        const int k = kt + rc_k + RC_STRIDE * e, n = j0 + rc_c;
        const gcfptr src = gB + (int64_t)k * p.ldb + n;
        if (FAST) {
          const f4 v = *(gcfptr4)(src);
          rb[e * 4] = v.x; rb[e * 4 + 1] = v.y; rb[e * 4 + 2] = v.z; rb[e * 4 + 3] = v.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++) rb[e * 4 + c] = (k < kend && n + c < p.N) ? src[c] : 0.f;
        }
      }
    }
  };

  // structural masks, conversion, scaling, then registers -> LDS (runs after the MFMAs of the current K-tile)
  auto store_tiles = [&](int buf, int kt) {
    float* As = smem32 + buf * STAGE;
    float* Bs = As + A_ELEMS;
    if constexpr (A_RC) {
#pragma unroll
      for (int e = 0; e < RC_E; e++) {
        const int k = kt + rc_k + RC_STRIDE * e;
        f4 v;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int i = i0 + rc_c + c;
          v[c] = (k < i) ? 0.f : (float)ra64[e * 4 + c];          // Lq^T: non-zero iff k >= i
        }
        *reinterpret_cast<f4*>(As + (rc_k + RC_STRIDE * e) * F32_SRC + rc_c) = v;
      }
    } else {
      const int i = i0 + kc_r;
#pragma unroll
      for (int e = 0; e < EPT; e += 2) {
        float v0, v1;
        if constexpr (A_F64) { v0 = (float)ra64[e]; v1 = (float)ra64[e + 1]; } else { v0 = ra32[e]; v1 = ra32[e + 1]; }
        if (TAG == 1) {                                            // W lower: non-zero iff k <= i
          if (kt + kc_k + e > i) v0 = 0.f;
          if (kt + kc_k + e + 1 > i) v1 = 0.f;
        }
        if constexpr (TAG == 4) { if (rowdot) { udot = fma((double)v0, rg[e], udot); udot = fma((double)v1, rg[e + 1], udot); } }
        *reinterpret_cast<float2*>(As + kc_r * F32_SKC + kc_k + e) = make_float2(v0, v1);
      }
    }
    if constexpr (B_KC) {
#pragma unroll
      for (int e = 0; e < EPT; e += 2) {
        float v0 = rb[e], v1 = rb[e + 1];
        if (scale) { v0 *= rs[e]; v1 *= rs[e + 1]; }
        *reinterpret_cast<float2*>(Bs + kc_r * F32_SKC + kc_k + e) = make_float2(v0, v1);
      }
    } else {
#pragma unroll
      for (int e = 0; e < RC_E; e++) {
        f4 v;
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = (TAG == 3) ? rb[e * 4 + c] * rs[e * 4 + c] : rb[e * 4 + c];
        *reinterpret_cast<f4*>(Bs + (rc_k + RC_STRIDE * e) * F32_SRC + rc_c) = v;
      }
    }
  };

  f4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

  constexpr bool KDOWN = (TAG == 2);      // the row-blocks of a strip walk down from their common end (gemm.hip)
  auto run_k_loop = [&](auto fast_tag) {
    const int nkt = (kend - kbeg + F32_BK - 1) / F32_BK;
    const int kfirst = KDOWN ? kbeg + (nkt - 1) * F32_BK : kbeg;
    const int kstep = KDOWN ? -F32_BK : F32_BK;
    load_tiles(kfirst, fast_tag);
    store_tiles(0, kfirst);
    __syncthreads();
    int buf = 0;
    for (int it = 0, kt = kfirst; it < nkt; it++, kt += kstep) {
      const bool more = (it + 1 < nkt);
      if (more) load_tiles(kt + kstep, fast_tag);
      const float* As = smem32 + buf * STAGE;
      const float* Bs = As + A_ELEMS;
      bool full = true;
      if (TAG == 1) full = (kt + F32_BK - 1 <= i0 + wrow0 + 15);
      if (TAG == 2) full = (kt >= i0 + wrow0 + 16 * TM - 16);
      __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int ks = 0; ks < F32_BK / 4; ks++) {
        const int k = ks * 4 + kq;
        int a_lo = 0, a_hi = TM;
        if (!full) {
          const int kg = kt + ks * 4;
          if (TAG == 1) a_lo = max(0, (kg - i0 - wrow0) >> 4);                 // need rowbase + 16 a + 15 >= kg
          if (TAG == 2) a_hi = min(TM, ((kg + 3 - i0 - wrow0) >> 4) + 1);      // need rowbase + 16 a <= kg + 3
        }
        float af[TM], bf[TN];
#pragma unroll
        for (int b = 0; b < TN; b++) {
          const int n = ctile[b] * 16 + lc;
          bf[b] = B_KC ? Bs[n * F32_SKC + k] : Bs[k * F32_SRC + n];
        }
#pragma unroll
        for (int a = 0; a < TM; a++) {
          const int i = wrow0 + a * 16 + lc;
          af[a] = A_RC ? As[k * F32_SRC + i] : As[i * F32_SKC + k];
        }
#pragma unroll
        for (int a = 0; a < TM; a++) {
          if ((TAG == 1 || TAG == 2) && !(a >= a_lo && a < a_hi)) continue;
#pragma unroll
          for (int b = 0; b < TN; b++)
            if (!PERM || a >= cmin[b])
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if (more) store_tiles(buf ^ 1, kt + kstep);
      __syncthreads();
      buf ^= 1;
    }
  };
  if (kbeg < kend) {
    const bool fastpath = a_vec && b_vec && (i0 + F32_BT <= p.M) && (j0 + F32_BT <= p.N) && (((kend - kbeg) % F32_BK) == 0);
    if (fastpath) run_k_loop(std::true_type{}); else run_k_loop(std::false_type{});
  }

  // ---- epilogue: accumulator element r of tile (a, b) is row i0 + 16 a + 4 kq + r, column j0 + 16 ctile[b] + lc ----
  if (TAG == 4) {
    gptr slab = go2 + (int64_t)ksl * p.M * p.N;
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = i0 + wrow0 + a * 16 + kq * 4 + r, j = j0 + ctile[b] * 16 + lc;
          if (i < p.M && j < p.N) slab[(int64_t)i * p.N + j] = (double)acc[a][b][r];
        }
    if (rowdot) {
#pragma unroll
      for (int o = 1; o < KC_TPR; o <<= 1) udot += __shfl_xor(udot, o, 64);   // the threads of a row hold disjoint k-ranges
      if ((tid % KC_TPR) == 0 && i0 + kc_r < p.M) go1[(int64_t)ksl * p.M + i0 + kc_r] = udot;
    }
    return;
  }
  const float alpha = (float)f.alpha;
  if (TAG == 3 && KT >= 0) {
    __syncthreads();                                // every wavefront is past its last LDS fragment read
    double* sm = reinterpret_cast<double*>(smem32);
    double* etab = sm;                 // GP_EXP_TAB = 64
    double* row_a = sm + 64;           // 128: z_i / l
    double* row_al = sm + 192;         // 128: alpha_i
    double* red = sm + 320;            // [wavefronts][2]
    const gcptr th = (gcptr)p.kern.theta;
    const double var = th[0], ls = th[1];
    gp_exp_tab_init(etab);
    if (tid < F32_BT) { row_a[tid] = ((gcptr)p.xa)[i0 + tid] / ls; row_al[tid] = gv0[i0 + tid]; }
    __syncthreads();
    const double inv_ls = 1.0 / ls;
    double acc_v = 0.0, acc_l = 0.0;
    const gcptr gx = (gcptr)f.xcols;
#pragma unroll
    for (int b = 0; b < TN; b++) {
      const int j = j0 + ctile[b] * 16 + lc;
      const double bcol = gx[j] / ls, bb = __dmul_rn(bcol, bcol), gmj = gv2[j];
#pragma unroll
      for (int a = 0; a < TM; a++) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int ii = wrow0 + a * 16 + kq * 4 + r;
          const double av = row_a[ii], aa = __dmul_rn(av, av);
          const double w = fma(row_al[ii], gmj, (double)(alpha * acc[a][b][r]));
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
    if (lane == 0) { red[wave * 2 + 0] = acc_v; red[wave * 2 + 1] = acc_l; }
    __syncthreads();
    if (tid < 2) {
      double sacc = 0.0;
#pragma unroll
      for (int wv = 0; wv < NTHREADS / 64; wv++) sacc += red[wv * 2 + tid];
      go0[((int64_t)tm * f.tilesN + tn) * 2 + tid] = sacc;
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
          const int i = i0 + wrow0 + a * 16 + kq * 4 + r, j = j0 + ctile[b] * 16 + lc;
          if (i < p.M && j < p.N) gC[(int64_t)i * p.ldc + j] = alpha * acc[a][b][r];
        }
  }
  if (f.epi & 6) {
    float* red_f = smem32;                          // reused as [2 kinds][2 wave rows][128] doubles when W8
    double* red = reinterpret_cast<double*>(red_f);
    if (W8) __syncthreads();                        // every wavefront is past its last LDS fragment read
#pragma unroll
    for (int b = 0; b < TN; b++) {
      double s2 = 0.0, sd = 0.0;
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = i0 + wrow0 + a * 16 + kq * 4 + r;
          const double v = (double)(alpha * acc[a][b][r]);      // rows >= M hold exact zeros
          s2 = fma(v, v, s2);
          if (f.epi & 4) sd = fma(v, (i < p.M) ? gv0[i] : 0.0, sd);
        }
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      sd += __shfl_xor(sd, 16, 64); sd += __shfl_xor(sd, 32, 64);
      const int n = ctile[b] * 16 + lc;
      if (!W8) {
        const int j = j0 + n;
        if (kq == 0 && j < p.N) {
          if (f.epi & 2) go0[(int64_t)tm * p.N + j] = s2;
          if (f.epi & 4) go1[(int64_t)tm * p.N + j] = sd;
        }
      } else if (kq == 0) {
        red[(0 * 2 + wr) * F32_BT + n] = s2;
        red[(1 * 2 + wr) * F32_BT + n] = sd;
      }
    }
    if (W8) {
      __syncthreads();
      if (tid < F32_BT) {
        const int j = j0 + tid;
        if (j < p.N) {
          if (f.epi & 2) go0[(int64_t)tm * p.N + j] = red[(0 * 2 + 0) * F32_BT + tid] + red[(0 * 2 + 1) * F32_BT + tid];
          if (f.epi & 4) go1[(int64_t)tm * p.N + j] = red[(1 * 2 + 0) * F32_BT + tid] + red[(1 * 2 + 1) * F32_BT + tid];
        }
      }
    }
  }
}

#ifndef GP_F32_W8_MASK
#define GP_F32_W8_MASK 15     // bit (role - 1): that role runs with 8 wavefronts per workgroup (same-box A/B, overlap 0: cond_A 3.02 -> 2.53 ms, Lq^T A 2.34 -> 2.24, A D A^T 2.90 -> 2.56, Kuf_bar 4.03 -> 3.60)
#endif
template <int TAG, int KT = -1>
static gp_status launch_f32(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, Gemm32Flags f) {
  constexpr bool W8 = ((GP_F32_W8_MASK >> (TAG - 1)) & 1) != 0;
  constexpr int A_ELEMS = (TAG == 2) ? F32_RC_ELEMS : F32_KC_ELEMS;
  constexpr int B_ELEMS = (TAG == 4) ? F32_KC_ELEMS : F32_RC_ELEMS;
  constexpr size_t BYTES = (size_t)2 * (A_ELEMS + B_ELEMS) * sizeof(float);
  f.tilesM = (maxM + F32_BT - 1) / F32_BT;
  if (TAG != 4 && (f.tm0 > 0 || f.tilesM_req > 0)) {
    const int all = f.tilesM;
    if (f.tm0 >= all) return GP_OK;
    f.tilesM = (f.tilesM_req > 0 && f.tm0 + f.tilesM_req < all) ? f.tilesM_req : all - f.tm0;
  }
  f.tilesN = (maxN + F32_BT - 1) / F32_BT;
  int ntiles = f.tilesM * f.tilesN;
  if (TAG == 4) { if (f.sym) ntiles = f.tilesM * (f.tilesM + 1) / 2; ntiles *= f.ksplit; }
  static std::atomic<int> attr_dev_mask{0};     // per instantiation; one bit per device (LDS limit is a per-device attribute)
  const int bit = 1 << (h->device & 31);
  if (!(attr_dev_mask.load(std::memory_order_acquire) & bit)) {
    GP_HIP_CHECK(h, hipFuncSetAttribute((const void*)gemm_f32_kernel<TAG, W8, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BYTES));
    attr_dev_mask.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_f32_kernel<TAG, W8, KT>), dim3(ntiles, 1, batch), dim3(W8 ? 512 : 256), BYTES, h->stream, d_probs, f);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

bool gemm_f32_fused_contraction_ok(int maxM, int maxN, int ktype) {
  const bool fuse = gp_switches().hyper_fuse != 0;
  const bool stat = (ktype == GP_KERN_MATERN12 || ktype == GP_KERN_MATERN32 || ktype == GP_KERN_MATERN52 || ktype == GP_KERN_RBF);
  return fuse && stat && maxM > 0 && (maxM % F32_BT) == 0 && (maxN % F32_BT) == 0;
}

// roles 1-3 (see the header of this file); flags as launch_gemm_batched (epilogue bits, alpha); operands: A float64
// M x M (lda in doubles), B / C float32 strips (ldb / ldc in floats)
gp_status launch_gemm_f32_role(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& gf) {
  if (batch <= 0 || maxM <= 0 || maxN <= 0) return GP_OK;
  GpTimerScope ts(h, gf.timer);
  {   // whole aligned strips with float32 scratch for the M x M operand: a 64 x 64 tile per wavefront (gemm_wave_f32.hip)
    gp_status st = GP_OK;
    if (launch_gemm_wave_f32(h, d_probs, batch, maxM, maxN, gf, &st)) return st;
  }
  if (gf.role == 5) {     // Kuf_bar with the stationary family's contraction as its epilogue (the caller asked gemm_f32_fused_contraction_ok)
    Gemm32Flags f;
    f.alpha = gf.alpha; f.epi = 0; f.scale = 1; f.sym = 0; f.ksplit = 1; f.tilesM = f.tilesN = 1; f.tm0 = 0; f.tilesM_req = 0;
    f.xcols = gf.aux_x;
    if (!gf.aux_x || (maxM % F32_BT) || (maxN % F32_BT)) return gp_fail(h, GP_ERR_BAD_ARG, "fused float32 Kuf_bar contraction: bad launch");
    switch (gf.aux_ktype) {
      case GP_KERN_MATERN12: return launch_f32<3, GP_KERN_MATERN12>(h, d_probs, batch, maxM, maxN, f);
      case GP_KERN_MATERN32: return launch_f32<3, GP_KERN_MATERN32>(h, d_probs, batch, maxM, maxN, f);
      case GP_KERN_MATERN52: return launch_f32<3, GP_KERN_MATERN52>(h, d_probs, batch, maxM, maxN, f);
      case GP_KERN_RBF: return launch_f32<3, GP_KERN_RBF>(h, d_probs, batch, maxM, maxN, f);
      default: return gp_fail(h, GP_ERR_BAD_ARG, "fused float32 Kuf_bar contraction: not a stationary kernel");
    }
  }
  {   // whole aligned strips: gemm_strip_f32.hip's lean form
    gp_status st = GP_OK;
    if (launch_gemm_strip_f32_lean(h, d_probs, batch, maxM, maxN, gf, &st)) return st;
  }
  Gemm32Flags f;
  f.alpha = gf.alpha; f.epi = gf.epilogue; f.scale = gf.scale_mode; f.sym = 0; f.ksplit = 1; f.tilesM = f.tilesN = 1;
  f.tm0 = gf.tile_m0; f.tilesM_req = gf.tile_mcount; f.xcols = nullptr;
  if (gf.beta != 0.0) return gp_fail(h, GP_ERR_UNSUPPORTED, "float32 strip product: beta != 0");
  switch (gf.role) {
    case 1: return launch_f32<1>(h, d_probs, batch, maxM, maxN, f);
    case 2: return launch_f32<2>(h, d_probs, batch, maxM, maxN, f);
    case 3: return launch_f32<3>(h, d_probs, batch, maxM, maxN, f);
    default: return gp_fail(h, GP_ERR_UNSUPPORTED, "float32 strip product: unknown role");
  }
}

// H = (X diag(d)) X^T over the frames with float32 strips: float32 MFMA per K-slice, slabs and their fixed-order sum in
// float64 (slab_reduce_kernel of gemm.hip, shared)
gp_status launch_slab_reduce(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int nsplit, int sym, double alpha);
gp_status launch_gemm_f32_nt_reduce_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxNlong,
                                            int nsplit, int sym, int scale_by_k, double alpha, int uniform_aligned) {
  if (batch <= 0 || maxM <= 0) return GP_OK;
  (void)uniform_aligned;    // (a lean form of this product measured slower than the 8-wavefront kernel: DESIGN.md section 3b)
  {
    GpTimerScope ts(h, GP_TIMER_NT_GEMM);
    Gemm32Flags f;
    f.alpha = 1.0; f.epi = 1; f.scale = scale_by_k; f.sym = sym; f.ksplit = nsplit > 1 ? nsplit : 2; f.tilesM = f.tilesN = 1;
    f.tm0 = 0; f.tilesM_req = 0; f.xcols = nullptr;
    GP_CHECK(launch_f32<4>(h, d_probs, batch, maxM, maxM, f));
  }
  return launch_slab_reduce(h, d_probs, batch, maxM, nsplit > 1 ? nsplit : 2, sym, alpha);
}
