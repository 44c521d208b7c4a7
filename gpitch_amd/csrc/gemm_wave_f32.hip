// gemm_wave_f32.hip — float32 form of gemm_wave.hip: the strip products of a latent GP whose M x N strips are float32
// (BASELINE configs 3 and 5 are quoted at fp32; the reference's dtype is a setting, gpitch/pdgp.py:13), a 64 x 64 tile per
// wavefront, MFMA operands straight from buffer loads, no LDS, no barrier (gfx950, round 4).
//
//   role 1  A   = W Kuf      tf.matrix_triangular_solve(Lm, Kmn) + reduce_sum(A^2), A^T q_mu   (GPflow conditional,
//   role 2  LTA = Lq^T A     tf.matmul(Lq^T, A) -> reduce_sum(LTA^2) only                        gpitch/pdgp.py:147-155)
//   role 3  G   = R (A D)    backward: Kuf_bar;   role 5: the same with a stationary family's contraction as its epilogue
//
// What is float32 / what stays float64 is gemm_f32.hip's contract: strips float32, v_mfma_f32_16x16x4_f32 accumulating in
// float32 over the inducing index; every reduction over the rows of a tile (sum A^2, A^T q_mu, sum LTA^2) and the
// contraction epilogue in float64.  Why this form: at cfg3's M = 256 a 128 x 128 LDS tile lives for 4-16 K-tiles between a
// prologue and an epilogue a third of its life long (0.42-0.71 matrix-core busy); a wavefront tile
// has neither a prologue to speak of nor a barrier, and at ~170 VGPRs three of them share a SIMD.
//  * The M x M operand is converted ONCE per launch into a float32 copy in the form the loads want (gwf_convert_kernel, 0.26 MB
//    per latent GP at M = 256): role 1 W as is (exact zeros above the diagonal), role 2 tril(Lq)^T — transposed AND masked, so
//    op(A) is k-contiguous and upper triangular like a mirrored W and needs no mask in the loop — roles 3 / 5 R.
//  * v_mfma_f32_16x16x4_f32: A lane (kq, lc) = op(A)[row lc][k kq], B lane (kq, lc) = B[k kq][column lc], D lane (kq, lc) register
//    r = C[row 4 kq + r][column lc].  One 16-byte load per lane fills FOUR fragments: op(A): row 16 a + lc, k0 + 4 kq .. + 3 (the
//    four k-steps of a 16-deep chunk); the strip: row k0 + 4 kq + s, columns 4 lc .. 4 lc + 3 (the four column tiles, column
//    j0 + 4 lc + b in tile b).  Per chunk 8 loads and 64 MFMAs; op(A) one chunk ahead (two register sets), the strip three
//    ahead (four), 96 VGPRs beside the 64 accumulator registers.
//  * Triangular op(A): row tiles in pairs (t, last - t) per wavefront, the diagonal block's 4 chunks with compile-time
//    MFMA-tile skipping, as gemm_wave.hip.  Stores: the 16 payloads of a tile (a row's four columns side by side) in 16
//    different register quads, reductions first, stores last, the wavefront waits for them before its next tile (the
//    store-data hazard of gemm_wave.hip).
// Whole aligned problems only (M a multiple of 64, N of 256, float32 scratch for the M x M operand in every problem's xb);
// the launcher returns false otherwise and gemm_strip_f32.hip / gemm_f32.hip run.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef const double __attribute__((address_space(1))) * gcptr;
typedef double __attribute__((address_space(1))) * gptr;
typedef const dbl2 __attribute__((address_space(1))) * gcptr2;
typedef dbl2 __attribute__((address_space(1))) * gptr2;
typedef float __attribute__((address_space(1))) * gfptr;
typedef unsigned int gwf_u4 __attribute__((ext_vector_type(4)));

#define GF_T 64                 // rows and columns of a wavefront's tile
#define GF_CH 16                // k per chunk (four MFMA k-steps)

struct WaveF32Flags {
  int t0, t1, nunits, epi;
  float alpha;        // roles 3 / 5: a power of two, folded into the column scales
  int N;              // frames (row stride of the partial-sum arrays)
  const double* xcols; // role 5: the frames x
};

// the float32 copy of the M x M operand: mode 1 / 3 as is, mode 2 dst[i][k] = Lq[k][i] for k >= i, else 0
__global__ void __launch_bounds__(256) gwf_convert_kernel(const GemmProblem* __restrict__ probs, int mode) {
  const GemmProblem p = probs[blockIdx.y];
  const gcptr src = (gcptr)p.A;
  const gfptr dst = (gfptr)p.xb;
  const int M = p.M;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)M * M; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx / M), k = (int)(idx % M);
    float v;
    if (mode == 2) v = (k >= i) ? (float)src[(int64_t)k * p.lda + i] : 0.0f;
    else v = (float)src[(int64_t)i * p.lda + k];
    dst[idx] = v;
  }
}

// One 64 x 64 tile.  TAG 1: op(A) lower triangular (k < i0 + 64), walked upwards.  TAG 2: op(A) upper triangular (k >= i0),
// walked downwards.  TAG 3: dense.  TAG 5 (KT = a stationary kernel type): TAG 3 with the contraction epilogue.
template <int TAG, int KT = -1>
__device__ __forceinline__ void gwf_tile(const GemmProblem& p, const WaveF32Flags& f, const int i0, const int j0, const int lane,
                                         double* etab = nullptr) {
  constexpr bool DENSE = (TAG == 3 || TAG == 5);
  constexpr bool KDOWN = (TAG == 2);
  const int lc = lane & 15, kq = lane >> 4;
  const int M = p.M;
  int kbeg = 0, kend = p.K;
  if (TAG == 1) kend = i0 + GF_T;
  if (TAG == 2) kbeg = i0;
  const int nch = (kend - kbeg) / GF_CH;
  const int kfirst = KDOWN ? kend - GF_CH : kbeg;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.xb, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7fffffff, 0x00020000);
  int soA[4], soB[4];
#pragma unroll
  for (int a = 0; a < 4; a++) soA[a] = (int)(((int64_t)(i0 + 16 * a) * M + kfirst) * 4);
#pragma unroll
  for (int s = 0; s < 4; s++) soB[s] = (int)(((int64_t)(kfirst + s) * p.ldb + j0) * 4);
  const int voffA = (int)(((int64_t)lc * M + 4 * kq) * 4), voffB = (int)(((int64_t)4 * kq * p.ldb + 4 * lc) * 4);
  const int stepA = (KDOWN ? -1 : 1) * GF_CH * 4, stepB = (KDOWN ? -1 : 1) * GF_CH * (int)p.ldb * 4;
  int nlA = 0, nlB = 0;
  // raw loads only; a request past the last chunk re-reads the last one
  auto load_A = [&](f4 (&Ar)[4]) {
    const int da = (nlA > 0 && nlA < nch) ? stepA : 0;
#pragma unroll
    for (int a = 0; a < 4; a++) soA[a] += da;
    nlA++;
#pragma unroll
    for (int a = 0; a < 4; a++) Ar[a] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rA, voffA, soA[a], 0));
  };
  auto load_B = [&](f4 (&Br)[4]) {
    const int db = (nlB > 0 && nlB < nch) ? stepB : 0;
#pragma unroll
    for (int s = 0; s < 4; s++) soB[s] += db;
    nlB++;
#pragma unroll
    for (int s = 0; s < 4; s++) Br[s] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rB, voffB, soB[s], 0));
  };
  f4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = f4{0.0f, 0.0f, 0.0f, 0.0f};
  // MFMAs of one chunk over row tiles [A_LO, A_HI) (compile-time: straight-line code)
  auto mfma_chunk = [&](const f4 (&Ar)[4], const f4 (&Br)[4], auto lo_tag, auto hi_tag) {
    constexpr int A_LO = decltype(lo_tag)::value, A_HI = decltype(hi_tag)::value;
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int a = A_LO; a < A_HI; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[a][s], Br[s][b], acc[a][b], 0, 0, 0);
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 4> c4{};
#define GF_SB __builtin_amdgcn_sched_barrier(0)
#define GF_CHUNK(AR, BR, LO, HI, LOADS) do { LOADS; GF_SB; mfma_chunk(AR, BR, LO, HI); GF_SB; } while (0)
  // chunk D (0 .. 3, in walking order) of the diagonal block.  TAG 1 (k0 = i0 + 16 D): row tile a is all-zero when
  // k0 > i0 + 16 a + 15 -> tiles [D, 4).  TAG 2 (k0 = i0 + 48 - 16 D, walking down): all-zero when k0 + 15 < i0 + 16 a -> tiles [0, 4 - D)
#define GF_DIAG(AR, BR, D, LOADS) \
  GF_CHUNK(AR, BR, (std::integral_constant<int, (TAG == 1) ? (D) : 0>{}), (std::integral_constant<int, (TAG == 1) ? 4 : 4 - (D)>{}), LOADS)
#define GF_WAIT(N) do { GF_SB; __builtin_amdgcn_s_waitcnt(0x0F70 | (N)); GF_SB; } while (0)
  f4 A0[4], A1[4], B0[4], B1[4], B2[4], B3[4];
  load_B(B0); load_B(B1); load_A(A0); load_B(B2);
  GF_WAIT(4);
  const int nplain = DENSE ? nch : nch - 4;               // chunks outside the diagonal block: a multiple of 4
  for (int c = 0; c < nplain; c += 4) {
    GF_CHUNK(A0, B0, c0, c4, (load_A(A1), load_B(B3)));
    GF_CHUNK(A1, B1, c0, c4, (load_A(A0), load_B(B0)));
    GF_CHUNK(A0, B2, c0, c4, (load_A(A1), load_B(B1)));
    GF_CHUNK(A1, B3, c0, c4, (load_A(A0), load_B(B2)));
    GF_WAIT(4);
  }
  if (!DENSE) {
    GF_DIAG(A0, B0, 0, (load_A(A1), load_B(B3)));
    GF_DIAG(A1, B1, 1, (load_A(A0)));
    GF_DIAG(A0, B2, 2, (load_A(A1)));
    GF_DIAG(A1, B3, 3, (void)0);
  }
#undef GF_DIAG
#undef GF_WAIT
#undef GF_CHUNK

  // ---- epilogue: acc[a][b][r] = C(i0 + 16 a + 4 kq + r, j0 + 4 lc + b) -------------------------------------------------------
  if (DENSE) {
    const gcptr2 gs = (gcptr2)((gcptr)p.v1 + j0 + 4 * lc);
    const dbl2 s0 = gs[0], s1 = gs[1];
    const float sc[4] = {f.alpha * (float)s0.x, f.alpha * (float)s0.y, f.alpha * (float)s1.x, f.alpha * (float)s1.y};
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[a][b][r] *= sc[b];
  }
  if (TAG == 5) {
    // per entry the arithmetic of hyper_contract_kernel<1, false, false, KT> on the value the float32 strip would have held
    etab[lane] = exp2((double)lane * (1.0 / 64.0));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const gcptr th = (gcptr)p.kern.theta;
    const double var = th[0], ls = th[1], inv_ls = 1.0 / ls;
    const gcptr gz = (gcptr)p.xa + i0 + 4 * kq, gal = (gcptr)p.v0 + i0 + 4 * kq;
    const gcptr2 gx = (gcptr2)((gcptr)f.xcols + j0 + 4 * lc), ggm = (gcptr2)((gcptr)p.v2 + j0 + 4 * lc);
    const dbl2 x0 = gx[0], x1 = gx[1], g0 = ggm[0], g1 = ggm[1];
    const double xc[4] = {x0.x, x0.y, x1.x, x1.y}, gmc[4] = {g0.x, g0.y, g1.x, g1.y};
    double acc_v = 0.0, acc_l = 0.0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
      double ra[4], ral[4];
#pragma unroll
      for (int r = 0; r < 4; r++) { ra[r] = gz[16 * a + r] / ls; ral[r] = gal[16 * a + r]; }
#pragma unroll
      for (int b = 0; b < 4; b++) {
        __builtin_amdgcn_sched_barrier(0);
        const double bcol = xc[b] / ls, bb = __dmul_rn(bcol, bcol), gmj = gmc[b];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double av = ra[r], aa = __dmul_rn(av, av);
          const double w = fma(ral[r], gmj, (double)acc[a][b][r]);
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
      const gptr out = (gptr)p.o0 + ((int64_t)(i0 / GF_T) * (f.N / GF_T) + j0 / GF_T) * 2;
      out[0] = acc_v; out[1] = acc_l;
    }
    return;
  }
  double s2[4], sd[4];
  if (f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) {
    // per-column sums over this tile's 64 rows in float64, one partial row per 64-row tile: o0 / o1 [i0 / 64][N]
    double v0r[16];
    if (f.epi & EPI_COLDOT) {
      const gcptr gv0 = (gcptr)p.v0 + i0 + 4 * kq;
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) v0r[4 * a + r] = gv0[16 * a + r];
    }
#pragma unroll
    for (int b = 0; b < 4; b++) {
      s2[b] = 0.0; sd[b] = 0.0;
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double v = (double)acc[a][b][r];
          s2[b] = fma(v, v, s2[b]);
          if (f.epi & EPI_COLDOT) sd[b] = fma(v, v0r[4 * a + r], sd[b]);
        }
      s2[b] += __shfl_xor(s2[b], 16, 64); s2[b] += __shfl_xor(s2[b], 32, 64);
      if (f.epi & EPI_COLDOT) { sd[b] += __shfl_xor(sd[b], 16, 64); sd[b] += __shfl_xor(sd[b], 32, 64); }
    }
  }
  if (f.epi & EPI_STORE) {
    // 16 payloads (a row's four columns side by side) in 16 DIFFERENT register quads, kept alive behind the stores
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, 0x7fffffff, 0x00020000);
    const int voffC = (int)(((int64_t)4 * kq * p.ldc + 4 * lc) * 4);
    f4 pay[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) pay[a][r] = f4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
    GF_SB;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int soC = (int)(((int64_t)(i0 + 16 * a + r) * p.ldc + j0) * 4);      // (scalar)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gwf_u4, pay[a][r]), rC, voffC, soC, 0);
      }
    GF_SB;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) asm volatile("" :: "v"(pay[a][r]));
  }
  if ((f.epi & (EPI_COLSUMSQ | EPI_COLDOT)) && kq == 0) {
    const int64_t prow = (int64_t)(i0 / GF_T) * f.N + j0 + 4 * lc;
    if (f.epi & EPI_COLSUMSQ) { *(gptr2)((gptr)p.o0 + prow) = dbl2{s2[0], s2[1]}; *(gptr2)((gptr)p.o0 + prow + 2) = dbl2{s2[2], s2[3]}; }
    if (f.epi & EPI_COLDOT) { *(gptr2)((gptr)p.o1 + prow) = dbl2{sd[0], sd[1]}; *(gptr2)((gptr)p.o1 + prow + 2) = dbl2{sd[2], sd[3]}; }
  }
  GF_SB;
  __builtin_amdgcn_s_waitcnt(0x0F70);       // the tile's stores have read their registers before the next tile writes them
  GF_SB;
#undef GF_SB
}

#ifndef GWF_OCC
#define GWF_OCC 3          // wavefronts per SIMD the register budget is cut for (168 VGPRs)
#endif
template <int TAG, int KT = -1>
__global__ void __launch_bounds__(256, GWF_OCC) gemm_wave_f32_kernel(const GemmProblem* __restrict__ probs, WaveF32Flags f) {
  __shared__ double etabs[(TAG == 5) ? 4 * 64 : 1];
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
  const int cg = bid / f.nunits;
  const int u = (bid % f.nunits + cg) % f.nunits;
  const int j0 = cg * 256 + wv * GF_T;
  if (TAG == 3 || TAG == 5) {
    gwf_tile<TAG, KT>(p, f, (f.t0 + u) * GF_T, j0, lane, etabs + ((TAG == 5) ? 64 * wv : 0));
  } else {
    const int ta = f.t0 + u, tb = f.t1 - 1 - u;
    const int first = (TAG == 1) ? tb : ta, second = (TAG == 1) ? ta : tb;
    const int npass = (second != first) ? 2 : 1;
#pragma nounroll
    for (int pass = 0; pass < npass; pass++)
      gwf_tile<TAG>(p, f, __builtin_amdgcn_readfirstlane((pass ? second : first) * GF_T), j0, lane);
  }
}

// would a float32 launch of that role and shape take the wave form?  (the caller sizes partial-sum rows and records by it)
bool gemm_wave_f32_takes(int role, int maxM, int maxN, int uniform_aligned) {
  const GpSwitches& sw = gp_switches();
  if (!sw.strip_wave || !uniform_aligned || role < 1 || (role > 3 && role != 5) || !((sw.strip_wave_f32 >> role) & 1)) return false;
  return maxM > 0 && (maxM % GF_T) == 0 && (maxN % 256) == 0;
}

template <int TAG, int KT = -1>
static gp_status gwf_launch(gp_handle h, const GemmProblem* d_probs, int batch, int M, int N, const GemmFlags& f) {
  WaveF32Flags wf;
  const int tiles = M / GF_T;
  wf.t0 = 0; wf.t1 = tiles;
  if (f.tile_m0 > 0 || f.tile_mcount > 0) {
    wf.t0 = 2 * f.tile_m0;
    if (wf.t0 >= tiles) return GP_OK;
    if (f.tile_mcount > 0 && 2 * (f.tile_m0 + f.tile_mcount) < tiles) wf.t1 = 2 * (f.tile_m0 + f.tile_mcount);
  }
  const int nt = wf.t1 - wf.t0;
  wf.nunits = (TAG == 3 || TAG == 5) ? nt : (nt + 1) / 2;
  wf.epi = f.epilogue; wf.alpha = (float)f.alpha; wf.N = N; wf.xcols = f.aux_x;
  int cblocks = (int)(((int64_t)M * M + 255) / 256);
  if (cblocks > 256) cblocks = 256;
  hipLaunchKernelGGL(gwf_convert_kernel, dim3(cblocks, batch), dim3(256), 0, h->stream, d_probs, (TAG == 2) ? 2 : 1);
  hipLaunchKernelGGL((gemm_wave_f32_kernel<TAG, KT>), dim3(wf.nunits * (N / 256), 1, batch), dim3(256), 0, h->stream, d_probs, wf);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// Returns true when the wave form took the launch (*st = its status).  f.a32_ok: every problem's xb points to M * M floats of
// scratch for the float32 copy of its M x M operand; roles 1 / 2 also need f.rows64_ok (partial rows per 64-row tile).
bool launch_gemm_wave_f32(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int maxN, const GemmFlags& f, gp_status* st) {
  if (!f.a32_ok || !gemm_wave_f32_takes(f.role, maxM, maxN, f.uniform_aligned)) return false;
  if ((f.role == 1 || f.role == 2) && !f.rows64_ok) return false;
  if (f.beta != 0.0 || f.triC != TRI_NONE) return false;
  if (f.role >= 3 ? !(f.scale_mode == 1 && (f.alpha == 1.0 || f.alpha == 2.0 || f.alpha == 0.5 || f.alpha == 4.0)) : (f.alpha != 1.0)) return false;
  if (f.role == 5) {
    if (!f.aux_x || f.tile_m0 || f.tile_mcount) { *st = gp_fail(h, GP_ERR_BAD_ARG, "fused float32 Kuf_bar contraction: bad launch"); return true; }
    switch (f.aux_ktype) {
      case GP_KERN_MATERN12: *st = gwf_launch<5, GP_KERN_MATERN12>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_MATERN32: *st = gwf_launch<5, GP_KERN_MATERN32>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_MATERN52: *st = gwf_launch<5, GP_KERN_MATERN52>(h, d_probs, batch, maxM, maxN, f); break;
      case GP_KERN_RBF: *st = gwf_launch<5, GP_KERN_RBF>(h, d_probs, batch, maxM, maxN, f); break;
      default: *st = gp_fail(h, GP_ERR_BAD_ARG, "fused float32 Kuf_bar contraction: not a stationary kernel");
    }
    return true;
  }
  if (f.role == 1) *st = gwf_launch<1>(h, d_probs, batch, maxM, maxN, f);
  else if (f.role == 2) *st = gwf_launch<2>(h, d_probs, batch, maxM, maxN, f);
  else *st = gwf_launch<3>(h, d_probs, batch, maxM, maxN, f);
  return true;
}
