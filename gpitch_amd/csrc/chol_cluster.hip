// chol_cluster.hip — Cholesky factor L and inverse W = L^-1 of ONE Kuu-sized matrix by a CLUSTER of workgroups (gfx950).
//
// Replaces tf.cholesky / tf.matrix_triangular_solve on the M x M inducing covariance where a single matrix is on the
// critical path of an evaluation (gpitch/sgpr_ss.py:44,48: Kuu and B = I + A A^T / sigma^2 of the collapsed bound).
// The one-workgroup kernels of chol.hip keep one CU busy for 0.43 ms at M = 512 (plus 0.2 ms of dependent launches for
// the inverse): half of that is the serial chain of 32 x 32 diagonal blocks, the other half the trailing updates that
// the same CU has to fit around it.  Here the chain gets a wavefront of its own and everything else is dataflow:
//
//   * 32 x 32 tiles; G workgroups of four wavefronts (one per SIMD) per matrix, G by the tile count (17 at M = 512, 5 at 256, 2 at
//     128).  Wavefront 0 of workgroup 0 is the CHAIN: for s = 0 .. T-1 it factorises tile (s, s) in registers
//     (chol_diag_block), solves the tile below it, L(s+1, s), and applies the last update to tile (s+1, s+1) — on LDS and
//     registers only.  Wavefront 1 is its COURIER: it fetches tiles (s+1, s) and (s+1, s+1) as the workers left them, adds the
//     update of column s-1 and leaves them in LDS for the chain; it publishes L_ss^-1, L(s, s), W(s, s) and L(s+1, s).
//   * every other wavefront is a WORKER that owns up to two tiles of L and two of W, held in MFMA accumulators for
//     the whole factorisation (right-looking: when column s of L is known, every owned tile right of it takes its
//     update; a tile of column s is solved against L_ss^-1 and published).  Tiles (k, k-1) and (k, k) are handed to the
//     courier two columns early, so the chain never waits for an exchange that started in its own period.
//   * W follows in the shadow of the chain: S(i, j) = sum_k L(i, k) W(k, j) accumulates as rows of W are published,
//     W(i, j) = -L_ii^-1 S(i, j) when L_ii^-1 is.
//   * exchange through memory: every exchanged value is stored and loaded with agent-scope relaxed atomics (sc1: written
//     through / read past the XCD's L2 — no cache write-back or invalidate, which would also flush the strips other
//     kernels are writing), a flag per tile announces it (store data, s_waitcnt vmcnt(0), store flag; poll flag, load
//     data).  Flags hold the launch's epoch + 1: nothing is cleared between launches, the last wavefront to finish bumps
//     the epoch (so a recorded launch sequence replays without a memset node).  Inside workgroup 0 the chain and its courier
//     hand over through counters in LDS.
//   * every wait is bounded: after ~2 s of polling a wavefront raises the cluster's abort word, all others leave their
//     loops at their next check, and the launch reports GP_ERR_HIP through the status word instead of hanging the GPU.
//     All wavefronts of a cluster must be resident for it to make progress: the launcher only takes launches of at most
//     CC_MAX_WGS workgroups (a 256-CU device holds 256 of them at this kernel's register count), everything else goes to
//     chol.hip.  DESIGN.md section 3.01 has the measurements.
#include "common.h"
#include "switches.h"
#include "chol_diag.h"
#include <map>
#include <mutex>
#include <vector>

#define CC_T 32
#define CC_MAXT 16                 // M <= 512
#define CC_THREADS 256
#define CC_G_MAX 17                // workgroups per matrix at M = 512: 66 workers, at most two tiles of L and two of W each
#define CC_LSLOTS 2
#define CC_WSLOTS 2
#define CC_MAX_WGS 128
// control block (ints), then the published L_ss^-1 tiles (row-major 32 x 32 doubles each)
#define CC_EPOCH 0
#define CC_ARRIVE 1
#define CC_ABORT 2
#define CC_FLAG_D 16                                   // [CC_MAXT]        L_ss^-1 (and W(s, s)) published
#define CC_FLAG_F 32                                   // [CC_MAXT][CC_MAXT] L(i, j) published
#define CC_FLAG_P (CC_FLAG_F + CC_MAXT * CC_MAXT)      // [CC_MAXT][2]     tiles (k, k-1), (k, k) handed to the chain
#define CC_FLAG_W (CC_FLAG_P + 2 * CC_MAXT)            // [CC_MAXT][CC_MAXT] W(i, j) published
#define CC_CTL_BYTES 4096
#define CC_SCRATCH_BYTES (CC_CTL_BYTES + CC_MAXT * CC_T * CC_T * 8)
static_assert((CC_FLAG_W + CC_MAXT * CC_MAXT) * 4 <= CC_CTL_BYTES, "control block");

#ifdef CC_STAMPS
// diagnostic build (never shipped; tools/chol_cluster_stamps.py): s_memtime stamps of the chain, 8 per period, and of every
// worker at the end of each of its steps
__device__ unsigned long long cc_stamps[CC_MAXT * 8 + 64 * CC_MAXT];
extern "C" int gp_debug_chol_cluster_stamps(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(cc_stamps), sizeof(cc_stamps)) == hipSuccess ? 0 : -1;
}
#define CC_STAMP(s, i) do { if (lane == 0) cc_stamps[8 * (s) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define CC_WSTAMP(w, s) do { if (lane == 0 && (w) < 64) cc_stamps[CC_MAXT * 8 + (w) * CC_MAXT + (s)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CC_STAMP(s, i) do { } while (0)
#define CC_WSTAMP(w, s) do { } while (0)
#endif

__device__ __forceinline__ int cc_flag_load(const int* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void cc_flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double cc_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cc_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// this wavefront's stores have been written through; then the flag that announces them
__device__ __forceinline__ void cc_publish(int* flag, int want, int lane) {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  asm volatile("" ::: "memory");
  if (lane == 0) cc_flag_store(flag, want);
}

// poll until *flag == want; false = the cluster was aborted (by this wavefront after ~2 s, or by another one)
__device__ __forceinline__ bool cc_wait(const int* flag, int want, int* ctl) {
  int spins = 0;
  while (cc_flag_load(flag) != want) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255) == 0) {
      if (cc_flag_load(ctl + CC_ABORT) == want) return false;
      if (spins > (1 << 21)) { cc_flag_store(ctl + CC_ABORT, want); return false; }
    }
  }
  asm volatile("" ::: "memory");
  return true;
}

// rows of tile (ti, tp) of a row-major matrix as MFMA fragments: f[h][kk] = T[16 h + lc][4 kk + kq]
__device__ __forceinline__ void cc_load_rows(const double* base, int64_t ld, int ti, int tp, int lc, int kq, double (&f)[2][8]) {
  const double* p = base + (int64_t)(CC_T * ti + lc) * ld + CC_T * tp + kq;
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int kk = 0; kk < 8; kk++) f[h][kk] = cc_ld(p + (int64_t)16 * h * ld + 4 * kk);
}
// columns of a row-major 32 x 32 block as MFMA fragments: f[h][kk] = T[4 kk + kq][16 h + lc]
__device__ __forceinline__ void cc_load_cols(const double* tile, int64_t ld, int lc, int kq, double (&f)[2][8]) {
  const double* p = tile + (int64_t)kq * ld + lc;
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int kk = 0; kk < 8; kk++) f[h][kk] = cc_ld(p + (int64_t)4 * kk * ld + 16 * h);
}
// L_ss^-1 (lower triangular, row-major 32 x 32, stride ld) as the A operand of a product that contracts over the ROWS of an
// accumulator tile: g[3 blocks (0,0), (1,0), (1,1)][r] = Dinv[16 a2 + lc][16 a + 4 r + kq]
template <typename PT>
__device__ __forceinline__ void cc_load_dinv(PT dinv, int ld, int lc, int kq, double (&g)[3][4], bool coherent) {
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int a2 = (q == 0) ? 0 : 1, a = (q == 2) ? 1 : 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int off = (16 * a2 + lc) * ld + 16 * a + 4 * r + kq;
      g[q][r] = coherent ? cc_ld((const double*)dinv + off) : dinv[off];
    }
  }
}
// out[a2][b] = -(Dinv acc)[block a2][block b] for an accumulator tile acc[a][b][r] = X[16 a + kq + 4 r][16 b + lc]
__device__ __forceinline__ void cc_apply_dinv_neg(const double (&g)[3][4], const d4 (&acc)[2][2], d4 (&out)[2][2]) {
#pragma unroll
  for (int b = 0; b < 2; b++) {
    out[0][b] = d4{0.0, 0.0, 0.0, 0.0}; out[1][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      out[0][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[0][r], acc[0][b][r], out[0][b], 0, 0, 1);
      out[1][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[1][r], acc[0][b][r], out[1][b], 0, 0, 1);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) out[1][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(g[2][r], acc[1][b][r], out[1][b], 0, 0, 1);
  }
}
// acc[a][b] += sum_kk fa[a][kk] fb[b][kk]
__device__ __forceinline__ void cc_product(const double (&fa)[2][8], const double (&fb)[2][8], d4 (&acc)[2][2]) {
#pragma unroll
  for (int kk = 0; kk < 8; kk++)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][kk], fb[b][kk], acc[a][b], 0, 0, 0);
}

// Tiles of L in a worker's (and the chain's) accumulators are held TRANSPOSED and NEGATED:
//   acc[a][b][r] = -(C^T)[16 a + kq + 4 r][16 b + lc] = -C[16 b + lc][16 a + kq + 4 r],   C = A(i, j) - sum_p L(i, p) L(j, p)^T,
// so that the solve X = C L_jj^-T is X^T = L_jj^-1 C^T: the accumulator registers are the B operand as they are.
__device__ __forceinline__ void cc_load_tile_negT(const double* A, int64_t ld, int ti, int tj, int lc, int kq, d4 (&acc)[2][2], bool coherent) {
  const double* p = A + (int64_t)(CC_T * ti + lc) * ld + CC_T * tj + kq;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double* q = p + (int64_t)16 * b * ld + 16 * a + 4 * r;
        acc[a][b][r] = -(coherent ? cc_ld(q) : *q);
      }
}
// X^T in accumulator layout (x[a2][b][r] = X[16 b + lc][16 a2 + kq + 4 r]) -> tile (ti, tj), sign * x
__device__ __forceinline__ void cc_store_tile_T(double* A, int64_t ld, int ti, int tj, int lc, int kq, const d4 (&x)[2][2], double sign) {
  double* p = A + (int64_t)(CC_T * ti + lc) * ld + CC_T * tj + kq;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) cc_st(p + (int64_t)16 * b * ld + 16 * a + 4 * r, sign * x[a][b][r]);
}
// a tile of zeros (the mirror image above the diagonal: L and W are handed on as dense operands)
__device__ __forceinline__ void cc_zero_tile(double* A, int64_t ld, int ti, int tj, int lane) {
  double* p = A + (int64_t)(CC_T * ti + (lane >> 1)) * ld + CC_T * tj + 16 * (lane & 1);
#pragma unroll
  for (int c = 0; c < 16; c++) p[c] = 0.0;
}

// tile number t (column by column, the tiles the chain keeps to itself left out) -> (i, j); false past the end
__device__ __forceinline__ bool cc_ltile(int t, int T, int& ti, int& tj) {
  for (int j = 0; j < T; j++)
    for (int i = j; i < T; i++) {
      if (i <= 2 && j >= i - 1) continue;          // (0,0) (1,0) (1,1) (2,1) (2,2): never leave the chain
      if (t == 0) { ti = i; tj = j; return true; }
      t--;
    }
  return false;
}
__device__ __forceinline__ bool cc_wtile(int u, int T, int& ti, int& tj) {
  for (int j = 0; j < T; j++) {
    const int n = T - 1 - j;
    if (u < n) { ti = j + 1 + u; tj = j; return true; }
    u -= n;
  }
  return false;
}

// workgroup-local hand-off between the chain and its courier: a counter in LDS, release / acquire at workgroup scope
__device__ __forceinline__ void cc_lds_post(int* flag, int v, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool cc_lds_wait(int* flag, int atleast, int want, int* ctl) {
  int spins = 0;
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < atleast) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023) == 0) {
      if (cc_flag_load(ctl + CC_ABORT) == want) return false;
      if (spins > (1 << 24)) { cc_flag_store(ctl + CC_ABORT, want); return false; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}

template <bool WITH_W>
__global__ void __launch_bounds__(CC_THREADS) chol_cluster_kernel(double* const* __restrict__ mats, double* const* __restrict__ Ws,
                                                                  const int* __restrict__ Ms, const int* __restrict__ lds_,
                                                                  double* single_A, double* single_W, int single_M, int single_ld,
                                                                  char* scratch, int* __restrict__ status, int pivot_base) {
  // workgroup 0 only: the chain's tiles and what it hands to / takes from its courier
  __shared__ __attribute__((aligned(16))) double Dg[2][CC_T][CH_LDP];     // tile (s, s): in, and L(s, s) out, by parity of s
  __shared__ __attribute__((aligned(16))) double Dinv[2][CC_T][CH_LDP];   // L_ss^-1 by parity of s
  __shared__ __attribute__((aligned(16))) double Xl[CC_T][CH_LDP];        // L(s+1, s), row-major
  __shared__ __attribute__((aligned(16))) double Qr[2][16][64];           // tiles (s+1, s), (s+1, s+1) as accumulator registers
  __shared__ int lflag[4];                                                // [0] diagonal blocks done, [1] X tiles written, [2] Q sets ready
  const int mat = blockIdx.y;
  double* const A = mats ? mats[mat] : single_A;
  double* const W = WITH_W ? (Ws ? Ws[mat] : single_W) : nullptr;
  const int64_t ld = mats ? lds_[mat] : single_ld;
  const int T = (mats ? Ms[mat] : single_M) / CC_T;
  int* const ctl = (int*)(scratch + (size_t)mat * CC_SCRATCH_BYTES);
  double* const dinv_g = (double*)((char*)ctl + CC_CTL_BYTES);
  const int lane = threadIdx.x & 63, lc = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wid = (int)blockIdx.x * 4 + wave;          // 0 = the chain, 1 = its courier
  const int nwaves = (int)gridDim.x * 4, NW = nwaves - 2;
  const int want = cc_flag_load(ctl + CC_EPOCH) + 1;
  int* const fD = ctl + CC_FLAG_D;
  int* const fF = ctl + CC_FLAG_F;
  int* const fP = ctl + CC_FLAG_P;
  int* const fW = ctl + CC_FLAG_W;
  bool ok = true;
  if (blockIdx.x == 0) {
    if (threadIdx.x < 4) lflag[threadIdx.x] = 0;
    __syncthreads();
  }

  if (wid == 0) {
    // ------------------------------------------------ the chain: LDS and registers only ---------------------------------
    {   // tile (0, 0) -> Dg[0]
      const int row = lane >> 1, c0 = 16 * (lane & 1);
#pragma unroll
      for (int c = 0; c < 16; c++) Dg[0][row][c0 + c] = A[(int64_t)row * ld + c0 + c];
    }
    for (int s = 0; s < T; s++) {
      const int pb = s & 1;
      CC_STAMP(s, 0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      chol_diag_block<double*>(&Dg[pb][0][0], CH_LDP, CC_T, 0, lane, Dinv[pb], status, mat, pivot_base + CC_T * s);
      CC_STAMP(s, 1);
      cc_lds_post(&lflag[0], s + 1, lane);
      if (s + 1 >= T) break;
      // tiles (s+1, s) and (s+1, s+1) with every update but the last, as the courier's accumulators
      ok = cc_lds_wait(&lflag[2], s + 1, want, ctl);
      if (!ok) break;
      CC_STAMP(s, 2);
      d4 c1[2][2], c2[2][2];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) { c1[a][b][r] = Qr[0][8 * a + 4 * b + r][lane]; c2[a][b][r] = Qr[1][8 * a + 4 * b + r][lane]; }
      // L(s+1, s): X^T = L_ss^-1 C^T(s+1, s)
      double g[3][4];
      cc_load_dinv(&Dinv[pb][0][0], CH_LDP, lc, kq, g, false);
      d4 x[2][2];
      cc_apply_dinv_neg(g, c1, x);
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) Xl[16 * b + lc][16 * a + kq + 4 * r] = x[a][b][r];
      cc_lds_post(&lflag[1], s + 1, lane);
      CC_STAMP(s, 3);
      {   // tile (s+1, s+1) -= L(s+1, s) L(s+1, s)^T, then into Dg for the next factorisation
        double fx[2][8];
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
          for (int r = 0; r < 4; r++) { fx[0][4 * a2 + r] = x[a2][0][r]; fx[1][4 * a2 + r] = x[a2][1][r]; }
        cc_product(fx, fx, c2);
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) Dg[pb ^ 1][16 * b + lc][16 * a + kq + 4 * r] = -c2[a][b][r];
      }
      CC_STAMP(s, 4);
    }
  } else if (wid == 1) {
    // ------------------------------------------------ the courier: everything the chain exchanges with memory -----------
    for (int s = 0; s < T && ok; s++) {
      const int pb = s & 1, i = s + 1;
      if (i < T) {
        // tiles (i, s) and (i, i) as the workers left them (their updates p <= i - 3), the update p = s - 1 here
        if (i >= 3) { ok = cc_wait(fP + 2 * i, want, ctl) && cc_wait(fP + 2 * i + 1, want, ctl); if (!ok) break; }
        d4 c1[2][2], c2[2][2];
        cc_load_tile_negT(A, ld, i, s, lc, kq, c1, true);
        cc_load_tile_negT(A, ld, i, i, lc, kq, c2, true);
        if (s >= 1) {
          ok = cc_wait(fF + i * CC_MAXT + (s - 1), want, ctl);
          if (!ok) break;
          double fi[2][8], fj[2][8];
          cc_load_rows(A, ld, i, s - 1, lc, kq, fi);
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int kk = 0; kk < 8; kk++) fj[h][kk] = Xl[16 * h + lc][4 * kk + kq];      // L(s, s-1): this wavefront saw lflag[1] >= s
          cc_product(fj, fi, c1);          // -C^T(i, s) += L(s, s-1) L(i, s-1)^T
          cc_product(fi, fi, c2);
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) { Qr[0][8 * a + 4 * b + r][lane] = c1[a][b][r]; Qr[1][8 * a + 4 * b + r][lane] = c2[a][b][r]; }
        cc_lds_post(&lflag[2], s + 1, lane);
      }
      // L_ss^-1 into the exchange area (whole lines per store instruction), then L(s, s) and W(s, s) into the matrices
      ok = cc_lds_wait(&lflag[0], s + 1, want, ctl);
      if (!ok) break;
      {
        double* pd = dinv_g + (size_t)s * (CC_T * CC_T);
#pragma unroll
        for (int c = 0; c < 16; c++) { const int e = 64 * c + lane; cc_st(pd + e, Dinv[pb][e >> 5][e & 31]); }
      }
      cc_publish(fD + s, want, lane);
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int e = 64 * c + lane, row = e >> 5, col = e & 31;
        A[(int64_t)(CC_T * s + row) * ld + CC_T * s + col] = (col <= row) ? Dg[pb][row][col] : 0.0;
        if (WITH_W) W[(int64_t)(CC_T * s + row) * ld + CC_T * s + col] = Dinv[pb][row][col];
      }
      if (i < T) {
        ok = cc_lds_wait(&lflag[1], s + 1, want, ctl);
        if (!ok) break;
#pragma unroll
        for (int c = 0; c < 16; c++) {
          const int e = 64 * c + lane, row = e >> 5, col = e & 31;
          cc_st(A + (int64_t)(CC_T * i + row) * ld + CC_T * s + col, Xl[row][col]);
        }
        cc_publish(fF + i * CC_MAXT + s, want, lane);
        cc_zero_tile(A, ld, s, i, lane);
      }
    }
  } else {
    // ------------------------------------------------ a worker -------------------------------------------------------
    const int w = wid - 2;
    int li[CC_LSLOTS], lj[CC_LSLOTS], wi[CC_WSLOTS], wj[CC_WSLOTS];
    d4 lacc[CC_LSLOTS][2][2], sacc[CC_WSLOTS][2][2];
#pragma unroll
    for (int q = 0; q < CC_LSLOTS; q++) {
      int ti = -1, tj = -1;
      if (!cc_ltile(w + q * NW, T, ti, tj)) ti = tj = -1;
      li[q] = __builtin_amdgcn_readfirstlane(ti); lj[q] = __builtin_amdgcn_readfirstlane(tj);
      if (li[q] >= 0) cc_load_tile_negT(A, ld, li[q], lj[q], lc, kq, lacc[q], false);
    }
#pragma unroll
    for (int q = 0; q < CC_WSLOTS; q++) {
      int ti = -1, tj = -1;
      if (!WITH_W || !cc_wtile((NW - 1 - w) + q * NW, T, ti, tj)) ti = tj = -1;
      wi[q] = __builtin_amdgcn_readfirstlane(ti); wj[q] = __builtin_amdgcn_readfirstlane(tj);
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) sacc[q][a][b] = d4{0.0, 0.0, 0.0, 0.0};
    }
    for (int s = 0; s < T && ok; s++) {
      // ---- A: owned tiles of column s are solved against L_ss^-1 and published --------------------------------------
      // ---- C: owned tiles of row s of W are finished the same way ---------------------------------------------------
      bool need_d = false;
#pragma unroll
      for (int q = 0; q < CC_LSLOTS; q++) need_d = need_d || (li[q] >= 0 && lj[q] == s);
#pragma unroll
      for (int q = 0; q < CC_WSLOTS; q++) need_d = need_d || (wi[q] == s);
      if (need_d) {
        ok = cc_wait(fD + s, want, ctl);
        if (!ok) break;
        double g[3][4];
        cc_load_dinv(dinv_g + (size_t)s * (CC_T * CC_T), CC_T, lc, kq, g, true);
#pragma unroll
        for (int q = 0; q < CC_LSLOTS; q++)
          if (li[q] >= 0 && lj[q] == s) {
            d4 x[2][2];
            cc_apply_dinv_neg(g, lacc[q], x);
            cc_store_tile_T(A, ld, li[q], s, lc, kq, x, 1.0);
            cc_publish(fF + li[q] * CC_MAXT + s, want, lane);
            cc_zero_tile(A, ld, s, li[q], lane);
            li[q] = -1;
          }
#pragma unroll
        for (int q = 0; q < CC_WSLOTS; q++)
          if (wi[q] == s) {
            // W(s, j) = -L_ss^-1 S, S in natural layout: out[a2][b][r] = W[16 a2 + kq + 4 r][16 b + lc]
            d4 o[2][2];
            cc_apply_dinv_neg(g, sacc[q], o);
            double* p = W + (int64_t)(CC_T * s + kq) * ld + CC_T * wj[q] + lc;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
              for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) cc_st(p + (int64_t)(16 * a + 4 * r) * ld + 16 * b, o[a][b][r]);
            cc_publish(fW + s * CC_MAXT + wj[q], want, lane);
            cc_zero_tile(W, ld, wj[q], s, lane);
            wi[q] = -1;
          }
      }
      if (s + 1 >= T) break;
      // ---- B: owned tiles right of column s take its update (the chain's side applies the last two of its own tiles);
      //      all flags first, then all operand loads, then the products ------------------------------------------------
#pragma unroll
      for (int q0 = 0; q0 < CC_LSLOTS; q0 += 2) {        // (two tiles' operands in flight at a time: registers)
        bool act[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int q = q0 + u;
          act[u] = false;
          if (q >= CC_LSLOTS) continue;
          const int i = li[q], j = lj[q];
          act[u] = !(i < 0 || j <= s || (i - s <= 2 && j >= i - 1));
          if (act[u] && ok) {
            ok = cc_wait(fF + i * CC_MAXT + s, want, ctl);
            if (ok && j != i) ok = cc_wait(fF + j * CC_MAXT + s, want, ctl);
          }
        }
        if (!ok) break;
        double fi[2][2][8], fj[2][2][8];
#pragma unroll
        for (int u = 0; u < 2; u++)
          if (act[u]) {
            const int q = (q0 + u < CC_LSLOTS) ? q0 + u : 0;
            cc_load_rows(A, ld, li[q], s, lc, kq, fi[u]);
            if (lj[q] != li[q]) cc_load_rows(A, ld, lj[q], s, lc, kq, fj[u]);
          }
#pragma unroll
        for (int u = 0; u < 2; u++)
          if (act[u]) {
            const int q = (q0 + u < CC_LSLOTS) ? q0 + u : 0;
            const int i = li[q], j = lj[q];
            if (j != i) cc_product(fj[u], fi[u], lacc[q]);
            else cc_product(fi[u], fi[u], lacc[q]);
            if (j >= i - 1 && s == i - 3) {      // the chain's side takes it from here: C = -acc^T back into the matrix
              cc_store_tile_T(A, ld, i, j, lc, kq, lacc[q], -1.0);
              cc_publish(fP + 2 * i + (j - (i - 1)), want, lane);
              li[q] = -1;
            }
          }
      }
      if (!ok) break;
      // ---- D: S(i, j) += L(i, s) W(s, j) for owned tiles of W with j <= s < i ----------------------------------------
      if (WITH_W) {
        bool act[CC_WSLOTS];
#pragma unroll
        for (int q = 0; q < CC_WSLOTS; q++) {
          const int i = wi[q], j = wj[q];
          act[q] = !(i < 0 || j > s || i <= s);
          if (act[q] && ok) {
            ok = cc_wait(fF + i * CC_MAXT + s, want, ctl);
            if (ok) ok = (j == s) ? cc_wait(fD + s, want, ctl) : cc_wait(fW + s * CC_MAXT + j, want, ctl);
          }
        }
        if (!ok) break;
#pragma unroll
        for (int q = 0; q < CC_WSLOTS; q++)
          if (act[q]) {
            double fi[2][8], fw[2][8];
            cc_load_rows(A, ld, wi[q], s, lc, kq, fi);
            if (wj[q] == s) cc_load_cols(dinv_g + (size_t)s * (CC_T * CC_T), CC_T, lc, kq, fw);
            else cc_load_cols(W + (int64_t)(CC_T * s) * ld + CC_T * wj[q], ld, lc, kq, fw);
            cc_product(fi, fw, sacc[q]);
          }
      }
      CC_WSTAMP(w, s);
    }
  }
  // ---- the end: the last wavefront to arrive opens the next epoch -----------------------------------------------------
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if (lane == 0) {
    if (!ok) atomicCAS(&status[0], 0, 2);          // 2: the cluster gave up waiting (check_not_pd reports it)
    const int old = __hip_atomic_fetch_add(ctl + CC_ARRIVE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwaves - 1) {
      if (cc_flag_load(ctl + CC_ABORT) == want) atomicCAS(&status[0], 0, 2);
      cc_flag_store(ctl + CC_ARRIVE, 0);
      cc_flag_store(ctl + CC_EPOCH, want);
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// exchange areas, one per (handle, first matrix pointer): allocated and zeroed on the first launch for that matrix (never
// inside a stream capture: the launcher then declines and the caller takes the one-workgroup kernels), freed with the handle
struct CcPool {
  std::mutex mu;
  std::map<std::pair<gp_handle, const void*>, std::pair<char*, int>> areas;
  std::vector<std::pair<gp_handle, const void*>> order;       // allocation order, for the cap below
  std::map<std::pair<gp_handle, const void*>, bool> pinned;   // handed to a launch that was being recorded: lives as long as the handle
};
// a process that keeps creating plans (a model per window, each with a workspace of its own) would otherwise collect one
// 133-KB area per matrix address it has ever factorised: past this many areas the older half is released
#define CC_POOL_CAP 64
static CcPool& cc_pool() { static CcPool p; return p; }

void cholesky_cluster_release(gp_handle h) {
  CcPool& P = cc_pool();
  std::lock_guard<std::mutex> g(P.mu);
  for (auto it = P.areas.begin(); it != P.areas.end();) {
    if (it->first.first == h) { (void)hipFree(it->second.first); it = P.areas.erase(it); }
    else ++it;
  }
  for (auto it = P.order.begin(); it != P.order.end();) it = (it->first == h) ? P.order.erase(it) : it + 1;
  for (auto it = P.pinned.begin(); it != P.pinned.end();) it = (it->first.first == h) ? P.pinned.erase(it) : std::next(it);
}

static char* cc_area(gp_handle h, const void* key, int count) {
  CcPool& P = cc_pool();
  std::lock_guard<std::mutex> g(P.mu);
  auto it = P.areas.find({h, key});
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = h->stream && hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (it != P.areas.end() && it->second.second >= count) {
    if (capturing) P.pinned[{h, key}] = true;        // the recorded launch keeps this address
    return it->second.first;
  }
  if (capturing) return nullptr;
  if (it != P.areas.end()) {
    (void)hipStreamSynchronize(h->stream); (void)hipFree(it->second.first); P.areas.erase(it);
    for (auto o = P.order.begin(); o != P.order.end();) o = (*o == std::make_pair(h, key)) ? P.order.erase(o) : o + 1;
  }
  if (P.order.size() >= CC_POOL_CAP + P.pinned.size()) {      // (not capturing here: the whole device may be drained)
    (void)hipDeviceSynchronize();
    std::vector<std::pair<gp_handle, const void*>> keep;
    for (size_t q = 0; q < P.order.size(); q++) {
      const bool old_half = q < CC_POOL_CAP / 2;
      if (old_half && !P.pinned.count(P.order[q])) {
        auto old = P.areas.find(P.order[q]);
        if (old != P.areas.end()) { (void)hipFree(old->second.first); P.areas.erase(old); }
      } else {
        keep.push_back(P.order[q]);
      }
    }
    P.order.swap(keep);
  }
  char* p = nullptr;
  if (hipMalloc(&p, (size_t)count * CC_SCRATCH_BYTES) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  // zeroed ON the handle's stream and waited for: a memset on the null stream is not ordered against a non-blocking stream, and
  // recycled device memory is not zero — the first launch then read a stale epoch in some wavefronts and a cleared one in others
  if (hipMemsetAsync(p, 0, (size_t)count * CC_SCRATCH_BYTES, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
    (void)hipFree(p); (void)hipGetLastError(); return nullptr;
  }
  P.areas[{h, key}] = {p, count};
  P.order.push_back({h, key});
  return p;
}

// workgroups per matrix: enough workers that none owns more than CC_LSLOTS tiles of L and CC_WSLOTS of W
static int cc_groups(int M) {
  const int T = M / CC_T;
  const int nl = T * (T + 1) / 2 - 5, nw = T * (T - 1) / 2;
  const int wl = (nl + CC_LSLOTS - 1) / CC_LSLOTS, ww = (nw + CC_WSLOTS - 1) / CC_WSLOTS;
  const int waves = 2 + (wl > ww ? wl : ww);
  return (waves + 3) / 4;
}

bool cholesky_cluster_takes(int M, int count) {
  return gp_switches().chol_cluster != 0 && M >= 4 * CC_T && M <= CC_MAXT * CC_T && (M % CC_T) == 0 && count >= 1 &&
         count * cc_groups(M) <= CC_MAX_WGS;
}

// A -> L in place (zeros above the diagonal) and, when W is given, W = L^-1 (zeros above the diagonal), one launch.
// Returns false when the launch is not one the cluster takes (the caller goes on to the one-workgroup kernels).
bool launch_cholesky_cluster_single(gp_handle h, double* A, double* W, int M, int64_t ld, int pivot_base, gp_status* st) {
  if (!cholesky_cluster_takes(M, 1) || ld < M || ld > (1 << 20)) return false;
  char* area = cc_area(h, A, 1);
  if (!area) return false;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  if (W)
    hipLaunchKernelGGL(chol_cluster_kernel<true>, dim3(CC_G_MAX, 1), dim3(CC_THREADS), 0, h->stream, (double* const*)nullptr,
                       (double* const*)nullptr, (const int*)nullptr, (const int*)nullptr, A, W, M, (int)ld, area, h->d_status, pivot_base);
  else
    hipLaunchKernelGGL(chol_cluster_kernel<false>, dim3(CC_G_MAX, 1), dim3(CC_THREADS), 0, h->stream, (double* const*)nullptr,
                       (double* const*)nullptr, (const int*)nullptr, (const int*)nullptr, A, (double*)nullptr, M, (int)ld, area, h->d_status,
                       pivot_base);
  *st = (hipGetLastError() == hipSuccess) ? GP_OK : gp_fail(h, GP_ERR_HIP, "chol_cluster_kernel launch failed");
  return true;
}

// `count` matrices (device arrays of pointers, sizes and row strides; every size a multiple of 32 in [128, 512] — the caller
// vouches for it, minM / maxM are its host copies of the range), factor and inverse of each by its own cluster, one launch
bool launch_cholesky_cluster_batched(gp_handle h, double* const* d_mats, double* const* d_W, const int* d_M, const int* d_ld,
                                     int count, int minM, int maxM, gp_status* st) {
  if (!cholesky_cluster_takes(minM, count) || !cholesky_cluster_takes(maxM, count) || !d_mats || !d_W) return false;
  char* area = cc_area(h, d_mats, count);
  if (!area) return false;
  GpTimerScope ts(h, GP_TIMER_CHOL);
  hipLaunchKernelGGL(chol_cluster_kernel<true>, dim3(cc_groups(maxM), count), dim3(CC_THREADS), 0, h->stream, d_mats, d_W, d_M, d_ld,
                     (double*)nullptr, (double*)nullptr, 0, 0, area, h->d_status, 0);
  *st = (hipGetLastError() == hipSuccess) ? GP_OK : gp_fail(h, GP_ERR_HIP, "chol_cluster_kernel launch failed");
  return true;
}
