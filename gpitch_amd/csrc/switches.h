// switches.h — every run-time switch of libgpitch_hip.so, in one place.
//
// All of them select between forms that give the SAME results (bit-identical where DESIGN.md says so, rounding-level
// otherwise); they exist for same-box A/B measurements and are read ONCE per process from ONE environment variable:
//
//     GPITCH_AMD_SWITCHES="name=value,name=value,..."        e.g.  GPITCH_AMD_SWITCHES=strip_wave=0,kufbar_split=1
//
//   name            default  meaning
//   strip_wave        1      gemm_wave.hip: float64 strip products with a 64 x 64 tile per wavefront, no LDS (0: gemm_strip.hip)
//   strip_wave_roles  46     bit (1 << role) — which roles (1 A = W Kuf, 2 Lq^T A, 3 Kuf_bar, 5 Kuf_bar + contraction) take it
//   strip_wave_f32    46     gemm_wave_f32.hip: the same form for float32 strips, a bitmask 1 << role (0: gemm_strip_f32.hip / gemm_f32.hip)
//   strip_lean        1      gemm_strip.hip / gemm_strip_f32.hip: the lean 128 x 128 LDS tiles (0: gemm.hip / gemm_f32.hip's)
//   hyper_fuse        1      a stationary family's Kuf-side contraction as the epilogue of its Kuf_bar product
//   kufbar_split     -1      Kuf_bar per kernel family: 2 stationary family first, 1 spectral-mixture family first, 0 one launch, -1: 1 with float32 strips, else 2
//   cond_a_early      1      first row-block of A = W Kuf underneath the block-row inverse of the Kuu factorisation
//   blocked_256       1      M in (128, 256] with long batches: resident factor + blocked inverse (0: one fused kernel)
//   cov_sum           1      SGPRSS: the kernel sum K = sum_p K_p built in one pass (0: one accumulate launch per kernel)
//   hyper_sum         1      SGPRSS: the P kernels' Kuf-side contractions in one pass over Kuf_bar
//   chol_cluster      1      chol_cluster.hip: M x M factor + inverse (128 <= M <= 512, M % 32 == 0) by a cluster of 17 / 5 / 2 workgroups per matrix, up to 128 per launch (0: chol.hip)
//   aux_priority      1      the handle's helper stream is created with the highest stream priority (0: default priority)
//   gemm_tile32       320    generic M x M products with fewer 64 x 64 tiles than this in the launch take 32 x 32 tiles (0: never)
//   nt_cover          0      split-K product: workgroups the K-slices are chosen to cover (0: 1024 below 128 output tiles in the launch, else 2048)
//
// Unknown names are reported once on stderr and ignored.
#pragma once
struct GpSwitches {
  int strip_wave = 1, strip_wave_f32 = (1 << 1) | (1 << 2) | (1 << 3) | (1 << 5), strip_wave_roles = (1 << 1) | (1 << 2) | (1 << 3) | (1 << 5), strip_lean = 1, hyper_fuse = 1, kufbar_split = -1,
      cond_a_early = 1, blocked_256 = 1, cov_sum = 1, hyper_sum = 1, chol_cluster = 1, aux_priority = 1, gemm_tile32 = 320, nt_cover = 0;
};
const GpSwitches& gp_switches();     // abi.hip
