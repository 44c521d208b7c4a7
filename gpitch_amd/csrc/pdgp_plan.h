// pdgp_plan.h — Pdgp plan object shared by pdgp.hip (forward) and bwd.hip (backward).
#pragma once
#include "engine.h"

static inline size_t pdgp_kl_region_bytes(int G) {
  const size_t a = kl_item_bytes(), b = klu_item_bytes();
  return gp_align_up((size_t)G * (a > b ? a : b), 256);
}

// descriptor slot (of 24) holding the trace-term problems of the unwhitened KL
static inline size_t pdgp_kltr_offset(int G) {
  return pdgp_kl_region_bytes(G) + 23 * gp_align_up((size_t)G * sizeof(GemmProblem), 256);
}

struct PdgpGP {
  int M = 0, ktype = 0, m = 0;
  int need_theta = 1, need_z = 1;   // gp_pdgp_set_grad_needs
  int f32 = 0;                      // this GP's M x N strips are float32 (gp_pdgp_set_precision / gp_pdgp_set_gp_precision)
  int64_t off_theta = 0, off_z = 0, off_qmu = 0, off_qsqrt = 0;
};

struct BwdBufs {  // per-GP backward workspace (device)
  double* H = nullptr;      // M x M   A D A^T
  double* E = nullptr;      // M x M   Lq Lq^T - I
  double* T1 = nullptr;     // M x M   scratch
  double* T2 = nullptr;     // M x M   scratch
  double* Wbar = nullptr;   // M x M
  double* R = nullptr;      // M x M   W^T E
  double* R32 = nullptr;    // float32 strips only: M * M floats, the float32 copy of R (gemm_wave_f32.hip)
  double* G = nullptr;      // M x N   K̄uf (dense part R (A D))
  double* u = nullptr;      // M       A gm
  double* upart = nullptr;  // nsplit x M   fused row-dot partials
  double* Lu = nullptr;     // M       L u
  double* alpha = nullptr;  // M       W^T q_mu
  double* hyp_part = nullptr;   // hyper-gradient partial sums (Kuf side)
  double* hyp_part_uu = nullptr;
  double* gz_part = nullptr;    // z-gradient partials
  double* gvsum = nullptr;      // sum_n gv
  // unwhitened model only: the equivalent whitened variational state q' = (W q_mu, W Lq) and its gradient
  double* qmu_w = nullptr; double* Lq_w = nullptr;       // M, M x M
  double* g_qmu_w = nullptr; double* g_Lq_w = nullptr;   // M, M x M
};

struct gp_pdgp_plan_s {
  gp_handle h = nullptr;
  int P = 0, G = 0, whiten = 1, nlin = 0, maxN = 0;
  int f32 = 0;               // gp_pdgp_set_precision: the M x N strips (Kuf, A, Kuf_bar) of EVERY latent GP are float32 (gemm_f32.hip)
  int n64 = 0;               // latent GPs [0, n64) keep float64 strips, [n64, G) have float32 ones (PdgpGP::f32): G, 0, or —
                             // gp_pdgp_set_gp_precision — in between (activation GPs float64, component GPs float32)
  double jitter = 1e-6;
  std::vector<PdgpGP> gps;
  int64_t nparams = 0;
  int maxM = 0, maxm = 0;
  // workspace
  void* ws = nullptr; size_t ws_bytes = 0;
  CondBatch cb;
  std::vector<BwdBufs> bw;
  std::vector<double*> tr_part;    // unwhitened KL: column-sum partials of (W Lq)^2
  double* fmean = nullptr; double* fvar = nullptr;   // [G][maxN]
  double* gFmu = nullptr; double* gFvar = nullptr;   // [G][maxN]
  double* kl = nullptr;                              // [G]
  double* lik_partials = nullptr;                    // [2 * blocks]
  double* slabs = nullptr;                           // split-K slabs
  char* d_misc = nullptr; size_t misc_bytes = 0;     // KL items + backward problem arrays
  std::vector<char> h_misc;
  size_t off_kl_items = 0;
  size_t off_bwd[24] = {0};
  size_t off_fin_items = 0; std::vector<char> h_fin_items;   // batched hyper-gradient finish (bwd.hip)
  // Kuf-side and Kuu-side contractions grouped by kernel family: one launch per family and side over an item array
  // (bwd.hip; G Kuf-side items, then G Kuu-side items, same order)
  size_t off_hy_items = 0;
  struct HyFamily { int type = 0, m = 0, first = 0, count = 0, M = 0, mfma = 0, f32 = 0; bool batched = false; std::vector<int> gps; };
  std::vector<HyFamily> hy_fams;
  size_t off_kl2 = 0;         // unwhitened backward: KL items of the equivalent whitened state
  double* qw_block = nullptr; size_t qw_doubles = 0;   // [q' | grad q'] of all GPs, contiguous (one memset)
  double* kl_dummy = nullptr;
  int nsplit = 1;
  int nK = 0;                 // number of GPs whose kernel gradients are needed (compacted batch)
  std::vector<int> kgps;      // their indices
  // cache keys for the descriptor upload
  const double* last_params = nullptr; const double* last_x = nullptr; double* last_grad = nullptr; int last_n = -1;
  bool bwd_carved = false;
  GemmProblem dummy_prob;      // sink for descriptor slots a GP does not need (pdgp_upload_bwd)
  // the ELBO's final reduction (and the noise-variance gradient it produces), handed to the backward pass: neither is
  // needed before the step ends, so it runs at the head of the helper stream's chain instead of between the last forward
  // product and Kuf_bar (bwd.hip: pdgp_backward)
  struct { const double* lik_partials = nullptr; int nb = 0; const double* kl = nullptr; int nkl = 0; double* elbo = nullptr;
           double* g_noise = nullptr; bool pending = false; } fin;
  int overlap = 2;             // gp_pdgp_set_overlap: 0 one stream, 1 Kuu factorisation / Kuu-side backward on the helper
                               // stream, 2 also the H = A D A^T chain next to Kuf_bar
  bool era_ready = false;      // pdgp_prefetch_backward ran for the current evaluation
  bool factor_valid = false;   // L / W hold the factorisation of the parameters last passed to gp_pdgp_predict
  // two-stage (pitch-sharded) evaluation: what gp_pdgp_elbo_begin staged for gp_pdgp_elbo_end
  int staged_n = 0; double* staged_grad = nullptr; const double* staged_params = nullptr;
  // GP-sharded plan (gp_pdgp_create_subset): this plan's G latent GPs are rows `grow[g]` of the whole model's 2 P latent
  // GPs (engine order [g_0..g_{P-1}, f_0..f_{P-1}]); P is the WHOLE model's source count (the likelihood's)
  bool subset = false;
  std::vector<int> grow;
  double* gF_full_mu = nullptr; double* gF_full_var = nullptr;   // [2 P][maxN]: d varexp / d fmean, fvar of every latent GP
};

