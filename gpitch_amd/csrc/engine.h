// engine.h — batched sparse-GP conditional engine shared by the Pdgp / SGPR plans and the one-shot
// C-ABI operators.  Host-side orchestration only; all arithmetic is in the .hip kernels.
#pragma once
#include "common.h"

// lik.hip helpers
size_t cond_finish_item_bytes();
void cond_finish_fill(void* host_item, const double* s1, int rb1, const double* s2, int rb2, const double* dot,
                      int rbdot, DevKern k, double* fmean, double* fvar);
gp_status launch_cond_finish(gp_handle h, const void* d_items, int count, int N);
size_t kl_item_bytes();
void kl_item_fill(void* host_item, const double* q_mu, const double* q_sqrt, int M, double* out, double* g_mu,
                  double* g_sqrt);
gp_status launch_kl_white(gp_handle h, const void* d_items, int count);
size_t klu_item_bytes();
void klu_item_fill(void* host_item, const double* q_mu, const double* q_sqrt, const double* L, const double* W,
                   const double* tr_part, int nrb, int M, double* out);
gp_status launch_kl_unwhite(gp_handle h, const void* d_items, int count);
gp_status launch_elbo_finish(gp_handle h, const double* lik_partials, int nblocks, const double* kl, int nkl,
                             double* elbo, double* g_noise);
gp_status launch_mean_source(gp_handle h, const double* fmean, int P, int n, int nlin, double* out);

// bwd.hip helpers
gp_status launch_rowdot_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM);
gp_status launch_sub_identity_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM);
gp_status launch_phi_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM);
gp_status launch_rank1_tril_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM);
gp_status launch_matvec_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM, int trans);
gp_status launch_addvec_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM);
gp_status launch_tril_add_batched(gp_handle h, const GemmProblem* d_probs, int batch, int maxM);
gp_status launch_hyper_contract(gp_handle h, DevKern k, const double* x1, int n1, const double* x2, int n2,
                                const double* G, int64_t ldg, const double* alpha, const double* gm, int symmetric,
                                const double* feat, double* partials, int* nparts, double* gz_partials,
                                const double* kvals = nullptr, int64_t ldk = 0, int g32 = 0);   // g32: G (and kvals) are float32 strips
gp_status launch_hyper_finish(gp_handle h, DevKern k, const double* partials, int nparts, const double* gv_sum,
                              double* g_theta, const double* gz_partials, int ncolblocks, int n1, double* g_z);
int hyper_num_sums(int m);
// item forms (one launch for many contractions / finishes: the window-batched SGPR plan)
struct HyperItem {
  DevKern k;
  const double* x1; const double* x2; const double* G; const double* alpha; const double* gm;
  const double* f1; const double* f2; double* partials; double* gz;
  const double* kvals;            // the covariance strip itself (Mercer Kuf side, matrix-core form), or null
  int64_t ldg, ldk; int n1, n2, symmetric, g32;
};
struct HyperFinishItem {
  DevKern k;
  const double* p_uf; const double* p_uu; const double* gv_sum;
  double* g_theta; const double* gz_uf; const double* gz_uu; double* g_z;
  int np_uf, np_uu, cb_uf, cb_uu, n1, pad;
};
// use_mfma: Mercer family, every item has kvals and no inducing-input gradient -> hyper_sm_rows_kernel over the items
// (g32_items: the items' strips are float32 — all of them, as their g32 fields say)
gp_status launch_hyper_contract_items(gp_handle h, int type, int m, const HyperItem* d_items, int count, int n1, int n2,
                                      int with_gz, int* nparts, int use_mfma = 0, const double* x2_shared = nullptr,
                                      int g32_items = 0, int lean_items = 0);
// one pass over G for the P MercerMatern12sm kernels of a sum (bwd.hip; true = taken)
bool launch_hyper_contract_sum(gp_handle h, const DevKern* kernels, double* const* feats, double* const* partials, int P,
                               const double* x1, int n1, const double* x2, int n2, const double* G, int64_t ldg,
                               const double* alpha, const double* gm, int g32, int* nparts, gp_status* st);
gp_status launch_hyper_finish_items(gp_handle h, const HyperFinishItem* d_items, int count, int maxblocks);
size_t hyper_finish_item_bytes();
// partial records the Kuf-side contraction of an M x N strip may write (the largest over its kernel variants)
size_t hyper_kuf_records(int N, int M);

// One latent GP inside a batch of conditionals.
struct CondTask {
  DevKern kern;
  const double* z = nullptr; int M = 0;
  const double* q_mu = nullptr; const double* q_sqrt = nullptr;  // q_sqrt may be null (no variational covariance)
  // workspace (device)
  double* L = nullptr;    // M x M  Kuu -> chol
  double* W = nullptr;    // M x M  L^-1
  double* Tblk = nullptr; // 128 x M scratch (blocked inverse)
  double* Kuf = nullptr;  // M x N
  double* A = nullptr;    // M x N  W Kuf
  double* A2 = nullptr;   // M x N  W^T A (unwhitened only)
  double* W32 = nullptr;  // float32 strips only: M * M floats each, the float32 copies of W and tril(Lq)^T (gemm_wave_f32.hip)
  double* Lq32 = nullptr;
  double* feat = nullptr; // spectral-mixture features (2m x (M + N))
  double* feat_uu = nullptr; // the same for the Kuu build (2m x M), separate because the two builds overlap
  double* s1 = nullptr; double* s2 = nullptr; double* dot = nullptr;  // [rowblocks][N] partials
  double* fmean = nullptr; double* fvar = nullptr;                    // N each
  bool f32 = false;       // this GP's M x N strips (Kuf, A, A2) are float32 (per-GP precision: CondBatch::n64)
};

struct CondBatch {
  std::vector<CondTask> tasks;
  int N = 0;
  int maxM = 0;
  // device descriptor storage (inside the caller's workspace)
  char* d_desc = nullptr; size_t desc_bytes = 0;
  std::vector<char> h_desc;
  // offsets of the descriptor arrays inside d_desc
  size_t off_chol_ptrs = 0, off_w_ptrs = 0, off_Ms = 0, off_lds = 0, off_f1 = 0, off_f1u = 0, off_f2 = 0, off_finish = 0;
  bool uploaded = false;
  bool overlap = true;    // run the Kuu factorisation on the handle's helper stream next to the Kuf builds
  bool f32 = false;       // every task's M x N strips (Kuf, A) are float32 and the strip products run on the float32 matrix path
  // per-GP precision: tasks [0, n64) keep float64 strips, tasks [n64, G) have float32 ones (CondTask::f32); -1 = uniform
  // (all float64 or, with f32 set, all float32).  cond_batch_upload checks the order and sets it.
  int n64 = -1;
  bool wave_a = false, wave_lta = false;
  bool wave_a32 = false, wave_lta32 = false;   // the same for the float32 tasks (gemm_wave_f32.hip)   // A = W Kuf / Lq^T A of the float64 tasks take gemm_wave.hip's form (64-row partial rows)
  // grouped covariance builds (one launch per kernel family)
  struct Group { int type = 0, m = 0, first = 0, maxM = 0; bool f32 = false; std::vector<int> members; };
  std::vector<Group> groups;
  size_t off_cov_uu = 0, off_cov_uf = 0, off_feat_zuu = 0, off_feat_zuf = 0, off_feat_x = 0;
  // blocked Kuu factorisation (engine.hip: cond_batch_factorize)
  bool blocked = false; int nblk = 0;
  bool diag_ready = false;   // set by a factorisation that recorded gp_handle_s::ev_diag after the diagonal blocks of W
  size_t off_blk_mats[8] = {0}, off_blk_w[8] = {0}, off_blk_M[8] = {0}, off_blk_gemm[8][4] = {{0}};
  size_t off_diag_mats = 0, off_diag_w = 0, off_diag_M = 0, off_diag_ld = 0;   // all panels' diagonal blocks, one batch
};

int cond_batch_uniform(const CondBatch& cb, int N);
size_t cond_task_workspace_doubles(int M, int N, int num_partials, bool whiten, bool f32 = false);
size_t cond_batch_desc_bytes(int count);
// carve the per-task buffers out of the arena
bool cond_task_carve(GpArena& ar, CondTask& t, int N, bool whiten, bool f32 = false);
gp_status cond_batch_upload(gp_handle h, CondBatch& cb, bool whiten, double jitter);
// run: Kuu -> chol -> W ; Kuf ; A = W Kuf ; (A2 = W^T A) ; Lq^T A ; reductions -> fmean, fvar
gp_status cond_batch_run(gp_handle h, CondBatch& cb, const double* x, int N, bool whiten, double jitter,
                         bool reuse_factor = false);
