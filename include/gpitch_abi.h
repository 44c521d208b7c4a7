/*
 * gpitch_abi.h — C-ABI of libgpitch_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the gpitch `pdgp` / `sgpr_ss` ELBO path.  The reference
 * (PabloAlvarado/gpitch) has NO FFI boundary of its own: the path is Python objects over a
 * TensorFlow graph.  Each entry point below therefore replaces the TF/GPflow op sequence behind one
 * reference *operator* call; the reference interface it replaces is cited as file:line relative to
 * the reference tree.  The ctypes binding a gpitch maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; all array arguments are DEVICE pointers to float64 unless the
 *    name ends in `_host`; matrices are row-major with explicit leading dimension where given;
 *  - the caller allocates and owns every buffer (including workspaces); the library owns only the
 *    opaque handle / plan objects;
 *  - every call returns a gp_status; nothing throws across the boundary; calls enqueue work on the
 *    handle's HIP stream and return immediately unless they produce a host scalar;
 *  - a handle is bound to one (process, GPU) and is not thread-safe.
 *  - there is NO CPU fallback: without a gfx950 device every compute entry point fails with
 *    GP_ERR_HIP / GP_ERR_NO_DEVICE.
 */
#ifndef GPITCH_ABI_H
#define GPITCH_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPITCH_ABI_VERSION 1

typedef int32_t gp_status;
enum {
  GP_OK = 0,
  GP_ERR_BAD_ARG = 1,
  GP_ERR_NOT_PD = 2,     /* Cholesky hit a non-positive pivot (TF InvalidArgumentError in the reference) */
  GP_ERR_HIP = 3,        /* a HIP runtime call failed; gp_last_error() has the text */
  GP_ERR_NO_DEVICE = 4,
  GP_ERR_WORKSPACE = 5,  /* caller-provided workspace too small */
  GP_ERR_UNSUPPORTED = 6
};

/* kernel (covariance function) types — the GPflow operator API `Kern.K(X, X2)`, `Kern.Kdiag(X)` */
enum {
  GP_KERN_MATERN12 = 0,          /* gpflow.kernels.Matern12  (init_models.py:83)                 */
  GP_KERN_MATERN32 = 1,          /* gpflow.kernels.Matern32  (init_kernels.py:12, demo-modgp.py:32) */
  GP_KERN_MATERN52 = 2,          /* gpflow.kernels.Matern52  (init_models.py:188)                */
  GP_KERN_RBF = 3,               /* gpflow.kernels.RBF                                            */
  GP_KERN_MERCER_MATERN12SM = 4, /* gpitch/matern12_spectral_mixture.py:70-133                    */
  GP_KERN_MATERN12SM = 5,        /* gpitch/matern12_spectral_mixture.py:14-67                     */
  GP_KERN_MATERN32SM = 6,        /* gpitch/kernels.py:204-258 (init_models.py:84,96): broadcast form, Matern-3/2
                                  * envelope; theta = [1 (unused, fixed), lengthscales, variance_k.., frequency_k..] */
  GP_KERN_MERCER_MATERN52SM = 7  /* Matern52 * MercerCosMix product (init_models.py:183-198, kernels.py:321-376):
                                  * theta = [v52 * v_cosmix, lengthscales, energy_k.., frequency_k..];
                                  * Kdiag = variance (MercerCosMix.Kdiag fills its variance, no energy sum) */
};

/* nonlinearities of the modulated likelihood — gpitch/methods.py:216-233 */
enum { GP_NLIN_LOGISTIC = 0, GP_NLIN_SOFTPLUS = 1, GP_NLIN_GAUSS = 2 };

/* Kernel descriptor.  `theta` is a DEVICE pointer to the constrained hyper-parameters
 *   [variance, lengthscales, energy_0..energy_{m-1}, frequency_0..frequency_{m-1}]  (m = num_partials,
 *   0 for the plain stationary kernels) — device-resident so an optimiser step never syncs the host. */
typedef struct {
  int32_t type;
  int32_t num_partials;
  const double* theta;
} gp_kernel_desc;

#define GP_THETA_LEN(m) (2 + 2 * (m))

typedef struct gp_handle_s* gp_handle;
typedef struct gp_pdgp_plan_s* gp_pdgp_plan;
typedef struct gp_sgpr_plan_s* gp_sgpr_plan;
typedef struct gp_sgprb_plan_s* gp_sgprb_plan;

/* ---- runtime -------------------------------------------------------------------------------- */
/* replaces gpitch.init_settings / the global TF session (gpitch/methods.py:155-180).
 * `stream` is the hipStream_t every call enqueues on; NULL = HIP's default (null) stream. */
gp_status gp_create(int32_t device_id, void* stream, gp_handle* out);
gp_status gp_destroy(gp_handle h);
gp_status gp_sync(gp_handle h);
const char* gp_last_error(gp_handle h);       /* text of the last failure on this handle */
int32_t gp_abi_version(void);
/* pivot index reported by the last GP_ERR_NOT_PD (−1 if none) */
int32_t gp_last_not_pd_index(gp_handle h);

/* ---- L2 operators: Kern.K / Kern.Kdiag --------------------------------------------------------
 * replaces MercerMatern12sm.K (matern12_spectral_mixture.py:102-117), Matern12sm.K (:38-56) and GPflow
 * Stationary.K for Matern12/32/52/RBF.  out[i*ld + j] = k(x1[i], x2[j]);  x2 == NULL means K(X) (x2 = x1).
 * `accumulate` != 0 adds into `out` (GPflow `Add` kernel: sgpr_ss.py:42-43). */
gp_status gp_kernel_build(gp_handle h, const gp_kernel_desc* kern, const double* x1, int32_t n1,
                          const double* x2, int32_t n2, double* out, int64_t ld, int32_t accumulate);
/* The same build with a float32 result (ld in floats; inputs and arithmetic float64, one rounding at the store):
 * the Kuf strips of the configurations the reference would run with float_type = float32
 * (gpitch/matern12_spectral_mixture.py:8-11 np_float_type). */
gp_status gp_kernel_build_f32(gp_handle h, const gp_kernel_desc* kern, const double* x1, int32_t n1,
                              const double* x2, int32_t n2, float* out, int64_t ld, int32_t accumulate);
/* Kern.Kdiag(X): exact fill (matern12_spectral_mixture.py:58-62,119-121) */
gp_status gp_kernel_diag(gp_handle h, const gp_kernel_desc* kern, int32_t n, double* out, int32_t accumulate);

/* ---- GPflow conditional pieces ------------------------------------------------------------------
 * Kmm = K(z) + jitter I ; Lm = chol(Kmm) ; Linv = Lm^-1   (GPflow conditional, from pdgp.py:147;
 * sgpr_ss.py:43-44).  L and Linv are M x M row-major (ld = M), strictly-upper part zeroed.
 * Either output may be NULL.  workspace: gp_chol_workspace_bytes(M). */
size_t gp_chol_workspace_bytes(int32_t M);
gp_status gp_kuu_cholesky(gp_handle h, const gp_kernel_desc* kern, const double* z, int32_t M, double jitter,
                          double* L, double* Linv, void* workspace, size_t workspace_bytes);
/* dense lower Cholesky of a caller-supplied SPD matrix (tf.cholesky: sgpr_ss.py:44,51,89), in place */
gp_status gp_cholesky_inplace(gp_handle h, double* A, int32_t M, int64_t ld);

/* gpflow.conditionals.conditional(Xnew, z, kern, f=q_mu, full_cov=False, q_sqrt, whiten)
 * (call sites pdgp.py:147-155,176-178,185-187,199-205).  q_sqrt is the M x M (x1) matrix; its lower
 * triangle is used (matrix_band_part).  fmean, fvar: N values each.  q_sqrt may be NULL.
 * workspace: gp_conditional_workspace_bytes(N, M). */
size_t gp_conditional_workspace_bytes(int32_t N, int32_t M);
gp_status gp_conditional_diag(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N,
                              const double* z, int32_t M, const double* q_mu, const double* q_sqrt,
                              int32_t whiten, double jitter, double* fmean, double* fvar,
                              void* workspace, size_t workspace_bytes);

/* float32 form of the whitened conditional (the reference's settings.dtypes.float_type = float32, pdgp.py:13):
 * Kuf and A = Lm^-1 Kuf are float32 strips, both strip products run on v_mfma_f32_16x16x4_f32; Kmm, Lm, Lm^-1 and all
 * reductions (sum A^2, A^T q_mu, sum LTA^2) are float64, as are the inputs and outputs.  Same workspace size bound. */
gp_status gp_conditional_diag_f32(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N,
                                  const double* z, int32_t M, const double* q_mu, const double* q_sqrt,
                                  double jitter, double* fmean, double* fvar,
                                  void* workspace, size_t workspace_bytes);
/* ... with the `whiten` argument of conditional(): in the reference `whiten` and `float_type` are independent settings
 * (gpitch/pdgp.py:13,49,122-129).  whiten = 0 adds A' = Lm^-T A as a third float32 strip product; its rounding is
 * amplified by cond(Kuu) instead of cond(Kuu)^(1/2) (tolerances: tests/test_gpu_f32.py). */
gp_status gp_conditional_diag_f32w(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N,
                                   const double* z, int32_t M, const double* q_mu, const double* q_sqrt,
                                   int32_t whiten, double jitter, double* fmean, double* fvar,
                                   void* workspace, size_t workspace_bytes);
/* full_cov = True of the same operator (GPflow 0.5 conditionals.conditional; gpitch/pdgp.py never asks for it): fmean (N)
 * and the N x N posterior covariance fcov (row-major, ld N) = K(xnew) - A^T A + (Lq^T A')^T (Lq^T A').
 * workspace: gp_conditional_full_workspace_bytes(N, M). */
size_t gp_conditional_full_workspace_bytes(int32_t N, int32_t M);
gp_status gp_conditional_full(gp_handle h, const gp_kernel_desc* kern, const double* xnew, int32_t N, const double* z,
                              int32_t M, const double* q_mu, const double* q_sqrt, int32_t whiten, double jitter,
                              double* fmean, double* fcov, void* workspace, size_t workspace_bytes);

/* gpflow.kullback_leiblers.gauss_kl(q_mu, q_sqrt, K=None) (pdgp.py:120-121 whitened; :126-129 with
 * K = kern.K(z) + jitter I built internally when kern != NULL).  Result to *out_host (syncs).
 * workspace: gp_gauss_kl_workspace_bytes(M, kern != NULL). */
size_t gp_gauss_kl_workspace_bytes(int32_t M, int32_t with_kernel);
gp_status gp_gauss_kl(gp_handle h, const double* q_mu, const double* q_sqrt, int32_t M,
                      const gp_kernel_desc* kern_or_null, const double* z, double jitter,
                      double* out_host, void* workspace, size_t workspace_bytes);
/* the same with the caller's own prior covariance: gauss_kl(q_mu, q_sqrt, K) for a device matrix K (M x M,
 * row-major, ld = M, symmetric positive definite; not modified).  GP_ERR_NOT_PD when its Cholesky fails. */
gp_status gp_gauss_kl_matrix(gp_handle h, const double* q_mu, const double* q_sqrt, int32_t M, const double* K,
                             double* out_host, void* workspace, size_t workspace_bytes);

/* MpdLik.variational_expectations(Fmu, Fvar, Y) (likelihoods.py:422-447 + hermgauss1d :33-45 +
 * log_lik_exp :47-68).  Fmu/Fvar are N x 2P row-major with columns [g_0..g_{P-1}, f_0..f_{P-1}]
 * (pdgp.py:157-164).  noise_var is a device scalar.  per_frame (N values) may be NULL; the sum over
 * frames is written to *sum_host when non-NULL (syncs). */
gp_status gp_mpd_varexp(gp_handle h, const double* Fmu, const double* Fvar, const double* y, int32_t N,
                        int32_t P, int32_t nlin, const double* noise_var, double* per_frame, double* sum_host);

/* ---- L3 model: Pdgp (gpitch/pdgp.py:48-208) ---------------------------------------------------
 * Parameter vector layout (float64, constrained space), offsets returned by gp_pdgp_layout:
 *   [ noise_var | for g in (act_0..act_{P-1}, com_0..com_{P-1}):
 *                   theta_g (2+2m_g) | z_g (M_g) | q_mu_g (M_g) | q_sqrt_g (M_g x M_g row-major) ]
 * The same layout is used for the gradient vector, the free-state vector and the Adam moments. */
typedef struct {
  int32_t num_sources;            /* P */
  int32_t whiten;                 /* pdgp.py:49  */
  int32_t nlin;                   /* GP_NLIN_* (pdgp.py:49 nlinfun) */
  int32_t max_batch;              /* largest minibatch N the plan will see */
  const int32_t* M_act;           /* host, P entries (pdgp.py:93) */
  const int32_t* M_com;           /* host, P entries (pdgp.py:94) */
  const int32_t* kern_type_act;   /* host, P entries GP_KERN_* */
  const int32_t* kern_type_com;
  const int32_t* partials_act;    /* host, P entries (0 for plain stationary) */
  const int32_t* partials_com;
  double jitter;                  /* settings.numerics.jitter_level = 1e-6 (pdgp.py:14) */
} gp_pdgp_config;

gp_status gp_pdgp_create(gp_handle h, const gp_pdgp_config* cfg, gp_pdgp_plan* out);
gp_status gp_pdgp_destroy(gp_pdgp_plan p);
int64_t gp_pdgp_num_params(gp_pdgp_plan p);
/* offsets into the parameter vector for GP index g in [0, 2P): g < P activation i=g, else component */
gp_status gp_pdgp_layout(gp_pdgp_plan p, int32_t g, int64_t* off_theta, int64_t* off_z, int64_t* off_qmu,
                         int64_t* off_qsqrt);
/* Arithmetic type of the O(M^2 N) part — the reference's `float_type` setting (pdgp.py:13,
 * matern12_spectral_mixture.py:8-11; BASELINE configs 3 and 5 are quoted at fp32).  bits = 64 (default) or 32:
 * with 32 the M x N strips Kuf, A = Lm^-1 Kuf and Kuf_bar are stored in float32 and the four strip products
 * (A = W Kuf, Lq^T A, R (A D), A D A^T) run on v_mfma_f32_16x16x4_f32; parameters, Kuu, its Cholesky factor and inverse,
 * every reduction over the inducing index or the frames, the likelihood, the KL terms and the gradient vector stay
 * float64, so the ABI's buffers do not change type.  Whitened models only.  Call before gp_pdgp_workspace_bytes /
 * gp_pdgp_set_workspace (the workspace is smaller).  Tolerance held against the float64 oracle: tests/test_gpu_f32.py. */
gp_status gp_pdgp_set_precision(gp_pdgp_plan p, int32_t bits);
/* The same per latent GP: bits[g] in {32, 64} for the plan's `count` latent GPs in engine order ([g_0..g_{P-1}, f_0..f_{P-1}],
 * a subset plan: its own rows in that order).  The reference's float_type is one global setting (pdgp.py:13); this is the
 * finer form the transcription model wants: its activation GPs (Matern-3/2 over a 16-kHz grid, cond(Kuu) ~ 1e9) are where
 * float32 strips cost accuracy (posterior mean 5e-2, lengthscale gradient 2e-1 of their scale), its component GPs are not,
 * so bits = [64 x P, 32 x P] keeps the float64 tolerances on the former and the float32 speed on the latter.  Float64
 * latent GPs must precede float32 ones (GP_ERR_UNSUPPORTED otherwise).  Call before gp_pdgp_workspace_bytes. */
gp_status gp_pdgp_set_gp_precision(gp_pdgp_plan p, const int32_t* bits, int32_t count);
size_t gp_pdgp_workspace_bytes(gp_pdgp_plan p);
gp_status gp_pdgp_set_workspace(gp_pdgp_plan p, void* workspace, size_t bytes);

/* Which gradients the caller will use for latent GP g (0 = skip that work).  GPflow's `.fixed = True`
 * (demo-modgp.py:40-41: m.za.fixed / m.zc.fixed; init_models.py:97-98: kern.fixed) removes a Param from the
 * TF gradient; here it lets the backward pass skip the inducing-input contraction (need_z = 0) or, when the
 * kernel hyper-parameters are fixed as well (need_theta = 0), the whole Kuf_bar / Kuu_bar chain of that GP.
 * Skipped entries of the gradient vector are left at zero.  Default: everything needed. */
gp_status gp_pdgp_set_grad_needs(gp_pdgp_plan p, int32_t g, int32_t need_theta, int32_t need_z);

/* How much of a step may run on the handle's internal helper stream (same results, different schedule):
 *   0  everything on the handle's stream, in order;
 *   1  forward: Kuu builds + Cholesky / inverse next to the Kuf builds; backward: the Kuu-side chain (Cholesky
 *      adjoint and its contraction) underneath the Kuf_bar product;
 *   2  (default) additionally the H = A diag(2 gv) A^T chain next to Kuf_bar — the two largest backward products then
 *      share the device, so each one's own launch takes longer while the step gets shorter.
 * Batches below 4096 frames always run on one stream. */
gp_status gp_pdgp_set_overlap(gp_pdgp_plan p, int32_t level);

/* Pdgp.build_likelihood (pdgp.py:133-170) on the batch (x, y) of n frames:
 *   elbo = (num_data / n) * sum_n varexp_n - KL.   elbo_dev points to TWO device doubles: [0] the ELBO, [1] the
 * summed KL term (Pdgp.build_prior_kl, pdgp.py:113-131).  When
 * elbo_host != NULL the ELBO is also copied to the host (syncs).  When grad != NULL also writes d elbo / d params
 * (constrained space, layout above) — what TF reverse-mode provides through Model.optimize. */
gp_status gp_pdgp_elbo(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                       double num_data, double* elbo_dev, double* elbo_host, double* grad);

/* Pitch-sharded evaluation of ONE Pdgp model over several GPUs (one process per GPU).  Each rank builds a plan
 * over its own subset of the P sources (both GPs of a pitch on the same rank; the noise variance params[0] is
 * replicated).  The likelihood (likelihoods.py:47-68) couples sources only through three per-frame sums
 *   A = sum_i E1_i m_f,i ,  B = sum_i E2_i (v_f,i + m_f,i^2) ,  D = sum_i (E1_i m_f,i)^2     (C = A^2 - D),
 * so the exchange step is ONE sum-all-reduce of 3n+1 doubles (the last slot carries the KL terms):
 *   gp_pdgp_elbo_begin  : conditionals of the local GPs; writes the local [A | B | D | sum KL] to exchange[0..3n]
 *   -- caller: ncclAllReduce(exchange, 3n+1, ncclSum) ordered after begin / before end on the handle's stream --
 *   gp_pdgp_elbo_end    : likelihood + noise gradient from the reduced sums (identical on every rank), local
 *                         backward pass; elbo_dev / elbo_host / grad as in gp_pdgp_elbo (the ELBO is the whole
 *                         model's; grad covers this rank's parameters, and grad[0] is the full noise gradient).
 * No collective is needed in the backward pass.  begin and end must be called with the same params, n and grad. */
gp_status gp_pdgp_elbo_begin(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                             double* grad, double* exchange);
gp_status gp_pdgp_elbo_end(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                           double num_data, const double* exchange, double* elbo_dev, double* elbo_host,
                           double* grad);

/* GP-sharded evaluation of ONE Pdgp model over several GPUs (option 2; one process per GPU).  The 2 P
 * conditionals of Pdgp.build_likelihood are independent (pdgp.py:146-155) and the likelihood needs only (fmean, fvar) of
 * every latent GP (likelihoods.py:422-447), so the latent GPs themselves are the sharding unit: 24 GPs at P = 12 are 3
 * per GPU on 8 GPUs (ceiling 8x; the pitch-sharded form above: 6x).
 *   gp_pdgp_create_subset : a plan over `count` of the model's 2 P latent GPs; gp_index[l] (strictly increasing) is the row
 *                           in the model's order [g_0..g_{P-1}, f_0..f_{P-1}] (pdgp.py:157-164).  `cfg` describes the WHOLE
 *                           model.  The parameter vector holds [noise | the listed GPs] (gp_pdgp_layout by local index).
 *   gp_pdgp_cond_begin    : conditionals of the local GPs -> fmean_local, fvar_local ([count][n], caller's buffers: the
 *                           send buffer of the exchange) and the sum of the local KL terms -> kl_local_sum[0].
 *   -- caller: ONE all-gather of 2 n doubles per latent GP (+ the KL scalar) on the handle's stream --
 *   gp_pdgp_cond_end      : likelihood of the whole model from fmean_full / fvar_full ([2 P][n], model order; computed
 *                           identically on every rank, so the replicated noise variance stays in step) with kl_total[0]
 *                           = the KL sum over all ranks; local backward pass.  elbo_dev / elbo_host / grad as in
 *                           gp_pdgp_elbo (grad covers this plan's parameters; grad[0] is the full noise gradient).
 * No collective in the backward pass.  gp_pdgp_predict on such a plan takes mean_source = NULL. */
gp_status gp_pdgp_create_subset(gp_handle h, const gp_pdgp_config* cfg, const int32_t* gp_index, int32_t count,
                                gp_pdgp_plan* out);
gp_status gp_pdgp_cond_begin(gp_pdgp_plan p, const double* params, const double* x, int32_t n, double* grad,
                             double* fmean_local, double* fvar_local, double* kl_local_sum);
gp_status gp_pdgp_cond_end(gp_pdgp_plan p, const double* params, const double* x, const double* y, int32_t n,
                           double num_data, const double* fmean_full, const double* fvar_full, const double* kl_total,
                           double* elbo_dev, double* elbo_host, double* grad);

/* ---- the sharded forms with their exchange step INSIDE the call (the handle owns "workspace, stream,
 * RCCL comm"; section 8e).  The reference has no distributed code: nothing of it is replaced here beyond what
 * gp_pdgp_elbo_begin/_end, gp_pdgp_cond_begin/_end and gp_sgpr_bound_begin/_end replace (pdgp.py:133-170, sgpr_ss.py:29-71);
 * these entry points issue the collective the "-- caller: --" lines above ask for themselves — ncclAllReduce / ncclAllGather
 * of RCCL on the handle's stream, between the two stages — so that one evaluation, and with `adam` one whole optimiser step,
 * is ONE enqueue with no host code in between.  RCCL is bound at run time (the copy already in the process, else
 * /opt/rocm/lib/librccl.so); without it gp_comm_create returns GP_ERR_UNSUPPORTED and nothing else is affected.
 *   gp_comm_unique_id : rank 0 fills 128 bytes (ncclGetUniqueId) and hands them to every rank by any channel
 *                       (torch.distributed broadcast, a file, MPI ...)
 *   gp_comm_create    : every rank, same id: ncclCommInitRank on the handle's device; the communicator's collectives run on
 *                       the handle's stream.  Plans used with a communicator must live on the same handle.
 * gp_adam_args (may be NULL): gp_adam_step's arguments, applied right behind the evaluation (params is updated in place). */
typedef struct gp_comm_s* gp_comm;
typedef struct {
  double* free_state; const uint8_t* tcode; double* m; double* v;
  int64_t nparams; int64_t t; double lr, beta1, beta2, eps;
} gp_adam_args;
gp_status gp_comm_unique_id(uint8_t* id128);
gp_status gp_comm_create(gp_handle h, const uint8_t* id128, int32_t rank, int32_t world, gp_comm* out);
gp_status gp_comm_destroy(gp_comm c);
int32_t gp_comm_world(gp_comm c);
int32_t gp_comm_rank(gp_comm c);
gp_status gp_comm_allreduce_sum(gp_comm c, double* buf, int64_t count);      /* in place, float64, on the handle's stream */
/* pitch-sharded: gp_pdgp_elbo_begin -> all-reduce of exchange[0 .. 3n] -> gp_pdgp_elbo_end [-> gp_adam_step] */
gp_status gp_pdgp_elbo_pitch_sharded(gp_pdgp_plan p, gp_comm c, double* params, const double* x, const double* y, int32_t n,
                                     double num_data, double* exchange, double* elbo_dev, double* elbo_host, double* grad,
                                     const gp_adam_args* adam);
/* GP-sharded: gp_pdgp_cond_begin -> all-gather of one block per rank -> rows assembled in the model's order ->
 * gp_pdgp_cond_end [-> gp_adam_step].  num_gps = 2 P; local_gps = this plan's count; latent GP g lives on rank g mod world
 * as its (g div world)-th row.  With per = ceil(num_gps / world): send holds 2 per n + 8 doubles
 * [fmean rows | fvar rows | KL sum + pad], recv world times that, full 2 num_gps n + 8 ([fmean | fvar | KL total]). */
gp_status gp_pdgp_elbo_gp_sharded(gp_pdgp_plan p, gp_comm c, double* params, const double* x, const double* y, int32_t n,
                                  double num_data, int32_t num_gps, int32_t local_gps, double* send, double* recv, double* full,
                                  double* elbo_dev, double* elbo_host, double* grad, const gp_adam_args* adam);
/* frame-sharded SGPRSS: gp_sgpr_bound_begin -> all-reduce(exchange) -> gp_sgpr_bound_end (the frame-independent terms on
 * rank 0) [-> all-reduce(grad)]: bound and, when grad != NULL, the full gradient on every rank. */
gp_status gp_sgpr_bound_grad_sharded(gp_sgpr_plan p, gp_comm c, const double* params, const double* X, const double* Y, int32_t N,
                                     int64_t N_total, const double* Z, double* exchange, double* bound_dev, double* bound_host,
                                     double* grad);

/* Pdgp.predict_act / predict_com / predict_act_n_com (pdgp.py:172-208): conditionals at xnew for all 2P
 * GPs.  fmean/fvar: 2P x n row-major (row g as in gp_pdgp_layout).  mean_source (P x n, may be NULL)
 * = nlinfun(mean_act_i) * mean_com_i (pdgp.py:207). */
gp_status gp_pdgp_predict(gp_pdgp_plan p, const double* params, const double* xnew, int32_t n,
                          double* fmean, double* fvar, double* mean_source);
/* The same at further inputs WITHOUT rebuilding Kuu and its Cholesky factor / inverse: valid only when params hold
 * the values of the preceding gp_pdgp_predict[_reuse] call on this plan (no ELBO evaluation or parameter update in
 * between).  pdgp.py:17-44 (predict_windowed) re-factorises for every window and for act / com separately. */
gp_status gp_pdgp_predict_reuse(gp_pdgp_plan p, const double* params, const double* xnew, int32_t n,
                                double* fmean, double* fvar, double* mean_source);

/* ---- overlap-add of per-window predictions (gpitch/window_overlap.py:19-59: merged_mean / merged_variance) ----------
 * windows: num_windows x ws (row-major, leading dimension ld) device array of per-window means (square = 0) or
 * variances (square = 1: squared Hann weights); ws odd, 50 % overlap (hop (ws-1)/2); out: n = (ws-1)/2 *
 * (num_windows + 1) + 1 frames.  First / last half-windows are kept flat exactly as the reference does. */
gp_status gp_overlap_merge(gp_handle h, const double* windows, int32_t num_windows, int32_t ws, int64_t ld, int32_t n,
                           int32_t square, double* out);

/* ---- optimiser on the free state (GPflow Model.optimize with tf.train.AdamOptimizer;
 *      demo-modgp.py:44-45; transforms: GPflow Log1pe 'positive') -----------------------------------
 * tcode[i]: 0 identity, 1 positive (y = log(1+e^x) + 1e-6), 2 fixed (identity, no update),
 *           3.. gpflow.transforms.Logistic(a, b): y = a + (b-a)/(1+e^-x), with (a, b) registered on the handle
 *           (kernels.py:219-223 Logistic(0,2) / Logistic(0,0.25); init_models.py:189 Logistic(0,0.5)). */
gp_status gp_transform_register_logistic(gp_handle h, double a, double b, uint8_t* code_out);
gp_status gp_transform_forward(gp_handle h, const double* free_state, const uint8_t* tcode, int64_t n,
                               double* params);
gp_status gp_transform_backward(gp_handle h, const double* params, const uint8_t* tcode, int64_t n,
                                double* free_state);
/* One Adam step maximising the ELBO: g_free = -(grad * dparams/dfree); TF-1.2 update rule; then
 * params = transform(free).  t is the 1-based step count. */
/* A failed Cholesky inside an asynchronous evaluation (sync-free training loops) sets a device flag; while it is set
 * gp_adam_step leaves the free state, the parameters and the moments untouched, so a training loop cannot walk on from
 * garbage gradients (the reference's TF session raises at the offending step).  gp_poll_not_pd looks at the flag without
 * blocking (pinned copy + event): *flag = 1 once a failure has been observed; gp_check_not_pd (drains the streams; or any
 * call that returns a host scalar) then gives GP_ERR_NOT_PD with the pivot index and clears the flag. */
gp_status gp_poll_not_pd(gp_handle h, int32_t* flag);
gp_status gp_check_not_pd(gp_handle h);
/* gp_take_not_pd: asynchronous per-evaluation snapshot — behind everything enqueued so far, copies the device status
 * word {flag, pivot index, matrix index (= window slot in a gp_sgprb_* batch), spare} into host_status4 (pinned host
 * memory, read it after an event recorded behind this call) and clears the device word, so a failure is reported by the
 * evaluation that caused it and does not leak into the next user of the handle.  Replaces what the reference gets from
 * TF raising InvalidArgumentError inside the one session.run that failed (transcription.py:283 per window). */
gp_status gp_take_not_pd(gp_handle h, int32_t* host_status4);
gp_status gp_adam_step(gp_handle h, double* free_state, double* params, const double* grad,
                       const uint8_t* tcode, double* m, double* v, int64_t n, int64_t t, double lr,
                       double beta1, double beta2, double eps);

/* ---- L3 model: SGPRSS (gpitch/sgpr_ss.py:10-114) ---------------------------------------------- */
typedef struct {
  int32_t num_kernels;          /* P kernels in the GPflow Add (kern.kern_list) */
  int32_t max_N;
  int32_t M;
  const int32_t* kern_type;     /* host, P entries */
  const int32_t* partials;      /* host, P entries */
  double jitter;
  int32_t reg;                  /* sgpr_ss.py:64-68: subtract 1000 * sum |variance_p| */
} gp_sgpr_config;
/* parameter vector: [ noise_var | theta_0 | ... | theta_{P-1} ]  (Z is a DataHolder: sgpr_ss.py:26) */
gp_status gp_sgpr_create(gp_handle h, const gp_sgpr_config* cfg, gp_sgpr_plan* out);
gp_status gp_sgpr_destroy(gp_sgpr_plan p);
int64_t gp_sgpr_num_params(gp_sgpr_plan p);
/* as gp_pdgp_set_precision: float32 strips (Kuf, A = L^-1 Kuf, Kuf_bar) and float32 matrix-core strip products for
 * the collapsed bound, its gradient and predict_f; Kuu, B = A A^T + I (accumulated in float64), both Cholesky factors
 * and the bound's scalars stay float64.  predict_s (exact N x N posterior) is float64 regardless. */
gp_status gp_sgpr_set_precision(gp_sgpr_plan p, int32_t bits);
size_t gp_sgpr_workspace_bytes(gp_sgpr_plan p);
gp_status gp_sgpr_set_workspace(gp_sgpr_plan p, void* workspace, size_t bytes);
/* SGPRSS.build_likelihood (sgpr_ss.py:29-71), D = 1 output column.  bound_host may be NULL. */
gp_status gp_sgpr_bound(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                        const double* Z, double* bound_dev, double* bound_host);
/* The bound and its gradient w.r.t. the parameter vector (what TF autodiff gives L-BFGS-B in SGPRSS.optimize:
 * transcription.py:283, separation.py:298).  grad has gp_sgpr_num_params entries. */
gp_status gp_sgpr_bound_grad(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                             const double* Z, double* bound_dev, double* bound_host, double* grad);
/* d bound / d err_n, err = Y - mean_function(X) (sgpr_ss.py:40), of the evaluation gp_sgpr_bound_grad has JUST run with the same
 * params, Y and N: r = A'^T (dF/du) - err / sigma^2, N doubles on the device.  This is what a GPflow mean function's trainable
 * Params (gpflow.mean_functions.Constant.c, Linear.A / .b; sgpr_ss.py:14,25) are differentiated through — TF autodiff in the
 * reference: d bound / d theta_m = - sum_n r_n d mean(x_n) / d theta_m.  GP_ERR_BAD_ARG when no such evaluation precedes it. */
gp_status gp_sgpr_residual_grad(gp_sgpr_plan p, const double* params, const double* Y, int32_t N, double* r_dev);

/* One window sharded over its FRAMES across several GPUs (one process per GPU; SURVEY 8e: "SGPRSS single large
 * window").  Each rank holds a slice (X, Y) of N frames; Z and params are replicated.  The collapsed bound couples
 * the slices only through H = A'A'^T (M x M), u = A'y (M), sum y^2 and tr(H):
 *   gp_sgpr_bound_begin : local Kuf, A' = L^-1 Kuf, H, u, scalars -> exchange[0 .. gp_sgpr_exchange_doubles())
 *   -- caller: ncclAllReduce(exchange, M*M + M + 2, sum) --
 *   gp_sgpr_bound_end   : Cholesky of B, the bound for N_total frames (identical on every rank) and, when
 *                         grad != NULL, this rank's share of the gradient: the frame-dependent terms of its slice,
 *                         plus the frame-independent ones (noise, Kdiag, Kuu side, L1 penalty) only when
 *                         include_replicated != 0 (pass 1 on exactly one rank)
 *   -- caller: ncclAllReduce(grad, gp_sgpr_num_params, sum) -> the full gradient everywhere. */
int64_t gp_sgpr_exchange_doubles(gp_sgpr_plan p);
gp_status gp_sgpr_bound_begin(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                              const double* Z, double* exchange);
gp_status gp_sgpr_bound_end(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                            int64_t N_total, const double* Z, const double* exchange, double* bound_dev,
                            double* bound_host, double* grad, int32_t include_replicated);

/* gp_sgpr_bound_grad is launch-bound at window sizes (N ~ 2001, ~50 small kernels), and L-BFGS-B calls it dozens
 * of times per window with the same buffers: from the second call with identical pointer arguments the launch
 * sequence is recorded into a hipGraph and replayed.  Needs a handle created on a real stream (the legacy null
 * stream cannot be captured) and timers off; otherwise the launches stay eager.  enable = 0 turns it off.
 * gp_sgpr_eval_counts reports how many evaluations ran eagerly / were captured / were replayed. */
gp_status gp_sgpr_set_graphs(gp_sgpr_plan p, int32_t enable);
gp_status gp_sgpr_eval_counts(gp_sgpr_plan p, int64_t* eager, int64_t* captured, int64_t* replayed);
/* GPflow SGPR.build_predict (predict_f; separation.py:306) at Xnew: mean, var (n values each) */
gp_status gp_sgpr_predict_f(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                            const double* Z, const double* Xnew, int32_t n, double* mean, double* var);
/* SGPRSS.build_predict_source / predict_s (sgpr_ss.py:73-114): exact GP with an N x N Cholesky;
 * mean/var are P x n row-major.  workspace: gp_sgpr_predict_source_workspace_bytes(N, n). */
/* full_cov = True forms (SGPR.build_predict / sgpr_ss.py:95-99; unused by the reference's callers): cov is n x n
 * (predict_f) or [P][n][n] (predict_source), row-major; `var` still receives the diagonal form.  float64 plans. */
gp_status gp_sgpr_predict_f_full(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                 const double* Z, const double* Xnew, int32_t n, double* mean, double* var, double* cov);
gp_status gp_sgpr_predict_source_full(gp_sgpr_plan p, const double* params, const double* X, const double* Y,
                                      int32_t N, const double* Xnew, int32_t n, double* mean, double* var, double* cov,
                                      void* workspace, size_t workspace_bytes);
size_t gp_sgpr_predict_source_workspace_bytes(int32_t N, int32_t n);
gp_status gp_sgpr_predict_source(gp_sgpr_plan p, const double* params, const double* X, const double* Y,
                                 int32_t N, const double* Xnew, int32_t n, double* mean, double* var,
                                 void* workspace, size_t workspace_bytes);

/* ---- many independent SGPRSS windows per launch sequence -------------------------------------------------------
 * replaces the window loop of AMT.optimize / SoSp.optimize (gpitch/transcription.py:265-288, gpitch/separation.py:279-313):
 * for each window, reset_model then model.optimize(maxiter) = a dozen-odd evaluations of SGPRSS.build_likelihood
 * (sgpr_ss.py:29-71) and its gradient.  Windows are independent; this plan evaluates W of them with one sequence of
 * launches (every kernel runs over a window index), so that N = 2001-frame windows fill the device.
 * All windows share max_N (= N), M and the kernel structure of `cfg`; contiguous layouts:
 *   params [W][gp_sgprb_num_params], X [W][N], Y [W][N], Z [W][M], bound_dev [W], grad [W][num_params] (may be NULL).
 * `count` <= W windows (the first `count` slots) are evaluated.  Asynchronous on the handle's stream; from the second
 * call with the same buffers the sequence is replayed from a hipGraph.  M <= 256. */
gp_status gp_sgprb_create(gp_handle h, const gp_sgpr_config* cfg, int32_t num_windows, gp_sgprb_plan* out);
gp_status gp_sgprb_destroy(gp_sgprb_plan p);
int64_t gp_sgprb_num_params(gp_sgprb_plan p);
int32_t gp_sgprb_num_windows(gp_sgprb_plan p);
size_t gp_sgprb_workspace_bytes(gp_sgprb_plan p);
gp_status gp_sgprb_set_workspace(gp_sgprb_plan p, void* workspace, size_t bytes);
gp_status gp_sgprb_bound_grad(gp_sgprb_plan p, const double* params, const double* X, const double* Y, const double* Z,
                              int32_t count, double* bound_dev, double* grad);
gp_status gp_sgprb_set_graphs(gp_sgprb_plan p, int32_t enable);
gp_status gp_sgprb_eval_counts(gp_sgprb_plan p, int64_t* eager, int64_t* captured, int64_t* replayed);
/* The rest of SoSp.optimize's loop body (gpitch/separation.py:300-313), window-batched: after a window's optimisation
 * the reference calls model.predict_f(X_i) (GPflow 0.5 SGPR.build_predict) and model.predict_s(X_i) (sgpr_ss.py:73-114).
 * gp_sgprb_predict_f: Xnew [count][n] (n <= N), mean / var [count][n]; runs the forward pass at `params` first.
 * gp_sgprb_predict_source: the exact GP on each window's N frames (N x N Cholesky per window, every launch of the blocked
 * factorisation carrying all windows); Xnew [count][n], mean / var [count][P][n]; workspace from
 * gp_sgprb_predict_source_workspace_bytes(p, count, n) (about 3 N^2 doubles per window), 256-byte aligned.
 * Both synchronise the stream and report a failed factorisation (GP_ERR_NOT_PD). */
gp_status gp_sgprb_predict_f(gp_sgprb_plan p, const double* params, const double* X, const double* Y, const double* Z,
                             const double* Xnew, int32_t n, int32_t count, double* mean, double* var);
size_t gp_sgprb_predict_source_workspace_bytes(gp_sgprb_plan p, int32_t count, int32_t n);
gp_status gp_sgprb_predict_source(gp_sgprb_plan p, const double* params, const double* X, const double* Y,
                                  const double* Xnew, int32_t n, int32_t count, double* mean, double* var,
                                  void* workspace, size_t workspace_bytes);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------
 * HIP-event timing of the dominant kernels on the handle's own stream.  Returns the accumulated time of
 * kernel class `which` since the last reset and the number of launches. */
enum { GP_TIMER_KUF_BUILD = 0,   /* cov_build_kernel<0,2>: stationary Kuf assembly (HBM-bound)                 */
       GP_TIMER_COND_A = 1,      /* gemm_f64_kernel<..,1>: A = L^-1 Kuf (tri-aware, M^2 N flops per GP)        */
       GP_TIMER_COND_LTA = 2,    /* gemm_f64_kernel<..,2>: Lq^T A column sums (tri-aware, M^2 N)               */
       GP_TIMER_NT_GEMM = 3,     /* gemm_f64_kernel<..,4>: H = A D A^T split-K over frames (M^2 N, symmetric)  */
       GP_TIMER_KUF_BAR = 4,     /* gemm_f64_kernel<..,3>: Kuf_bar = R (A D) (dense, 2 M^2 N)                  */
       GP_TIMER_CHOL = 5, GP_TIMER_LIK = 6, GP_TIMER_SMALL_GEMM = 7, GP_TIMER_HYPER = 8,
       GP_TIMER_KUF_BUILD_SM = 9, /* cov_build_kernel<1,..>: spectral-mixture Kuf (features + build)              */
       GP_TIMER_COUNT = 10 };
gp_status gp_timers_enable(gp_handle h, int32_t on);
gp_status gp_timers_reset(gp_handle h);
gp_status gp_timers_read(gp_handle h, int32_t which, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* GPITCH_ABI_H */
