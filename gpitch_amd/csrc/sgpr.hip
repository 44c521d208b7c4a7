// sgpr.hip — SGPRSS plan (gpitch/sgpr_ss.py:10-114).  Entry points are declared in gpitch_abi.h.
#include "engine.h"

struct gp_sgpr_plan_s {
  gp_handle h = nullptr;
};

extern "C" {

gp_status gp_sgpr_create(gp_handle h, const gp_sgpr_config* cfg, gp_sgpr_plan* out) {
  (void)cfg; if (out) *out = nullptr;
  return gp_fail(h, GP_ERR_UNSUPPORTED, "gp_sgpr_create: not implemented yet");
}
gp_status gp_sgpr_destroy(gp_sgpr_plan p) { delete p; return GP_OK; }
int64_t gp_sgpr_num_params(gp_sgpr_plan p) { (void)p; return 0; }
size_t gp_sgpr_workspace_bytes(gp_sgpr_plan p) { (void)p; return 0; }
gp_status gp_sgpr_set_workspace(gp_sgpr_plan p, void* workspace, size_t bytes) { (void)p; (void)workspace; (void)bytes; return GP_ERR_UNSUPPORTED; }
gp_status gp_sgpr_bound(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                        const double* Z, double* bound_dev, double* bound_host) {
  (void)p; (void)params; (void)X; (void)Y; (void)N; (void)Z; (void)bound_dev; (void)bound_host; return GP_ERR_UNSUPPORTED;
}
gp_status gp_sgpr_predict_f(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                            const double* Z, const double* Xnew, int32_t n, double* mean, double* var) {
  (void)p; (void)params; (void)X; (void)Y; (void)N; (void)Z; (void)Xnew; (void)n; (void)mean; (void)var; return GP_ERR_UNSUPPORTED;
}
size_t gp_sgpr_predict_source_workspace_bytes(int32_t N, int32_t n) { (void)N; (void)n; return 0; }
gp_status gp_sgpr_predict_source(gp_sgpr_plan p, const double* params, const double* X, const double* Y, int32_t N,
                                 const double* Xnew, int32_t n, double* mean, double* var, void* workspace,
                                 size_t workspace_bytes) {
  (void)p; (void)params; (void)X; (void)Y; (void)N; (void)Xnew; (void)n; (void)mean; (void)var; (void)workspace; (void)workspace_bytes;
  return GP_ERR_UNSUPPORTED;
}

}  // extern "C"
