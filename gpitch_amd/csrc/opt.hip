// opt.hip — GPflow free-state transforms and the TF-1.2 Adam update, on device (gfx950).
//
// Replaces the host side of gpflow Model.optimize(method=tf.train.AdamOptimizer) as driven by
// demos/scripts/demo-modgp.py:44-45: Param free-state packing with transforms.positive (Log1pe,
// lower = 1e-6; used at gpitch/likelihoods.py:283, gpitch/matern12_spectral_mixture.py:26-32,86-94)
// and the Adam slots.  HBM-bound elementwise pass over the flat parameter vector, 16-byte friendly.
#include "common.h"

__device__ __forceinline__ double softplus_pos(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))) + 1e-6; }

// transform codes: 0 identity, 1 positive (Log1pe + 1e-6), 2 fixed, 3.. = gpflow.transforms.Logistic(a, b) with
// (a, b) = the handle's table entry code - 3 (gp_transform_register_logistic; kernels.py:219-223,333,
// init_models.py:189):  y = a + (b - a) / (1 + exp(-x)),  x = -log((b - a) / (y - a) - 1)
__device__ __forceinline__ double logistic_fwd(const GpLogisticTable& T, int t, double x) {
  const double a = T.a[t - 3], b = T.b[t - 3];
  return a + (b - a) / (1.0 + exp(-x));
}

__global__ void __launch_bounds__(256) transform_fwd_kernel(const double* __restrict__ fs, const uint8_t* __restrict__ tc,
                                                            int64_t n, double* __restrict__ params, GpLogisticTable T) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = tc[i];
    params[i] = (t == 1) ? softplus_pos(fs[i]) : (t >= 3 ? logistic_fwd(T, t, fs[i]) : fs[i]);
  }
}

__global__ void __launch_bounds__(256) transform_bwd_kernel(const double* __restrict__ params, const uint8_t* __restrict__ tc,
                                                            int64_t n, double* __restrict__ fs, GpLogisticTable T) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double y = params[i];
    const int t = tc[i];
    if (t == 1) { y -= 1e-6; y = y + log(-expm1(-y)); }
    else if (t >= 3) { const double a = T.a[t - 3], b = T.b[t - 3]; y = -log((b - a) / (y - a) - 1.0); }
    fs[i] = y;
  }
}

// maximise ELBO: minimise -ELBO.  g = -(dELBO/dparam) * dparam/dfree.
__global__ void __launch_bounds__(256) adam_kernel(double* __restrict__ fs, double* __restrict__ params,
                                                   const double* __restrict__ grad, const uint8_t* __restrict__ tc,
                                                   double* __restrict__ m, double* __restrict__ v, int64_t n, double lr_t,
                                                   double b1, double b2, double eps, GpLogisticTable T,
                                                   const int32_t* __restrict__ status) {
  // a Cholesky of this evaluation hit a non-positive pivot (chol_diag_block carried on with a substitute): the gradient
  // is garbage and the reference would have raised here (TF InvalidArgumentError) — leave the state untouched, the
  // host sees the flag at its next poll
  if (status && status[0] != 0) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint8_t t = tc[i];
    if (t == 2) continue;
    double x = fs[i];
    double g = -grad[i];
    if (t == 1) g *= 1.0 / (1.0 + exp(-x));
    else if (t >= 3) { const double s = 1.0 / (1.0 + exp(-x)); g *= (T.b[t - 3] - T.a[t - 3]) * s * (1.0 - s); }
    double mi = b1 * m[i] + (1.0 - b1) * g;
    double vi = b2 * v[i] + (1.0 - b2) * g * g;
    m[i] = mi; v[i] = vi;
    x -= lr_t * mi / (sqrt(vi) + eps);
    fs[i] = x;
    params[i] = (t == 1) ? softplus_pos(x) : (t >= 3 ? logistic_fwd(T, t, x) : x);
  }
}

static int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

gp_status launch_transform_forward(gp_handle h, const double* fs, const uint8_t* tc, int64_t n, double* params) {
  if (n <= 0) return GP_OK;
  hipLaunchKernelGGL(transform_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, fs, tc, n, params, h->logistic);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

gp_status launch_transform_backward(gp_handle h, const double* params, const uint8_t* tc, int64_t n, double* fs) {
  if (n <= 0) return GP_OK;
  hipLaunchKernelGGL(transform_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, params, tc, n, fs, h->logistic);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

gp_status launch_adam(gp_handle h, double* fs, double* params, const double* grad, const uint8_t* tc, double* m,
                      double* v, int64_t n, int64_t t, double lr, double b1, double b2, double eps) {
  if (n <= 0) return GP_OK;
  const double lr_t = lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t));
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, fs, params, grad, tc, m, v, n, lr_t, b1,
                     b2, eps, h->logistic, (const int32_t*)h->d_status);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Hann overlap-add of per-window predictions (gpitch/window_overlap.py:19-59, merged_mean / merged_variance): windows
// of odd length ws, hop ll = (ws - 1) / 2, n = ll (nw + 1) + 1 output frames.  Written as a gather: every output frame
// takes its (at most two) contributions, reproducing the reference's slice assignments, including which iteration
// wins on the shared boundary frames.  win = scipy.signal.hann(ws) (symmetric), flat over the first half of the first
// window and the last half of the last one; squared for variances.
__global__ void __launch_bounds__(256) overlap_merge_kernel(const double* __restrict__ y, int nw, int ws, int64_t ldy,
                                                            int n, int square, double* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int ll = (ws - 1) / 2;
  auto wv = [&](int k, int j) -> double {      // weighted sample j of window k
    double w = 0.5 - 0.5 * cos(6.283185307179586 * (double)j / (double)(ws - 1));
    if (k == 0 && j < ll) w = 1.0;
    if (k == nw - 1 && j >= ws - ll) w = 1.0;
    if (square) w = w * w;
    return y[(int64_t)k * ldy + j] * w;
  };
  double v;
  if (t < ll) v = wv(0, t);
  else if (t >= n - ll) v = wv(nw - 1, t - (nw - 1) * ll);
  else {
    int i = t / ll - 1;
    if (i > nw - 2) i = nw - 2;
    const int j = t - (i + 1) * ll;
    v = wv(i, ll + j) + wv(i + 1, j);
  }
  out[t] = v;
}

gp_status launch_overlap_merge(gp_handle h, const double* y, int nw, int ws, int64_t ldy, int n, int square, double* out) {
  hipLaunchKernelGGL(overlap_merge_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, y, nw, ws, ldy, n, square, out);
  GP_HIP_CHECK(h, hipGetLastError());
  return GP_OK;
}
