// comm.hip — the exchange step of a model sharded over the GPUs of one node as part of ONE library call (SURVEY section 5,
// section 8b: the handle "owns workspace, stream, RCCL comm"; section 8e).
//
// The reference has no distributed code (gpitch/transcription.py:265-288 fits windows one after another); the sharded forms
// are this build's (include/gpitch_abi.h: gp_pdgp_elbo_begin/_end, gp_pdgp_cond_begin/_end, gp_sgpr_bound_begin/_end).  Rounds
// 2-3 issued the collective between the two stages from Python through torch.distributed; here the library issues
// ncclAllReduce / ncclAllGather itself, on the handle's stream, between the stages — so begin -> exchange -> end (-> Adam) is
// one enqueue with no host code in between, and the same sequence can sit in a hipGraph.  RCCL is bound at run time (dlopen:
// the copy the process already holds — torch's — or ROCm's), so the library loads, and every other entry point works, where
// there is no RCCL; gp_comm_create then returns GP_ERR_UNSUPPORTED.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {
// the handful of RCCL entry points used (rccl.h: ncclResult_t = int, 0 = ncclSuccess; ncclUniqueId = 128 bytes;
// ncclDataType_t ncclFloat64 = 8, ncclRedOp_t ncclSum = 0)
struct NcclId { char internal[128]; };
typedef void* NcclComm;
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
const int kFloat64 = 8, kSum = 0;

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process (torch's)
    for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather;
  });
  return r;
}
}  // namespace

struct gp_comm_s {
  gp_handle h = nullptr;
  NcclComm comm = nullptr;
  int rank = 0, world = 1;
};

static gp_status comm_fail(gp_handle h, int rc, const char* what) {
  char b[256];
  const Rccl& r = rccl();
  snprintf(b, sizeof(b), "%s failed: %s", what, (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
  return gp_fail(h, GP_ERR_HIP, b);
}

// fmean_full[g], fvar_full[g] (model order) <- rank (g mod world)'s row (g div world) of the gathered blocks; kl_total = the
// sum of the ranks' KL slots in rank order (fixed order: identical on every rank)
__global__ void __launch_bounds__(256) gp_assemble_kernel(const double* __restrict__ recv, int num_gps, int world, int per, int n,
                                                           double* __restrict__ fm, double* __restrict__ fv, double* __restrict__ kl) {
  const int64_t blk = (int64_t)2 * per * n + 8;
  const int g = blockIdx.y;
  const double* src = recv + (int64_t)(g % world) * blk + (int64_t)(g / world) * n;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    fm[(int64_t)g * n + j] = src[j];
    fv[(int64_t)g * n + j] = src[(int64_t)per * n + j];
  }
  if (g == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int r = 0; r < world; r++) s += recv[(int64_t)r * blk + (int64_t)2 * per * n];
    kl[0] = s;
  }
}

extern "C" {

gp_status gp_comm_unique_id(uint8_t* id128) {
  if (!id128) return GP_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return GP_ERR_UNSUPPORTED;
  NcclId id;
  if (r.GetUniqueId(&id) != 0) return GP_ERR_HIP;
  memcpy(id128, id.internal, 128);
  return GP_OK;
}

gp_status gp_comm_create(gp_handle h, const uint8_t* id128, int32_t rank, int32_t world, gp_comm* out) {
  if (!h || !id128 || !out || world < 1 || rank < 0 || rank >= world) return gp_fail(h, GP_ERR_BAD_ARG, "gp_comm_create: bad argument");
  const Rccl& r = rccl();
  if (!r.ok) return gp_fail(h, GP_ERR_UNSUPPORTED, "gp_comm_create: no RCCL library (librccl.so) in this process or under /opt/rocm/lib");
  GP_HIP_CHECK(h, hipSetDevice(h->device));
  NcclId id;
  memcpy(id.internal, id128, 128);
  gp_comm c = new gp_comm_s();
  c->h = h; c->rank = rank; c->world = world;
  const int rc = r.CommInitRank(&c->comm, world, id, rank);
  if (rc != 0) { delete c; return comm_fail(h, rc, "ncclCommInitRank"); }
  *out = c;
  return GP_OK;
}

gp_status gp_comm_destroy(gp_comm c) {
  if (!c) return GP_OK;
  if (c->comm) { (void)hipStreamSynchronize(c->h->stream); rccl().CommDestroy(c->comm); }
  delete c;
  return GP_OK;
}

int32_t gp_comm_world(gp_comm c) { return c ? c->world : 0; }
int32_t gp_comm_rank(gp_comm c) { return c ? c->rank : -1; }

gp_status gp_comm_allreduce_sum(gp_comm c, double* buf, int64_t count) {
  if (!c || !buf || count < 0) return GP_ERR_BAD_ARG;
  const int rc = rccl().AllReduce(buf, buf, (size_t)count, kFloat64, kSum, c->comm, c->h->stream);
  return rc == 0 ? GP_OK : comm_fail(c->h, rc, "ncclAllReduce");
}

static gp_status adam_if(gp_comm c, const gp_adam_args* a, double* params, const double* grad) {
  if (!a) return GP_OK;
  return gp_adam_step(c->h, a->free_state, params, grad, a->tcode, a->m, a->v, a->nparams, a->t, a->lr, a->beta1, a->beta2, a->eps);
}

gp_status gp_pdgp_elbo_pitch_sharded(gp_pdgp_plan p, gp_comm c, double* params, const double* x, const double* y, int32_t n,
                                     double num_data, double* exchange, double* elbo_dev, double* elbo_host, double* grad,
                                     const gp_adam_args* adam) {
  if (!p || !c || !exchange) return GP_ERR_BAD_ARG;
  if (adam && !grad) return gp_fail(c->h, GP_ERR_BAD_ARG, "gp_pdgp_elbo_pitch_sharded: an optimiser step needs the gradient");
  GP_CHECK(gp_pdgp_elbo_begin(p, params, x, y, n, grad, exchange));
  GP_CHECK(gp_comm_allreduce_sum(c, exchange, (int64_t)3 * n + 1));
  GP_CHECK(gp_pdgp_elbo_end(p, params, x, y, n, num_data, exchange, elbo_dev, adam ? nullptr : elbo_host, grad));
  GP_CHECK(adam_if(c, adam, params, grad));
  if (adam && elbo_host) {     // (the ELBO of the parameters BEFORE the step, as gp_pdgp_elbo followed by gp_adam_step would give)
    GP_HIP_CHECK(c->h, hipMemcpyAsync(elbo_host, elbo_dev, sizeof(double), hipMemcpyDeviceToHost, c->h->stream));
    GP_HIP_CHECK(c->h, hipStreamSynchronize(c->h->stream));
  }
  return GP_OK;
}

gp_status gp_pdgp_elbo_gp_sharded(gp_pdgp_plan p, gp_comm c, double* params, const double* x, const double* y, int32_t n,
                                  double num_data, int32_t num_gps, int32_t local_gps, double* send, double* recv, double* full,
                                  double* elbo_dev, double* elbo_host, double* grad, const gp_adam_args* adam) {
  if (!p || !c || !send || !recv || !full || num_gps < 1 || local_gps < 0 || n < 1) return GP_ERR_BAD_ARG;
  if (adam && !grad) return gp_fail(c->h, GP_ERR_BAD_ARG, "gp_pdgp_elbo_gp_sharded: an optimiser step needs the gradient");
  gp_handle h = c->h;
  const int world = c->world, per = (num_gps + world - 1) / world;
  const int64_t blk = (int64_t)2 * per * n + 8;
  if (local_gps < per) GP_HIP_CHECK(h, hipMemsetAsync(send, 0, (size_t)blk * sizeof(double), h->stream));   // the last row slot is padding
  GP_CHECK(gp_pdgp_cond_begin(p, params, x, n, grad, send, send + (int64_t)per * n, send + (int64_t)2 * per * n));
  {
    const int rc = rccl().AllGather(send, recv, (size_t)blk, kFloat64, c->comm, h->stream);
    if (rc != 0) return comm_fail(h, rc, "ncclAllGather");
  }
  double* fm = full; double* fv = full + (int64_t)num_gps * n; double* kl = full + (int64_t)2 * num_gps * n;
  hipLaunchKernelGGL(gp_assemble_kernel, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256, num_gps), dim3(256), 0, h->stream, recv,
                     num_gps, world, per, n, fm, fv, kl);
  GP_HIP_CHECK(h, hipGetLastError());
  GP_CHECK(gp_pdgp_cond_end(p, params, x, y, n, num_data, fm, fv, kl, elbo_dev, adam ? nullptr : elbo_host, grad));
  GP_CHECK(adam_if(c, adam, params, grad));
  if (adam && elbo_host) {
    GP_HIP_CHECK(h, hipMemcpyAsync(elbo_host, elbo_dev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GP_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  return GP_OK;
}

gp_status gp_sgpr_bound_grad_sharded(gp_sgpr_plan p, gp_comm c, const double* params, const double* X, const double* Y, int32_t N,
                                     int64_t N_total, const double* Z, double* exchange, double* bound_dev, double* bound_host,
                                     double* grad) {
  if (!p || !c || !exchange) return GP_ERR_BAD_ARG;
  GP_CHECK(gp_sgpr_bound_begin(p, params, X, Y, N, Z, exchange));
  GP_CHECK(gp_comm_allreduce_sum(c, exchange, gp_sgpr_exchange_doubles(p)));
  GP_CHECK(gp_sgpr_bound_end(p, params, X, Y, N, N_total, Z, exchange, bound_dev, grad ? nullptr : bound_host, grad, c->rank == 0 ? 1 : 0));
  if (grad) {
    GP_CHECK(gp_comm_allreduce_sum(c, grad, gp_sgpr_num_params(p)));
    if (bound_host) {
      GP_HIP_CHECK(c->h, hipMemcpyAsync(bound_host, bound_dev, sizeof(double), hipMemcpyDeviceToHost, c->h->stream));
      GP_HIP_CHECK(c->h, hipStreamSynchronize(c->h->stream));
    }
  }
  return GP_OK;
}

}  // extern "C"
