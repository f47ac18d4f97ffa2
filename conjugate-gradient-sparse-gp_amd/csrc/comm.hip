// comm.hip -- RCCL entry points of the C ABI: one communicator rank per GPU, the
// single exchange of the path is an in-place all-reduce(sum) of the [Bt, M] partial product
// (plus one agreement word, cg.hip) per operator application, enqueued on the solve's stream.
//
// The reference has no collective anywhere; nothing here mirrors reference code.
// Payloads are tiny (32 KiB at C3, Bt = 1), so the call is latency-bound: it is issued directly
// from mgp_pcg_solve's enqueue loop on the compute stream -- no host callback, no extra copy.
#include <rccl/rccl.h>

#include "mgp_common.h"

struct mgp_comm {
  ncclComm_t comm = nullptr;
  int device = -1;
  int nranks = 0;
  int rank = -1;
};

namespace {
thread_local char g_comm_err[512] = {0};

int comm_fail(int code, const char* what, ncclResult_t r) {
  snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, ncclGetErrorString(r));
  return code;
}
int comm_fail_msg(int code, const char* msg) {
  snprintf(g_comm_err, sizeof(g_comm_err), "%s", msg);
  return code;
}
}  // namespace

static_assert(MGP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "mgp.h advertises the RCCL unique-id size");

extern "C" const char* mgp_comm_last_error(void) { return g_comm_err; }

extern "C" int mgp_comm_unique_id(void* id_out) {
  if (!id_out) return comm_fail_msg(MGP_E_BADARG, "id_out is NULL");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return comm_fail(MGP_E_COMM, "ncclGetUniqueId", r);
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MGP_OK;
}

extern "C" int mgp_comm_init_rank(mgp_comm** out, int device, int nranks, int rank, const void* id) {
  if (!out || !id) return comm_fail_msg(MGP_E_BADARG, "out / id is NULL");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return comm_fail_msg(MGP_E_BADARG, "rank outside [0, nranks)");
  if (hipSetDevice(device) != hipSuccess) return comm_fail_msg(MGP_E_HIP, "hipSetDevice failed");
  mgp_comm* c = new (std::nothrow) mgp_comm();
  if (!c) return comm_fail_msg(MGP_E_NOMEM, "out of host memory");
  ncclUniqueId uid;
  memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return comm_fail(MGP_E_COMM, "ncclCommInitRank", r);
  }
  c->device = device;
  c->nranks = nranks;
  c->rank = rank;
  *out = c;
  return MGP_OK;
}

extern "C" int mgp_comm_init_all(int ndev, const int* devs, mgp_comm** comms) {
  if (ndev < 1 || !comms) return comm_fail_msg(MGP_E_BADARG, "ndev < 1 or comms is NULL");
  std::vector<ncclComm_t> raw((size_t)ndev);
  std::vector<int> list((size_t)ndev);
  for (int i = 0; i < ndev; ++i) list[(size_t)i] = devs ? devs[i] : i;
  const ncclResult_t r = ncclCommInitAll(raw.data(), ndev, list.data());
  if (r != ncclSuccess) return comm_fail(MGP_E_COMM, "ncclCommInitAll", r);
  for (int i = 0; i < ndev; ++i) {
    mgp_comm* c = new (std::nothrow) mgp_comm();
    if (!c) {
      for (int j = 0; j < i; ++j) delete comms[j];
      for (int j = 0; j < ndev; ++j) (void)ncclCommDestroy(raw[(size_t)j]);
      return comm_fail_msg(MGP_E_NOMEM, "out of host memory");
    }
    c->comm = raw[(size_t)i];
    c->device = list[(size_t)i];
    c->nranks = ndev;
    c->rank = i;
    comms[i] = c;
  }
  return MGP_OK;
}

extern "C" int mgp_comm_destroy(mgp_comm* c) {
  if (!c) return MGP_OK;
  ncclResult_t r = ncclSuccess;
  if (c->comm) r = ncclCommDestroy(c->comm);
  delete c;
  return r == ncclSuccess ? MGP_OK : comm_fail(MGP_E_COMM, "ncclCommDestroy", r);
}

extern "C" int mgp_comm_size(const mgp_comm* c) { return c ? c->nranks : 0; }
extern "C" int mgp_comm_rank(const mgp_comm* c) { return c ? c->rank : -1; }

extern "C" int mgp_comm_group_begin(void) {
  const ncclResult_t r = ncclGroupStart();
  return r == ncclSuccess ? MGP_OK : comm_fail(MGP_E_COMM, "ncclGroupStart", r);
}
extern "C" int mgp_comm_group_end(void) {
  const ncclResult_t r = ncclGroupEnd();
  return r == ncclSuccess ? MGP_OK : comm_fail(MGP_E_COMM, "ncclGroupEnd", r);
}

extern "C" int mgp_allreduce_sum(void* buf, size_t count, int dtype, mgp_comm* comm, void* stream) {
  if (!comm || !comm->comm) return comm_fail_msg(MGP_E_BADARG, "comm is NULL");
  if (dtype != MGP_F32 && dtype != MGP_F64) return comm_fail_msg(MGP_E_DTYPE, "bad dtype");
  if (count == 0) return MGP_OK;
  if (!buf) return comm_fail_msg(MGP_E_BADARG, "buf is NULL");
  const ncclResult_t r = ncclAllReduce(buf, buf, count, dtype == MGP_F64 ? ncclDouble : ncclFloat, ncclSum, comm->comm,
                                       (hipStream_t)stream);
  return r == ncclSuccess ? MGP_OK : comm_fail(MGP_E_COMM, "ncclAllReduce", r);
}

// used by cg.hip: the operator's collective on the handle's stream, errors into the handle
int mgp_comm_allreduce_on(mgp_handle* h, mgp_comm* comm, void* buf, size_t count, int dtype) {
  const int rc = mgp_allreduce_sum(buf, count, dtype, comm, (void*)h->stream);
  if (rc != MGP_OK) return mgp_fail(h, rc, "all-reduce failed: %s", g_comm_err);
  return MGP_OK;
}
