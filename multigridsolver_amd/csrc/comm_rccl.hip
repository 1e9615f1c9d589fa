// comm_rccl.hip — native RCCL transport of the row-sharded cycle (one process per GPU, xGMI point-to-point).
//
// The exchange step of the path is a neighbour exchange of halo values: one RCCL group of
// ncclSend/ncclRecv to the peers that own halo columns — straight from the source vector where a peer's rows are a few
// contiguous ranges (plane shards), behind a pack kernel otherwise — enqueued by the C++ cycle itself on the
// context's stream, so a sharded cycle needs no host callback and no cross-stream dependency per exchange.  The
// replicated coarse tail gathers its right-hand side with one ncclAllGather.
//
// librccl is NOT a link dependency: the process has already loaded the copy its launcher (torch.distributed) uses;
// the entry points are resolved from that same file with dlopen/dlsym (path handed in by the host side), so there is
// exactly one RCCL instance per process.  <rccl/rccl.h> supplies types and enums only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "mgs_internal.hpp"

struct mgs_comm {
  mgs_ctx *ctx = nullptr;
  mgs_p2p *p2p = nullptr;     // peer-to-peer transport (comm_p2p.hip): the three operations below go through it, RCCL is not loaded
  void *dl = nullptr;
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0;
  bool capturable = true;     // false: host-synchronous stand-in (tests/fake_rccl exports mgs_fake_rccl_marker = 1; = 2: its stream-ordered mode, capturable)
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

#define MGS_NCCL(c, call)                                                                                  \
  do {                                                                                                     \
    ncclResult_t r_ = (call);                                                                              \
    if (r_ != ncclSuccess)                                                                                 \
      return mgs_fail((c)->ctx, MGS_ERR_STATE, "%s failed: %s", #call, (c)->GetErrorString ? (c)->GetErrorString(r_) : "?"); \
  } while (0)

static int load_api(mgs_ctx *ctx, const char *librccl, mgs_comm *c) {
  c->ctx = ctx;
  c->dl = dlopen(librccl, RTLD_NOW | RTLD_GLOBAL);
  if (!c->dl) return mgs_fail(ctx, MGS_ERR_STATE, "dlopen(%s): %s", librccl ? librccl : "(null)", dlerror());
#define SYM(field, name)                                                               \
  c->field = reinterpret_cast<decltype(c->field)>(dlsym(c->dl, name));                 \
  if (!c->field) return mgs_fail(ctx, MGS_ERR_STATE, "%s: symbol %s missing", librccl, name)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  const int *marker = reinterpret_cast<const int *>(dlsym(c->dl, "mgs_fake_rccl_marker"));     // tests/fake_rccl: 1 = host-synchronous calls, 2 = stream-ordered
  c->capturable = !marker || *marker == 2;
  return MGS_OK;
}
bool mgs_comm_capturable(const mgs_comm *c) { return c && c->capturable; }

extern "C" {

int mgs_comm_unique_id(mgs_ctx *ctx, const char *librccl, void *id_out) {
  MGS_CHECK(ctx, id_out, MGS_ERR_INVALID, "mgs_comm_unique_id: NULL output");
  mgs_comm tmp;
  MGS_TRY(load_api(ctx, librccl, &tmp));
  ncclUniqueId id;
  MGS_NCCL(&tmp, tmp.GetUniqueId(&id));
  static_assert(sizeof(id) == MGS_COMM_ID_BYTES, "RCCL unique id size");
  memcpy(id_out, &id, sizeof id);
  return MGS_OK;
}

int mgs_comm_create(mgs_ctx *ctx, const char *librccl, const void *id_in, int world, int rank, mgs_comm **out) {
  MGS_CHECK(ctx, out && id_in && world >= 1 && rank >= 0 && rank < world, MGS_ERR_INVALID, "mgs_comm_create: bad arguments");
  mgs_comm *c = new mgs_comm();
  int rc = load_api(ctx, librccl, c);
  if (rc != MGS_OK) { delete c; return rc; }
  c->world = world; c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof id);
  hipSetDevice(ctx->device);
  ncclResult_t r = c->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { rc = mgs_fail(ctx, MGS_ERR_STATE, "ncclCommInitRank(world %d, rank %d): %s", world, rank, c->GetErrorString(r)); delete c; return rc; }
  *out = c;
  return MGS_OK;
}

int mgs_comm_destroy(mgs_comm *c) {
  if (!c) return MGS_OK;
  if (c->p2p) { mgs_p2p_destroy(c->p2p); delete c; return MGS_OK; }
  if (c->comm && c->CommDestroy) c->CommDestroy(c->comm);
  delete c;      // the library handle stays open: the process keeps using it
  return MGS_OK;
}

// ---- peer-to-peer transport behind the same handle (comm_p2p.hip)
int mgs_comm_p2p_create(mgs_ctx *ctx, int world, int rank, size_t slot_doubles, void *handle_out, mgs_comm **out) {
  MGS_CHECK(ctx, out, MGS_ERR_INVALID, "mgs_comm_p2p_create: NULL output");
  mgs_comm *c = new mgs_comm();
  c->ctx = ctx; c->world = world; c->rank = rank; c->capturable = true;
  const int rc = mgs_p2p_create(ctx, world, rank, slot_doubles, handle_out, &c->p2p);
  if (rc != MGS_OK) { delete c; return rc; }
  *out = c;
  return MGS_OK;
}
int mgs_comm_p2p_connect(mgs_comm *c, const void *handles) {
  MGS_CHECK(c->ctx, c->p2p, MGS_ERR_STATE, "mgs_comm_p2p_connect: not a peer-to-peer communicator");
  return mgs_p2p_connect(c->p2p, handles);
}
int mgs_comm_p2p_selftest(mgs_comm *c, int rounds, long long *mismatches) {
  MGS_CHECK(c->ctx, c->p2p, MGS_ERR_STATE, "mgs_comm_p2p_selftest: not a peer-to-peer communicator");
  return mgs_p2p_selftest(c->p2p, c->ctx->stream, rounds, mismatches);
}
int mgs_comm_p2p_info(const mgs_comm *c, long long out[6]) {
  MGS_CHECK(c->ctx, c->p2p && out, MGS_ERR_STATE, "mgs_comm_p2p_info: not a peer-to-peer communicator");
  return mgs_p2p_info(c->p2p, out);
}

// raw operations of a communicator on the context's stream (transport tests and microbenchmarks; the cycle calls the C++ forms below)
int mgs_comm_exchange_raw(mgs_comm *c, int nops, const int *peer, const size_t *count, const void *const *send_dev, void *const *recv_dev) {
  MGS_CHECK(c->ctx, nops >= 0 && (nops == 0 || (peer && count && send_dev && recv_dev)), MGS_ERR_INVALID, "mgs_comm_exchange_raw: bad arguments");
  std::vector<mgs_xfer_op> ops((size_t)nops);
  for (int q = 0; q < nops; ++q) {
    MGS_CHECK(c->ctx, (send_dev[q] != nullptr) != (recv_dev[q] != nullptr) || count[q] == 0, MGS_ERR_INVALID, "mgs_comm_exchange_raw: op %d must be a send or a receive", q);
    ops[(size_t)q] = mgs_xfer_op{(const double *)send_dev[q], (double *)recv_dev[q], count[q], peer[q]};
  }
  return mgs_comm_exchange_ops(c, ops.data(), nops);
}
int mgs_comm_allgather_raw(mgs_comm *c, const void *send_dev, void *recv_dev, size_t count) { return mgs_comm_allgather(c, (const double *)send_dev, (double *)recv_dev, count); }
int mgs_comm_allreduce_raw(mgs_comm *c, void *buf_dev, size_t count) { return mgs_comm_allreduce_sum(c, (double *)buf_dev, count); }

int mgs_comm_size(const mgs_comm *c, int *world, int *rank) { if (world) *world = c->world; if (rank) *rank = c->rank; return MGS_OK; }

}  // extern "C"

// neighbour exchange on the context's stream: one RCCL group of sends and receives.  The ops addressed to one peer are matched with
// that peer's ops in posting order (k-th send to p ↔ p's k-th receive from this rank), so a peer's payload may travel as several
// contiguous ranges taken straight from the source vector.  Zero counts issue nothing.
int mgs_comm_exchange_ops(mgs_comm *c, const mgs_xfer_op *ops, int nops) {
  hipStream_t s = c->ctx->stream;
  if (c->p2p) return mgs_p2p_exchange_ops(c->p2p, s, ops, nops);
  MGS_NCCL(c, c->GroupStart());
  ncclResult_t bad = ncclSuccess;
  for (int q = 0; q < nops && bad == ncclSuccess; ++q) {
    const mgs_xfer_op &o = ops[q];
    if (!o.count) continue;
    bad = o.sptr ? c->Send(o.sptr, o.count, ncclDouble, o.peer, c->comm, s) : c->Recv(o.rptr, o.count, ncclDouble, o.peer, c->comm, s);
  }
  const ncclResult_t end = c->GroupEnd();          // always closed, also after a failed post
  if (bad != ncclSuccess) return mgs_fail(c->ctx, MGS_ERR_STATE, "ncclSend/ncclRecv failed: %s", c->GetErrorString(bad));
  if (end != ncclSuccess) return mgs_fail(c->ctx, MGS_ERR_STATE, "ncclGroupEnd failed: %s", c->GetErrorString(end));
  return MGS_OK;
}
int mgs_comm_allgather(mgs_comm *c, const double *send, double *recv, size_t count) {
  if (c->p2p) return mgs_p2p_allgather(c->p2p, c->ctx->stream, send, recv, count);
  MGS_NCCL(c, c->AllGather(send, recv, count, ncclDouble, c->comm, c->ctx->stream));
  return MGS_OK;
}
int mgs_comm_allreduce_sum(mgs_comm *c, double *buf, size_t count) {
  if (c->p2p) return mgs_p2p_allreduce_sum(c->p2p, c->ctx->stream, buf, count);
  MGS_NCCL(c, c->AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, c->ctx->stream));
  return MGS_OK;
}
