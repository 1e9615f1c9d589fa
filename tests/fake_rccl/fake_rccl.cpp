// fake_rccl.cpp — TEST INFRASTRUCTURE.  A stand-in for the handful of RCCL entry points libmgs resolves with dlsym
// (csrc/comm_rccl.hip), so the NATIVE transport of the sharded cycle can be exercised with several ranks on a box
// that has one GPU (real RCCL refuses two ranks on one device).  Messages travel through files under /tmp; every
// call is host-synchronous (stream drained, device→host→file→host→device), which keeps RCCL's ordering semantics:
// inside a group all sends are posted before any receive is waited for.  Never used by the product path:
// multigridsolver_amd/dist.py loads it only when MGS_LIBRCCL points here (tests/test_gpu_dist.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

// tells libmgs what these entry points are: 1 = host-synchronous (never captured into a hipGraph); 2 = STREAM-ORDERED
// (MGS_FAKE_RCCL_STREAM=1): every transfer is a device→pinned copy, a host function that moves the bytes through the files, and a
// pinned→device copy, all enqueued on the caller's stream — so the calls can be captured into a hipGraph and replayed like RCCL's
// kernels, and the captured multi-rank cycle (exchanges, tail all-gather, all-reduced K-cycle scalars) runs on one GPU.
extern "C" int mgs_fake_rccl_marker = 1;
namespace { struct ModeInit { ModeInit() { const char *e = getenv("MGS_FAKE_RCCL_STREAM"); if (e && e[0] == '1') mgs_fake_rccl_marker = 2; } } g_mode_init; }

struct ncclComm {
  std::string dir;
  int world = 0, rank = 0;
  std::vector<unsigned long long> sent, recvd;   // per peer sequence numbers
  char *arena = nullptr; size_t arena_cap = 0, arena_used = 0;   // stream-ordered mode: pinned staging, bump-allocated (no HIP call while a stream captures)
};

namespace {
struct Op { bool send; const void *sbuf; void *rbuf; size_t bytes; int peer; ncclComm *c; hipStream_t s; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

std::string msg_path(ncclComm *c, int src, int dst, unsigned long long seq) {
  char b[512]; snprintf(b, sizeof b, "%s/m_%d_%d_%llu", c->dir.c_str(), src, dst, seq); return b;
}
ncclResult_t do_send(const Op &o) {
  if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<char> h(o.bytes);
  if (o.bytes && hipMemcpy(h.data(), o.sbuf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  const std::string p = msg_path(o.c, o.c->rank, o.peer, o.c->sent[o.peer]++), tmp = p + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return ncclSystemError;
  if (o.bytes && fwrite(h.data(), 1, o.bytes, f) != o.bytes) { fclose(f); return ncclSystemError; }
  fclose(f);
  return rename(tmp.c_str(), p.c_str()) == 0 ? ncclSuccess : ncclSystemError;
}
ncclResult_t do_recv(const Op &o) {
  const std::string p = msg_path(o.c, o.peer, o.c->rank, o.c->recvd[o.peer]++);
  const auto t0 = std::chrono::steady_clock::now();
  struct stat st;
  while (stat(p.c_str(), &st) != 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "fake_rccl: rank %d timed out waiting for %s\n", o.c->rank, p.c_str()); return ncclSystemError; }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  if ((size_t)st.st_size != o.bytes) { fprintf(stderr, "fake_rccl: %s has %lld bytes, receiver expects %zu\n", p.c_str(), (long long)st.st_size, o.bytes); return ncclInvalidArgument; }
  std::vector<char> h(o.bytes);
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return ncclSystemError;
  if (o.bytes && fread(h.data(), 1, o.bytes, f) != o.bytes) { fclose(f); return ncclSystemError; }
  fclose(f); unlink(p.c_str());
  if (hipStreamSynchronize(o.s) != hipSuccess) return ncclUnhandledCudaError;
  if (o.bytes && hipMemcpy(o.rbuf, h.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
// ---- stream-ordered mode: file transfer from host callbacks (no HIP call inside them) ----
bool host_send(ncclComm *c, int peer, const void *data, size_t bytes) {
  const std::string p = msg_path(c, c->rank, peer, c->sent[peer]++), tmp = p + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  if (bytes && fwrite(data, 1, bytes, f) != bytes) { fclose(f); return false; }
  fclose(f);
  return rename(tmp.c_str(), p.c_str()) == 0;
}
bool host_recv(ncclComm *c, int peer, void *data, size_t bytes) {
  const std::string p = msg_path(c, peer, c->rank, c->recvd[peer]++);
  const auto t0 = std::chrono::steady_clock::now();
  struct stat st;
  while (stat(p.c_str(), &st) != 0) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "fake_rccl: rank %d timed out waiting for %s\n", c->rank, p.c_str()); return false; }
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  if ((size_t)st.st_size != bytes) { fprintf(stderr, "fake_rccl: %s has %lld bytes, receiver expects %zu\n", p.c_str(), (long long)st.st_size, bytes); return false; }
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return false;
  const bool ok = !bytes || fread(data, 1, bytes, f) == bytes;
  fclose(f); unlink(p.c_str());
  return ok;
}
struct Node { ncclComm *c; int peer; size_t bytes; void *stage; size_t count; };   // lives as long as the communicator's arena: graphs replay it
void *arena_take(ncclComm *c, size_t bytes) {
  const size_t a = (c->arena_used + 63) & ~(size_t)63;
  if (a + bytes > c->arena_cap) return nullptr;
  c->arena_used = a + bytes;
  return c->arena + a;
}
void cb_send(void *u) { Node *n = (Node *)u; if (!host_send(n->c, n->peer, n->stage, n->bytes)) fprintf(stderr, "fake_rccl: send failed (rank %d -> %d)\n", n->c->rank, n->peer); }
void cb_recv(void *u) { Node *n = (Node *)u; if (!host_recv(n->c, n->peer, n->stage, n->bytes)) { fprintf(stderr, "fake_rccl: receive failed (rank %d <- %d)\n", n->c->rank, n->peer); memset(n->stage, 0xff, n->bytes); } }
void cb_allreduce(void *u) {      // stage: [count doubles mine | count doubles scratch]; result (rank-ordered sum) back into the first part
  Node *n = (Node *)u; ncclComm *c = n->c;
  double *mine = (double *)n->stage, *tmp = mine + n->count;
  for (int p = 0; p < c->world; ++p) host_send(c, p, mine, n->bytes);
  std::vector<double> acc(n->count, 0.0);
  for (int p = 0; p < c->world; ++p) { if (!host_recv(c, p, tmp, n->bytes)) { fprintf(stderr, "fake_rccl: all-reduce failed\n"); return; } for (size_t i = 0; i < n->count; ++i) acc[i] += tmp[i]; }
  memcpy(mine, acc.data(), n->bytes);
}
ncclResult_t flush_stream(std::vector<Op> &ops) {
  for (const Op &o : ops) if (o.send) {
    Node *n = new Node{o.c, o.peer, o.bytes, arena_take(o.c, o.bytes), 0};
    if (!n->stage) { fprintf(stderr, "fake_rccl: staging arena exhausted\n"); return ncclSystemError; }
    if (o.bytes && hipMemcpyAsync(n->stage, o.sbuf, o.bytes, hipMemcpyDeviceToHost, o.s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(o.s, cb_send, n) != hipSuccess) return ncclUnhandledCudaError;
  }
  for (const Op &o : ops) if (!o.send) {
    Node *n = new Node{o.c, o.peer, o.bytes, arena_take(o.c, o.bytes), 0};
    if (!n->stage) { fprintf(stderr, "fake_rccl: staging arena exhausted\n"); return ncclSystemError; }
    if (hipLaunchHostFunc(o.s, cb_recv, n) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpyAsync(o.rbuf, n->stage, o.bytes, hipMemcpyHostToDevice, o.s) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}
ncclResult_t flush() {
  std::vector<Op> ops; ops.swap(g_ops);
  if (mgs_fake_rccl_marker == 2) return flush_stream(ops);
  for (const Op &o : ops) if (o.send) { ncclResult_t r = do_send(o); if (r != ncclSuccess) return r; }
  for (const Op &o : ops) if (!o.send) { ncclResult_t r = do_recv(o); if (r != ncclSuccess) return r; }
  return ncclSuccess;
}
size_t tsize(ncclDataType_t t) { return t == ncclDouble ? 8 : (t == ncclFloat || t == ncclInt32 || t == ncclUint32) ? 4 : (t == ncclInt64 || t == ncclUint64) ? 8 : 1; }
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "mgsfake_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  ncclComm *c = new ncclComm();
  id.internal[sizeof id.internal - 1] = 0;
  c->dir = std::string("/tmp/") + id.internal; c->world = nranks; c->rank = rank;
  c->sent.assign(nranks, 0); c->recvd.assign(nranks, 0);
  mkdir(c->dir.c_str(), 0700);
  if (mgs_fake_rccl_marker == 2) {
    c->arena_cap = (size_t)256 << 20;
    if (hipHostMalloc((void **)&c->arena, c->arena_cap, hipHostMallocDefault) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  }
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (comm) { hipDeviceSynchronize(); if (comm->arena) hipHostFree(comm->arena); rmdir(comm->dir.c_str()); delete comm; }
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "fake_rccl error"; }
ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--g_depth == 0) return flush(); return ncclSuccess; }
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  g_ops.push_back(Op{true, buf, nullptr, count * tsize(t), peer, c, s});
  return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  g_ops.push_back(Op{false, nullptr, buf, count * tsize(t), peer, c, s});
  return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
  const size_t b = count * tsize(t);
  ncclGroupStart();
  for (int p = 0; p < c->world; ++p) { ncclSend(send, count, t, p, c, s); ncclRecv((char *)recv + (size_t)p * b, count, t, p, c, s); }
  return ncclGroupEnd();
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (t != ncclDouble || op != ncclSum) return ncclInvalidArgument;
  if (mgs_fake_rccl_marker == 2) {      // stream-ordered: device → pinned, host function (files, rank-ordered sum), pinned → device
    Node *n = new Node{c, -1, sizeof(double) * count, arena_take(c, 2 * sizeof(double) * count), count};
    if (!n->stage) return ncclSystemError;
    if (hipMemcpyAsync(n->stage, send, n->bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(s, cb_allreduce, n) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, n->stage, n->bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
  }
  double *all = nullptr;
  if (hipMalloc((void **)&all, sizeof(double) * count * c->world) != hipSuccess) return ncclUnhandledCudaError;
  ncclResult_t r = ncclAllGather(send, all, count, t, c, s);
  if (r == ncclSuccess) {
    std::vector<double> h(count * c->world), o(count, 0.0);
    hipMemcpy(h.data(), all, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    for (int p = 0; p < c->world; ++p) for (size_t i = 0; i < count; ++i) o[i] += h[(size_t)p * count + i];
    hipMemcpy(recv, o.data(), sizeof(double) * count, hipMemcpyHostToDevice);
  }
  hipFree(all);
  return r;
}

}  // extern "C"
