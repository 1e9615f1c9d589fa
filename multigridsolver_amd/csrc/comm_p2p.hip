// comm_p2p.hip — peer-to-peer transport of the row-sharded cycle: one process per GPU, the neighbours' halo values are STORED
// straight into this rank's memory over xGMI by the neighbours' own kernels (no RCCL kernel, no proxy thread, no host round trip).
//
// Why it exists: the exchange step of the path moves 8 B … 2 MiB per neighbour, seven times per cycle; through RCCL each
// of them is a `rcclGenericKernel` of 12–16 µs that does not shrink with the rank count (profiles/r03_rank8_timeline.md) — 12 % of a
// rank's cycle at 8 GPUs.  Here ONE kernel per exchange does all of it:
//
//   phase 1  every workgroup copies its share of the ranges this rank owes its peers into the PEERS' windows (IPC-mapped device
//            memory, hipIpcGetMemHandle / hipIpcOpenMemHandle), each thread fences its stores at system scope, and the workgroup
//            that arrives last publishes this rank's sequence number in every participating peer's flag word (release, system scope);
//   phase 2  one lane per participating peer polls this rank's own flag word until that peer's sequence number shows up
//            (bounded: MGS_P2P_TIMEOUT_S, default 20 s — every wave reaches the exit; a timeout sets the error word the host checks);
//   phase 3  every workgroup copies what the peers stored from the window to where the cycle wants it (halo slots behind the owned
//            entries of a vector, a payload buffer, the tail's gather buffer).
//
// No acknowledgements are needed: a pair of ranks takes part in the same sequence of exchanges (participation is symmetric:
// p sends to q ⇔ q receives from p, and a pair that communicates in one direction signals in both), the window holds TWO slots per
// peer used alternately (sequence parity), and a rank can only be two exchanges ahead of a neighbour after that neighbour has
// signalled the exchange in between — which it does after unpacking the one before (stream order).  Sequence numbers live in
// device memory, so a captured cycle replays without any host-side state: kernel arguments are constants.
//
// Stream memory operations (hipStreamWaitValue64 / hipStreamWriteValue64) were the alternative asked for; on this stack they are
// kernels too (`__amd_rocclr_streamOpsWait` spins on a CU) unless the word is hipMallocSignalMemory — an HSA signal that cannot be
// exported to another process — and they cost one dispatch per operation (tools/microbench/p2p_probe.cpp measures both).
//
// The window is allocated uncached (hipDeviceMallocUncached → MTYPE_UC): a peer's stores arrive in HBM behind the XCD L2s' backs,
// so the window must never be served from a stale L2 line; coarse-grained memory is the fallback (one GPU shared by the ranks:
// tests) and is reported by mgs_comm_p2p_info.  Nothing here has a CPU fallback.
#include <unistd.h>

#include "mgs_internal.hpp"

namespace {
constexpr int P2P_TB = 256, P2P_MAXSEG = 112, P2P_MAXPEER = 8, P2P_FLAG_STRIDE = 128, P2P_FLAG_BYTES = 4096;
struct P2PSeg {
  const double *src; double *dst;
  unsigned n;
  short peer;   // index into the exchange's participating peers (parity of that pair's sequence number picks the window slot)
  short kind;   // 0 push: dst is slot 0 of the PEER's window area for this rank; 1 unpack: src is slot 0 of this rank's area for the peer; 2 local copy
};
struct P2PArgs {
  P2PSeg seg[P2P_MAXSEG];
  unsigned long long *flag_out[P2P_MAXPEER];   // peer's window: flag word this rank writes
  unsigned long long *flag_in[P2P_MAXPEER];    // own window: flag word the peer writes
  unsigned long long *seq[P2P_MAXPEER];        // own plain device memory: exchanges done with that peer
  unsigned *arrive, *err;
  unsigned long long slot_doubles, timeout_ticks;
  int nseg, npeer, rank;
  int tune;     // bits 0-1 release side (0 system fence, 1 agent fence, 2 wait for the stores' acknowledgement only), bit 2: no acquire fence.
                // Uncached windows run 6, cached ones 0 (see mgs_p2p_create); MGS_P2P_TUNE overrides for experiments
};

// grid-stride copy with four 16-byte (or 8-byte) loads per lane in flight: the window is uncached memory, every load pays the full trip
__device__ __forceinline__ void copy_range(const double *__restrict__ src, double *__restrict__ dst, unsigned n) {
  const unsigned stride = gridDim.x * P2P_TB, t = blockIdx.x * P2P_TB + threadIdx.x;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {       // 16 B per lane where both ends allow it
    typedef double vd2 __attribute__((ext_vector_type(2)));
    const unsigned n2 = n >> 1;
    const vd2 *s2 = reinterpret_cast<const vd2 *>(src); vd2 *d2 = reinterpret_cast<vd2 *>(dst);
    for (unsigned i = t; i < n2; i += 4 * stride) {
      const unsigned i1 = i + stride, i2 = i + 2 * stride, i3 = i + 3 * stride;
      const vd2 a = s2[i];
      vd2 b = a, c = a, d = a;
      if (i1 < n2) b = s2[i1];
      if (i2 < n2) c = s2[i2];
      if (i3 < n2) d = s2[i3];
      d2[i] = a;
      if (i1 < n2) d2[i1] = b;
      if (i2 < n2) d2[i2] = c;
      if (i3 < n2) d2[i3] = d;
    }
    if ((n & 1) && t == 0) dst[n - 1] = src[n - 1];
  } else {
    for (unsigned i = t; i < n; i += 4 * stride) {
      const unsigned i1 = i + stride, i2 = i + 2 * stride, i3 = i + 3 * stride;
      const double a = src[i];
      double b = a, c = a, d = a;
      if (i1 < n) b = src[i1];
      if (i2 < n) c = src[i2];
      if (i3 < n) d = src[i3];
      dst[i] = a;
      if (i1 < n) dst[i1] = b;
      if (i2 < n) dst[i2] = c;
      if (i3 < n) dst[i3] = d;
    }
  }
}

__global__ __launch_bounds__(P2P_TB) void p2p_exchange_kernel(const P2PArgs a) {
  __shared__ unsigned long long s_seq[P2P_MAXPEER];
  __shared__ int s_last, s_bad, s_dead;
  const int tid = threadIdx.x;
  if (tid < a.npeer) s_seq[tid] = *a.seq[tid];
  if (tid == 0) { s_last = 0; s_bad = 0; s_dead = *a.err != 0; }
  __syncthreads();
  if (s_dead) return;                            // a wait has timed out before: the transport is dead, every later exchange returns at once (workgroup-uniform; the host reads the word)
  // ---- phase 1: my ranges into the peers' windows (+ local copies)
  for (int q = 0; q < a.nseg; ++q) {
    const P2PSeg sg = a.seg[q];
    if (sg.kind == 1) continue;
    copy_range(sg.src, sg.dst + (sg.kind == 0 ? (s_seq[sg.peer] & 1ull) * a.slot_doubles : 0ull), sg.n);
  }
  // every thread: its stores are performed system-wide before the workgroup arrives
  if ((a.tune & 3) == 0) __threadfence_system();
  else if ((a.tune & 3) == 1) __threadfence();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_last = gridDim.x == 1 || atomicAdd(a.arrive, 1u) == gridDim.x - 1;      // (a single workgroup has nobody to wait for)
  __syncthreads();
  if (s_last) {                                 // everybody's ranges are out (and every workgroup has read the sequence numbers): publish
    if ((a.tune & 3) == 0) __threadfence_system(); else if ((a.tune & 3) == 1) __threadfence();
    if (tid < a.npeer) {
      __hip_atomic_store(a.flag_out[tid], s_seq[tid] + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      *a.seq[tid] = s_seq[tid] + 1ull;
    }
    if (tid == 0 && gridDim.x > 1) *a.arrive = 0u;
  }
  // ---- phase 2: the peers' ranges of THIS exchange are in my window once their sequence numbers are
  if (tid < a.npeer) {
    const unsigned long long want = s_seq[tid] + 1ull, t0 = wall_clock64();
    while (__hip_atomic_load(a.flag_in[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > a.timeout_ticks) { atomicExch(a.err, 1u + (unsigned)tid); s_bad = 1; break; }
    }
  }
  __syncthreads();
  if (!(a.tune & 4)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // every wave: nothing of the window is served from a line fetched before the flag
  if (s_bad) return;
  // ---- phase 3: window → destination
  for (int q = 0; q < a.nseg; ++q) {
    const P2PSeg sg = a.seg[q];
    if (sg.kind != 1) continue;
    copy_range(sg.src + (s_seq[sg.peer] & 1ull) * a.slot_doubles, sg.dst, sg.n);
  }
}

// sum of `cnt` doubles over the ranks: every rank stores its values into every peer's window, then adds the world's values in RANK
// ORDER — the same bits on every rank, whatever the arrival order (one workgroup; cnt ≤ 64)
struct P2PRedArgs {
  double *peer_area[P2P_MAXPEER];              // peer's window area for this rank, slot 0
  const double *my_area[P2P_MAXPEER];          // own window area for that peer, slot 0
  unsigned long long *flag_out[P2P_MAXPEER], *flag_in[P2P_MAXPEER], *seq[P2P_MAXPEER];
  short peer_rank[P2P_MAXPEER];
  unsigned *err;
  double *buf;
  unsigned long long slot_doubles, timeout_ticks;
  int npeer, rank, cnt, tune;
};
__global__ __launch_bounds__(64) void p2p_allreduce_kernel(const P2PRedArgs a) {
  __shared__ unsigned long long s_seq[P2P_MAXPEER];
  __shared__ int s_bad;
  __shared__ int s_dead;
  const int tid = threadIdx.x;
  if (tid < a.npeer) s_seq[tid] = *a.seq[tid];
  if (tid == 0) { s_bad = 0; s_dead = *a.err != 0; }
  __syncthreads();
  if (s_dead) return;
  const double mine = tid < a.cnt ? a.buf[tid] : 0.0;
  if (tid < a.cnt) for (int p = 0; p < a.npeer; ++p) a.peer_area[p][(s_seq[p] & 1ull) * a.slot_doubles + tid] = mine;
  if ((a.tune & 3) == 0) __threadfence_system(); else if ((a.tune & 3) == 1) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < a.npeer) {
    __hip_atomic_store(a.flag_out[tid], s_seq[tid] + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    *a.seq[tid] = s_seq[tid] + 1ull;
    const unsigned long long want = s_seq[tid] + 1ull, t0 = wall_clock64();
    while (__hip_atomic_load(a.flag_in[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > a.timeout_ticks) { atomicExch(a.err, 1u + (unsigned)tid); s_bad = 1; break; }
    }
  }
  __syncthreads();
  if (!(a.tune & 4)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (s_bad || tid >= a.cnt) return;
  double s = 0.0;
  int p = 0;
  for (; p < a.npeer && a.peer_rank[p] < a.rank; ++p) s += a.my_area[p][(s_seq[p] & 1ull) * a.slot_doubles + tid];
  s += mine;
  for (; p < a.npeer; ++p) s += a.my_area[p][(s_seq[p] & 1ull) * a.slot_doubles + tid];
  a.buf[tid] = s;
}
}  // namespace

struct mgs_p2p {
  mgs_ctx *ctx = nullptr;
  int world = 0, rank = 0;
  size_t slot_doubles = 0, win_bytes = 0;
  char *win = nullptr;                       // own window: flag words, then two slots per peer
  int mem_kind = 0;                          // 1 uncached, 2 fine-grained, 3 coarse-grained (hipMalloc)
  std::vector<char *> peer_win;              // peers' windows as this process maps them (own rank: win)
  std::vector<bool> opened;
  unsigned long long *seq = nullptr;         // world counters (plain device memory)
  unsigned *arrive = nullptr, *err = nullptr;
  unsigned long long timeout_ticks = 0;
  unsigned long long launches = 0;
  int tune = 0; size_t block_doubles = 512;
  bool connected = false;
  unsigned long long *flag_of(char *w, int writer) const { return reinterpret_cast<unsigned long long *>(w + (size_t)writer * P2P_FLAG_STRIDE); }
  double *area_of(char *w, int writer) const { return reinterpret_cast<double *>(w + P2P_FLAG_BYTES + (size_t)writer * 2 * slot_doubles * sizeof(double)); }
};

struct P2PHandle { hipIpcMemHandle_t h; long long pid; unsigned long long bytes; };
static_assert(sizeof(P2PHandle) <= MGS_P2P_HANDLE_BYTES, "handle record size");

int mgs_p2p_create(mgs_ctx *ctx, int world, int rank, size_t slot_doubles, void *handle_out, mgs_p2p **out) {
  MGS_CHECK(ctx, out && handle_out && world >= 1 && world <= P2P_MAXPEER && rank >= 0 && rank < world && slot_doubles >= 64, MGS_ERR_INVALID,
            "mgs_comm_p2p_create: bad arguments (world 1..%d, slot >= 64 doubles)", P2P_MAXPEER);
  static_assert(P2P_MAXPEER * P2P_FLAG_STRIDE <= P2P_FLAG_BYTES, "flag words fit their page");
  mgs_p2p *c = new mgs_p2p();
  c->ctx = ctx; c->world = world; c->rank = rank;
  c->slot_doubles = (slot_doubles + 1) & ~(size_t)1;           // slots stay 16-byte aligned
  c->win_bytes = P2P_FLAG_BYTES + (size_t)world * 2 * c->slot_doubles * sizeof(double);
  hipSetDevice(ctx->device);
  const char *force = getenv("MGS_P2P_MEM");                    // A/B only: "fine" = fine-grained, "coarse" = plain device memory
  void *p = nullptr;
  if (!(force && !strcmp(force, "coarse"))) {
    if (!(force && !strcmp(force, "fine")) && hipExtMallocWithFlags(&p, c->win_bytes, hipDeviceMallocUncached) == hipSuccess) c->mem_kind = 1;
    else { (void)hipGetLastError(); if (hipExtMallocWithFlags(&p, c->win_bytes, hipDeviceMallocFinegrained) == hipSuccess) c->mem_kind = 2; else (void)hipGetLastError(); }
  }
  if (!p) {
    if (hipExtMallocWithFlags(&p, c->win_bytes, hipDeviceMallocDefault) != hipSuccess) { (void)hipGetLastError(); delete c; return mgs_fail(ctx, MGS_ERR_ALLOC, "p2p window of %zu bytes: allocation failed", c->win_bytes); }
    c->mem_kind = 3;
  }
  c->win = (char *)p;
  int rc = MGS_OK;
  if (hipMemset(c->win, 0, c->win_bytes) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "p2p window: memset failed");
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &c->seq, (size_t)world);
  if (rc == MGS_OK) rc = mgs_dev_alloc(ctx, &c->arrive, 2);
  if (rc == MGS_OK) { c->err = c->arrive + 1; if (hipMemset(c->seq, 0, sizeof(unsigned long long) * (size_t)world) != hipSuccess || hipMemset(c->arrive, 0, 2 * sizeof(unsigned)) != hipSuccess) rc = mgs_fail(ctx, MGS_ERR_HIP, "p2p state: memset failed"); }
  P2PHandle H; memset(&H, 0, sizeof H);
  if (rc == MGS_OK && world > 1 && hipIpcGetMemHandle(&H.h, c->win) != hipSuccess) { (void)hipGetLastError(); rc = mgs_fail(ctx, MGS_ERR_HIP, "hipIpcGetMemHandle on the p2p window failed (memory kind %d)", c->mem_kind); }
  if (rc != MGS_OK) { mgs_p2p_destroy(c); return rc; }
  H.pid = (long long)getpid(); H.bytes = c->win_bytes;
  memset(handle_out, 0, MGS_P2P_HANDLE_BYTES); memcpy(handle_out, &H, sizeof H);
  const char *e = getenv("MGS_P2P_TIMEOUT_S");
  const double secs = e && atof(e) > 0.0 ? atof(e) : 20.0;
  c->timeout_ticks = (unsigned long long)(secs * 100e6);       // wall_clock64(): constant 100 MHz
  // Uncached window (the default): its stores bypass every cache, so "performed" is the store's own acknowledgement (s_waitcnt vmcnt(0)) and its
  // loads can never be served from a stale line — no L2 write-back / invalidate on either side.  Measured (tools/microbench/p2p_probe.cpp,
  // profiles/r04_p2p_probe.md): 4.4 / 5.8 / 6.4 / 9.5 µs per exchange of 8 B / 32 KiB / 256 KiB / 2 MiB per neighbour against 5.9 / 7.8 / 17 / 30 µs
  // with system-scope fences; the same shortcut on a cached window gives WRONG data between two processes (probe, "coarse"/"fine" rows),
  // so cached windows keep the fences.
  c->tune = c->mem_kind == 1 ? 6 : 0;
  if (const char *t = getenv("MGS_P2P_TUNE")) c->tune = atoi(t);
  if (const char *t = getenv("MGS_P2P_BLOCK_DOUBLES")) c->block_doubles = (size_t)std::max(atoi(t), 64);
  c->peer_win.assign((size_t)world, nullptr); c->opened.assign((size_t)world, false);
  c->peer_win[(size_t)rank] = c->win;
  c->connected = world == 1;
  *out = c;
  return MGS_OK;
}

int mgs_p2p_connect(mgs_p2p *c, const void *handles) {
  mgs_ctx *ctx = c->ctx;
  MGS_CHECK(ctx, handles, MGS_ERR_INVALID, "mgs_comm_p2p_connect: NULL handles");
  hipSetDevice(ctx->device);
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank || c->peer_win[(size_t)p]) continue;
    P2PHandle H; memcpy(&H, (const char *)handles + (size_t)p * MGS_P2P_HANDLE_BYTES, sizeof H);
    MGS_CHECK(ctx, H.bytes == c->win_bytes, MGS_ERR_INVALID, "p2p: rank %d has a window of %llu bytes, this rank %zu", p, H.bytes, c->win_bytes);
    MGS_CHECK(ctx, H.pid != (long long)getpid(), MGS_ERR_INVALID, "p2p: rank %d lives in this process (one process per rank)", p);
    void *q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, H.h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return mgs_fail(ctx, MGS_ERR_HIP, "hipIpcOpenMemHandle(rank %d): %s", p, hipGetErrorString(e)); }
    c->peer_win[(size_t)p] = (char *)q; c->opened[(size_t)p] = true;
  }
  c->connected = true;
  return MGS_OK;
}

void mgs_p2p_destroy(mgs_p2p *c) {
  if (!c) return;
  if (c->ctx) { hipSetDevice(c->ctx->device); (void)hipDeviceSynchronize(); }
  for (size_t p = 0; p < c->peer_win.size(); ++p) if (c->opened[p] && c->peer_win[p]) (void)hipIpcCloseMemHandle(c->peer_win[p]);
  if (c->seq) mgs_hip_free(c->seq);
  if (c->arrive) mgs_hip_free(c->arrive);
  if (c->win) (void)hipFree(c->win);      // the window is an allocation of its own (never from the arena): IPC handles name whole allocations
  delete c;
}

int mgs_p2p_info(const mgs_p2p *c, long long out[6]) {
  unsigned e = 0;
  if (hipMemcpy(&e, c->err, sizeof e, hipMemcpyDeviceToHost) != hipSuccess) return mgs_fail(c->ctx, MGS_ERR_HIP, "p2p: reading the error word failed");
  out[0] = c->mem_kind; out[1] = (long long)c->win_bytes; out[2] = (long long)c->slot_doubles; out[3] = (long long)c->launches; out[4] = (long long)e; out[5] = c->connected ? 1 : 0;
  return MGS_OK;
}

// workgroups of one exchange: one 16-byte load per lane up to 1 MiB per phase, four beyond (copy_range keeps four in flight); a message
// of a few hundred doubles is one workgroup and skips the arrival counter
// ---- self-test: `rounds` exchanges with every peer, every value a function of (round, sender, receiver, position), verified ON THE DEVICE
__global__ void p2p_fill_kernel(double *buf, unsigned n, int round, int from, int world) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x, tot = n * (unsigned)world;
  if (i < tot) { const int to = (int)(i / n); buf[i] = (double)round * 1048576.0 + (double)from * 65536.0 + (double)to * 4096.0 + (double)(i % n) * 0.5; }
}
__global__ void p2p_check_kernel(const double *buf, unsigned n, int round, int me, int world, unsigned *bad) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x, tot = n * (unsigned)world;
  if (i < tot) {
    const int from = (int)(i / n);
    if (from == me) return;
    const double want = (double)round * 1048576.0 + (double)from * 65536.0 + (double)me * 4096.0 + (double)(i % n) * 0.5;
    if (buf[i] != want) atomicAdd(bad, 1u);
  }
}

static int grid_for(size_t doubles, size_t per_block = (size_t)P2P_TB * 2) {
  const size_t nb = (doubles + per_block - 1) / per_block;
  return (int)std::min<size_t>(std::max<size_t>(nb, 1), 256);
}

// the ops of one RCCL-style group (see mgs_comm_exchange_ops): k-th send to p ↔ p's k-th receive from this rank, in posting order
int mgs_p2p_exchange_ops(mgs_p2p *c, hipStream_t s, const mgs_xfer_op *ops, int nops) {
  mgs_ctx *ctx = c->ctx;
  MGS_CHECK(ctx, c->connected, MGS_ERR_STATE, "p2p transport used before mgs_comm_p2p_connect");
  P2PArgs a; a.nseg = 0; a.npeer = 0; a.rank = c->rank;
  int slot_of[P2P_MAXPEER]; size_t soff[P2P_MAXPEER], roff[P2P_MAXPEER];
  for (int p = 0; p < c->world; ++p) { slot_of[p] = -1; soff[p] = roff[p] = 0; }
  size_t most = 0, pushed = 0, unpacked = 0;
  for (int q = 0; q < nops; ++q) {
    const mgs_xfer_op &o = ops[q];
    if (!o.count) continue;
    MGS_CHECK(ctx, o.peer >= 0 && o.peer < c->world, MGS_ERR_INVALID, "p2p exchange: peer %d out of range", o.peer);
    if (slot_of[o.peer] < 0) {
      const int k = a.npeer++;
      slot_of[o.peer] = k;
      a.flag_out[k] = c->flag_of(c->peer_win[(size_t)o.peer], c->rank);
      a.flag_in[k] = c->flag_of(c->win, o.peer);
      a.seq[k] = c->seq + o.peer;
    }
    MGS_CHECK(ctx, a.nseg < P2P_MAXSEG, MGS_ERR_INVALID, "p2p exchange: more than %d ranges in one group", P2P_MAXSEG);
    P2PSeg &sg = a.seg[a.nseg++];
    sg.n = (unsigned)o.count; sg.peer = (short)slot_of[o.peer];
    if (o.sptr) {
      MGS_CHECK(ctx, soff[o.peer] + o.count <= c->slot_doubles, MGS_ERR_INVALID, "p2p exchange: %zu doubles for rank %d exceed the window slot (%zu)", soff[o.peer] + o.count, o.peer, c->slot_doubles);
      sg.kind = 0; sg.src = o.sptr; sg.dst = c->area_of(c->peer_win[(size_t)o.peer], c->rank) + soff[o.peer];
      soff[o.peer] += o.count; pushed += o.count;
    } else {
      MGS_CHECK(ctx, roff[o.peer] + o.count <= c->slot_doubles, MGS_ERR_INVALID, "p2p exchange: %zu doubles from rank %d exceed the window slot (%zu)", roff[o.peer] + o.count, o.peer, c->slot_doubles);
      sg.kind = 1; sg.src = c->area_of(c->win, o.peer) + roff[o.peer]; sg.dst = o.rptr;
      roff[o.peer] += o.count; unpacked += o.count;
    }
  }
  if (!a.npeer) return MGS_OK;
  most = std::max(pushed, unpacked);
  a.arrive = c->arrive; a.err = c->err; a.slot_doubles = c->slot_doubles; a.timeout_ticks = c->timeout_ticks; a.tune = c->tune;
  hipLaunchKernelGGL(p2p_exchange_kernel, dim3(grid_for(most, c->block_doubles)), dim3(P2P_TB), 0, s, a);
  MGS_HIP(ctx, hipGetLastError());
  ++c->launches;
  return MGS_OK;
}

// COLLECTIVE: every rank sends every peer n values per round and checks what it received, rounds of growing size; *mismatches = wrong values seen here
int mgs_p2p_selftest(mgs_p2p *c, hipStream_t s, int rounds, long long *mismatches) {
  mgs_ctx *ctx = c->ctx;
  MGS_CHECK(ctx, c->connected && mismatches && rounds >= 1, MGS_ERR_INVALID, "mgs_comm_p2p_selftest: bad arguments / not connected");
  *mismatches = 0;
  if (c->world == 1) return MGS_OK;
  const size_t nmax = std::min<size_t>(c->slot_doubles, (size_t)1 << 16);
  double *src = nullptr, *dst = nullptr; unsigned *bad = nullptr;
  MGS_TRY(mgs_dev_alloc(ctx, &src, nmax * (size_t)c->world)); MGS_TRY(mgs_dev_alloc(ctx, &dst, nmax * (size_t)c->world)); MGS_TRY(mgs_dev_alloc(ctx, &bad, 1));
  MGS_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(unsigned), s));
  int rc = MGS_OK;
  for (int r = 0; r < rounds && rc == MGS_OK; ++r) {
    const size_t n = std::max<size_t>(1, (nmax >> (r % 12)));            // 64 Ki doubles down to a handful, again and again
    const unsigned tot = (unsigned)(n * (size_t)c->world);
    hipLaunchKernelGGL(p2p_fill_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, src, (unsigned)n, r, c->rank, c->world);
    std::vector<mgs_xfer_op> ops;
    for (int p = 0; p < c->world; ++p) if (p != c->rank) { ops.push_back({src + (size_t)p * n, nullptr, n, p}); ops.push_back({nullptr, dst + (size_t)p * n, n, p}); }
    rc = mgs_p2p_exchange_ops(c, s, ops.data(), (int)ops.size());
    hipLaunchKernelGGL(p2p_check_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, dst, (unsigned)n, r, c->rank, c->world, bad);
  }
  unsigned hb = 0;
  if (rc == MGS_OK && (hipMemcpyAsync(&hb, bad, sizeof hb, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = mgs_fail(ctx, MGS_ERR_HIP, "p2p self-test: reading the result failed");
  else (void)hipStreamSynchronize(s);
  mgs_hip_free(src); mgs_hip_free(dst); mgs_hip_free(bad);
  *mismatches = (long long)hb;
  return rc;
}

int mgs_p2p_allgather(mgs_p2p *c, hipStream_t s, const double *send, double *recv, size_t count) {
  mgs_ctx *ctx = c->ctx;
  if (c->world == 1) { if (send != recv) MGS_HIP(ctx, hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, s)); return MGS_OK; }
  // own slice: a local copy inside the same launch
  MGS_CHECK(ctx, c->connected, MGS_ERR_STATE, "p2p transport used before mgs_comm_p2p_connect");
  P2PArgs a; a.nseg = 0; a.npeer = 0; a.rank = c->rank;
  MGS_CHECK(ctx, count <= c->slot_doubles, MGS_ERR_INVALID, "p2p all-gather: %zu doubles exceed the window slot (%zu)", count, c->slot_doubles);
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    const int k = a.npeer++;
    a.flag_out[k] = c->flag_of(c->peer_win[(size_t)p], c->rank); a.flag_in[k] = c->flag_of(c->win, p); a.seq[k] = c->seq + p;
    a.seg[a.nseg++] = P2PSeg{send, c->area_of(c->peer_win[(size_t)p], c->rank), (unsigned)count, (short)k, 0};
    a.seg[a.nseg++] = P2PSeg{c->area_of(c->win, p), recv + (size_t)p * count, (unsigned)count, (short)k, 1};
  }
  if (send != recv + (size_t)c->rank * count) a.seg[a.nseg++] = P2PSeg{send, recv + (size_t)c->rank * count, (unsigned)count, 0, 2};
  a.arrive = c->arrive; a.err = c->err; a.slot_doubles = c->slot_doubles; a.timeout_ticks = c->timeout_ticks; a.tune = c->tune;
  hipLaunchKernelGGL(p2p_exchange_kernel, dim3(grid_for(count * (size_t)(c->world - 1), c->block_doubles)), dim3(P2P_TB), 0, s, a);
  MGS_HIP(ctx, hipGetLastError());
  ++c->launches;
  return MGS_OK;
}

int mgs_p2p_allreduce_sum(mgs_p2p *c, hipStream_t s, double *buf, size_t count) {
  mgs_ctx *ctx = c->ctx;
  if (c->world == 1 || count == 0) return MGS_OK;
  MGS_CHECK(ctx, c->connected, MGS_ERR_STATE, "p2p transport used before mgs_comm_p2p_connect");
  MGS_CHECK(ctx, count <= 64 && count <= c->slot_doubles, MGS_ERR_INVALID, "p2p all-reduce: %zu values (at most 64)", count);
  P2PRedArgs a; a.npeer = 0; a.rank = c->rank; a.cnt = (int)count; a.buf = buf;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    const int k = a.npeer++;
    a.peer_area[k] = c->area_of(c->peer_win[(size_t)p], c->rank); a.my_area[k] = c->area_of(c->win, p);
    a.flag_out[k] = c->flag_of(c->peer_win[(size_t)p], c->rank); a.flag_in[k] = c->flag_of(c->win, p); a.seq[k] = c->seq + p;
    a.peer_rank[k] = (short)p;
  }
  a.err = c->err; a.slot_doubles = c->slot_doubles; a.timeout_ticks = c->timeout_ticks; a.tune = c->tune;
  hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(64), 0, s, a);
  MGS_HIP(ctx, hipGetLastError());
  ++c->launches;
  return MGS_OK;
}
