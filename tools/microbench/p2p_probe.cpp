// p2p_probe.cpp — W processes sharing ONE GPU drive the library's peer-to-peer transport (csrc/comm_p2p.hip) through the C-ABI:
//   1. ring exchange (each rank sends `n` doubles to both neighbours, receives theirs), verified and timed eagerly and replayed from a
//      hipGraph of 10 exchanges, for n = 1, 4096 (32 KiB), 262144 (2 MiB);
//   2. all-gather and all-reduce, verified;
//   3. rank 0 alone: what a stream memory operation costs on this stack (hipStreamWriteValue64 / hipStreamWaitValue64 on plain device
//      memory and on hipMallocSignalMemory), and whether it can be captured into a graph.
// build: hipcc --offload-arch=gfx950 -O2 -I include tools/microbench/p2p_probe.cpp -L multigridsolver_amd -lmgs -Wl,-rpath,$PWD/multigridsolver_amd -o /tmp/p2p_probe
// run:   /tmp/p2p_probe [world=2]            (forks world-1 children BEFORE any HIP call)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mgs.h"

struct Shared {
  volatile int arrived[16];     // barrier generations per rank
  volatile int bad;
  char handles[8][MGS_P2P_HANDLE_BYTES];
};
static Shared *S;
static int W = 2, R = 0;
static void barrier(int gen) {
  __sync_synchronize(); S->arrived[R] = gen; __sync_synchronize();
  const auto t0 = std::chrono::steady_clock::now();
  for (int p = 0; p < W; ++p)
    while (S->arrived[p] < gen) {
      if (S->bad || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { fprintf(stderr, "rank %d: barrier %d gave up\n", R, gen); _exit(3); }
      usleep(50);
    }
}
#define CK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "rank %d: %s -> %d: %s\n", R, #call, rc_, mgs_last_error(ctx)); S->bad = 1; _exit(2); } } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", R, #call, hipGetErrorString(e_)); S->bad = 1; _exit(2); } } while (0)

int main(int argc, char **argv) {
  W = argc > 1 ? atoi(argv[1]) : 2;
  if (W < 1 || W > 8) return 1;
  S = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset((void *)S, 0, sizeof(Shared));
  std::vector<pid_t> kids;
  for (int r = 1; r < W; ++r) { pid_t p = fork(); if (p == 0) { R = r; break; } kids.push_back(p); }
  mgs_ctx *ctx = nullptr;
  CK(mgs_ctx_create(0, nullptr, &ctx));
  hipStream_t st = (hipStream_t)mgs_ctx_stream(ctx);
  const size_t slot = (size_t)2 * 262144 + 64;
  mgs_comm *c = nullptr;
  CK(mgs_comm_p2p_create(ctx, W, R, slot, (void *)S->handles[R], &c));
  barrier(1);
  CK(mgs_comm_p2p_connect(c, (const void *)S->handles));
  barrier(2);
  long long info[6];
  CK(mgs_comm_p2p_info(c, info));
  if (R == 0) printf("world %d on one GPU; window %lld bytes, memory kind %lld (1 uncached, 2 fine-grained, 3 coarse); MGS_P2P_TUNE=%s MGS_P2P_BLOCK_DOUBLES=%s\n", W, info[1], info[0],
                     getenv("MGS_P2P_TUNE") ? getenv("MGS_P2P_TUNE") : "-", getenv("MGS_P2P_BLOCK_DOUBLES") ? getenv("MGS_P2P_BLOCK_DOUBLES") : "-");
  const bool brief = getenv("P2P_PROBE_BRIEF") != nullptr;
  const int lo = (R + W - 1) % W, hi = (R + 1) % W;
  int gen = 2;
  for (size_t n : {(size_t)1, (size_t)4096, (size_t)32768, (size_t)262144}) {
    mgs_vec *src = nullptr, *dst = nullptr;
    CK(mgs_vec_create(ctx, (int64_t)n * 2, &src)); CK(mgs_vec_create(ctx, (int64_t)n * 2, &dst));
    std::vector<double> h(2 * n), g(2 * n);
    double *sp = (double *)mgs_vec_ptr(src), *dp = (double *)mgs_vec_ptr(dst);
    int peer[4] = {lo, hi, lo, hi};
    size_t cnt[4] = {n, n, n, n};
    const void *sptr[4] = {sp, sp + n, nullptr, nullptr};
    void *rptr[4] = {nullptr, nullptr, dp, dp + n};
    if (W == 1) { /* self: both messages to rank 0, matched in posting order */ }
    double worst = 0.0;
    for (int it = 0; it < 6; ++it) {                   // both window slots, several times
      for (size_t i = 0; i < 2 * n; ++i) h[i] = 1000.0 * R + it + 1e-6 * (double)i;
      CK(mgs_vec_upload(src, h.data(), (int64_t)(2 * n))); CK(mgs_vec_fill(dst, -1.0));
      CK(mgs_comm_exchange_raw(c, 4, peer, cnt, sptr, rptr));
      CK(mgs_sync(ctx));
      CK(mgs_vec_download(dst, g.data(), (int64_t)(2 * n)));
      // dst[0..n) = what `lo` sent to its `hi` (= me): its second half; dst[n..2n) = what `hi` sent to its `lo`: its first half.  With W == 2 (lo == hi) the
      // two messages from the one peer arrive in its posting order: first its "to lo" (first half), then its "to hi" (second half); W == 1 alike.
      for (size_t i = 0; i < n; ++i) {
        const double e0 = (W <= 2) ? 1000.0 * lo + it + 1e-6 * (double)i : 1000.0 * lo + it + 1e-6 * (double)(n + i);
        const double e1 = (W <= 2) ? 1000.0 * hi + it + 1e-6 * (double)(n + i) : 1000.0 * hi + it + 1e-6 * (double)i;
        worst = fmax(worst, fmax(fabs(g[i] - e0), fabs(g[n + i] - e1)));
      }
    }
    if (worst != 0.0) { fprintf(stderr, "rank %d: exchange of %zu doubles WRONG (max deviation %g)\n", R, n, worst); S->bad = 1; _exit(4); }
    barrier(++gen);
    // eager timing
    hipEvent_t e0, e1; HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    const int reps = 200;
    for (int i = 0; i < 10; ++i) CK(mgs_comm_exchange_raw(c, 4, peer, cnt, sptr, rptr));
    CK(mgs_sync(ctx)); barrier(++gen);
    HK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(mgs_comm_exchange_raw(c, 4, peer, cnt, sptr, rptr));
    HK(hipEventRecord(e1, st)); HK(hipEventSynchronize(e1));
    float ms_eager = 0; HK(hipEventElapsedTime(&ms_eager, e0, e1));
    barrier(++gen);
    // the same from a graph of 10 exchanges
    hipGraph_t gr = nullptr; hipGraphExec_t ex = nullptr;
    HK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 10; ++i) CK(mgs_comm_exchange_raw(c, 4, peer, cnt, sptr, rptr));
    HK(hipStreamEndCapture(st, &gr)); HK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) HK(hipGraphLaunch(ex, st));
    HK(hipStreamSynchronize(st)); barrier(++gen);
    HK(hipEventRecord(e0, st));
    for (int i = 0; i < reps / 10; ++i) HK(hipGraphLaunch(ex, st));
    HK(hipEventRecord(e1, st)); HK(hipEventSynchronize(e1));
    float ms_graph = 0; HK(hipEventElapsedTime(&ms_graph, e0, e1));
    CK(mgs_vec_download(dst, g.data(), (int64_t)(2 * n)));
    barrier(++gen);
    if (R == 0) printf("exchange with both neighbours, %7zu doubles each way: eager %.2f us, graph replay %.2f us per exchange (results exact)\n", n, 1e3 * ms_eager / reps, 1e3 * ms_graph / reps);
    hipGraphExecDestroy(ex); hipGraphDestroy(gr); hipEventDestroy(e0); hipEventDestroy(e1);
    mgs_vec_destroy(src); mgs_vec_destroy(dst);
  }
  {   // all-gather, all-reduce
    const size_t n = 1000;
    mgs_vec *s = nullptr, *all = nullptr, *red = nullptr;
    CK(mgs_vec_create(ctx, (int64_t)n, &s)); CK(mgs_vec_create(ctx, (int64_t)(n * W), &all)); CK(mgs_vec_create(ctx, 8, &red));
    std::vector<double> h(n), g(n * W), r8(8);
    for (int it = 0; it < 4; ++it) {
      for (size_t i = 0; i < n; ++i) h[i] = R * 10.0 + it + 1e-3 * (double)i;
      CK(mgs_vec_upload(s, h.data(), (int64_t)n));
      CK(mgs_comm_allgather_raw(c, mgs_vec_ptr(s), mgs_vec_ptr(all), n));
      for (int q = 0; q < 8; ++q) r8[q] = (R + 1) * 0.1 + q + it;
      CK(mgs_vec_upload(red, r8.data(), 8));
      CK(mgs_comm_allreduce_raw(c, mgs_vec_ptr(red), 5));
      CK(mgs_sync(ctx));
      CK(mgs_vec_download(all, g.data(), (int64_t)(n * W))); CK(mgs_vec_download(red, r8.data(), 8));
      for (int p = 0; p < W; ++p) for (size_t i = 0; i < n; ++i) if (g[p * n + i] != p * 10.0 + it + 1e-3 * (double)i) { fprintf(stderr, "rank %d: all-gather WRONG at rank %d entry %zu\n", R, p, i); S->bad = 1; _exit(5); }
      for (int q = 0; q < 5; ++q) { double e = 0.0; for (int p = 0; p < W; ++p) e += (p + 1) * 0.1 + q + it; if (r8[q] != e) { fprintf(stderr, "rank %d: all-reduce WRONG: %.17g vs %.17g\n", R, r8[q], e); S->bad = 1; _exit(6); } }
      for (int q = 5; q < 8; ++q) if (r8[q] != (R + 1) * 0.1 + q + it) { fprintf(stderr, "rank %d: all-reduce touched entry %d\n", R, q); S->bad = 1; _exit(6); }
    }
    barrier(++gen);
    hipEvent_t e0, e1; HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    HK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; ++i) CK(mgs_comm_allreduce_raw(c, mgs_vec_ptr(red), 5));
    HK(hipEventRecord(e1, st)); HK(hipEventSynchronize(e1));
    float ms = 0; HK(hipEventElapsedTime(&ms, e0, e1));
    barrier(++gen);
    if (R == 0) printf("all-gather and all-reduce exact on every rank; all-reduce of 5 doubles %.2f us (eager)\n", 1e3 * ms / 200);
    mgs_vec_destroy(s); mgs_vec_destroy(all); mgs_vec_destroy(red);
  }
  {   // the collective self-test the launcher runs before it trusts the transport
    long long bad = -1;
    const auto t0 = std::chrono::steady_clock::now();
    const int rounds = getenv("P2P_PROBE_SELFTEST_ROUNDS") ? atoi(getenv("P2P_PROBE_SELFTEST_ROUNDS")) : 240;
    CK(mgs_comm_p2p_selftest(c, rounds, &bad));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (bad != 0) { fprintf(stderr, "rank %d: self-test saw %lld wrong values\n", R, bad); S->bad = 1; _exit(8); }
    barrier(++gen);
    if (R == 0) printf("self-test: %d pattern exchanges with every peer, 0 wrong values, %.1f ms\n", rounds, ms);
  }
  CK(mgs_comm_p2p_info(c, info));
  if (info[4] != 0) { fprintf(stderr, "rank %d: error word %lld\n", R, info[4]); S->bad = 1; _exit(7); }
  barrier(++gen);
  if (R == 0 && !brief) {   // stream memory operations on this stack
    unsigned long long *plain = nullptr, *sig = nullptr;
    HK(hipMalloc((void **)&plain, 64)); HK(hipMemset(plain, 0, 64));
    const hipError_t es = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
    if (es != hipSuccess) { (void)hipGetLastError(); sig = nullptr; printf("hipMallocSignalMemory: %s\n", hipGetErrorString(es)); }
    hipEvent_t e0, e1; HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
      unsigned long long *w = which ? sig : plain;
      if (!w) continue;
      hipError_t e = hipStreamWriteValue64(st, w, 1, 0);
      if (e == hipSuccess) e = hipStreamWaitValue64(st, w, 1, hipStreamWaitValueEq, ~0ull);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) { (void)hipGetLastError(); printf("stream memory operations on %s memory: %s\n", which ? "signal" : "plain device", hipGetErrorString(e)); continue; }
      HK(hipEventRecord(e0, st));
      for (int i = 0; i < 200; ++i) { HK(hipStreamWriteValue64(st, w, (unsigned long long)(i + 2), 0)); HK(hipStreamWaitValue64(st, w, (unsigned long long)(i + 2), hipStreamWaitValueEq, ~0ull)); }
      HK(hipEventRecord(e1, st)); HK(hipEventSynchronize(e1));
      float ms = 0; HK(hipEventElapsedTime(&ms, e0, e1));
      printf("hipStreamWriteValue64 + hipStreamWaitValue64 (already satisfied) on %s memory: %.2f us per pair\n", which ? "hipMallocSignalMemory" : "plain device", 1e3 * ms / 200);
      hipGraph_t gr = nullptr;
      e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      hipError_t ew = hipStreamWriteValue64(st, w, 7, 0), ewt = hipStreamWaitValue64(st, w, 7, hipStreamWaitValueEq, ~0ull);
      hipError_t ee = hipStreamEndCapture(st, &gr);
      printf("  under stream capture: write -> %s, wait -> %s, end capture -> %s\n", hipGetErrorString(ew), hipGetErrorString(ewt), hipGetErrorString(ee));
      (void)hipGetLastError();
      if (gr) hipGraphDestroy(gr);
    }
    hipFree(plain); if (sig) hipFree(sig);
  }
  CK(mgs_comm_destroy(c));
  CK(mgs_ctx_destroy(ctx));
  int rc = 0;
  if (R == 0) { for (pid_t p : kids) { int stt = 0; waitpid(p, &stt, 0); if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0) rc = 1; } printf(rc ? "PROBE FAILED\n" : "PROBE OK\n"); }
  return rc;
}
