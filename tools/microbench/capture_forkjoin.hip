// capture_forkjoin.hip — does hipStreamEndCapture survive a capture that forks to the SAME second stream more than once?
// (experiment for the overlapped halo exchange of the captured native cycle; build: hipcc --offload-arch=gfx950 -o capture_forkjoin capture_forkjoin.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void add1(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
int main(int argc, char **argv) {
  const int forks = argc > 1 ? atoi(argv[1]) : 2, fresh = argc > 2 ? atoi(argv[2]) : 0;
  const int n = 1 << 20;
  double *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  hipStream_t m, side[8]; CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
  for (auto &s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev[32]; for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CK(hipStreamBeginCapture(m, hipStreamCaptureModeThreadLocal));
  for (int f = 0; f < forks; ++f) {
    hipStream_t s = side[fresh ? f % 8 : 0];
    add1<<<n / 256, 256, 0, m>>>(a, n);
    CK(hipEventRecord(ev[2 * f], m)); CK(hipStreamWaitEvent(s, ev[2 * f], 0));
    add1<<<n / 256, 256, 0, s>>>(b, n);
    CK(hipEventRecord(ev[2 * f + 1], s));
    add1<<<n / 256, 256, 0, m>>>(a, n);
    CK(hipStreamWaitEvent(m, ev[2 * f + 1], 0));
    add1<<<n / 256, 256, 0, m>>>(b, n);
  }
  hipGraph_t g; printf("ending capture (forks=%d fresh=%d)\n", forks, fresh); fflush(stdout);
  CK(hipStreamEndCapture(m, &g));
  hipGraphExec_t x; CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(x, m));
  CK(hipStreamSynchronize(m));
  double ha, hb; CK(hipMemcpy(&ha, a, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b, 8, hipMemcpyDeviceToHost));
  printf("ok: a=%g (expect %d) b=%g (expect %d)\n", ha, 3 * 2 * forks, hb, 3 * 2 * forks);
  return 0;
}
