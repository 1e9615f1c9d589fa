// rw_mix.hip — microbenchmark (dev tool, not product code): what does the memory system deliver for the SpMV's traffic MIX without
// any of its structure?  A workgroup streams the 14 KiB value slice of its 256 "rows" with coalesced 16-byte loads (no LDS, no
// barrier, no gather) and writes W bytes per row.  Variants: no store, 8 B/row plain / non-temporal / 16-byte stores by half the
// lanes, reads non-temporal, plus a float4 copy and a pure read for the ceilings of this box.
// build: hipcc --offload-arch=gfx950 -O3 -o rw_mix rw_mix.hip ; run: ./rw_mix [N=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
constexpr int RB = 256, NZ = 7, SLICE = RB * NZ;

// MODE 0: no store; 1: y[row] = s (8 B per lane); 2: non-temporal store; 3: lanes pair up, even lanes store 16 B; 4: like 1 with
// non-temporal LOADS of the slice
template <int MODE>
__global__ __launch_bounds__(RB) void mix(const double *__restrict__ val, double *__restrict__ y, long n) {
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RB;
  if (r0 >= n) return;
  const double *src = val + r0 * NZ;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = tid + k * RB;
    if (c < SLICE / 2) {
      const double2_t v = MODE == 4 ? __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(src + 2 * c)) : *reinterpret_cast<const double2_t *>(src + 2 * c);
      s += v.x + v.y;
    }
  }
  if (MODE == 0) { if (s == 1.2345e-300) y[r0 + tid] = s; }
  else if (MODE == 1 || MODE == 4) y[r0 + tid] = s;
  else if (MODE == 2) __builtin_nontemporal_store(s, y + r0 + tid);
  else {
    const double o = __shfl_xor(s, 1);
    if (!(tid & 1)) { double2_t p; p.x = s; p.y = o; *reinterpret_cast<double2_t *>(y + r0 + tid) = p; }
  }
}
// G consecutive row blocks per workgroup (fewer, longer-lived workgroups), store as in MODE 1
template <int G>
__global__ __launch_bounds__(RB) void mix_g(const double *__restrict__ val, double *__restrict__ y, long n) {
  const int tid = threadIdx.x;
  for (int g = 0; g < G; ++g) {
    const long r0 = ((long)blockIdx.x * G + g) * RB;
    if (r0 >= n) return;
    const double *src = val + r0 * NZ;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = tid + k * RB;
      if (c < SLICE / 2) { const double2_t v = *reinterpret_cast<const double2_t *>(src + 2 * c); s += v.x + v.y; }
    }
    y[r0 + tid] = s;
  }
}
__global__ __launch_bounds__(256) void copy4(const float4_t *__restrict__ a, float4_t *__restrict__ b, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; const long st = (long)gridDim.x * 256;
#pragma unroll 4
  for (; i < n4; i += st) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read4(const float4_t *__restrict__ a, float4_t *__restrict__ b, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x; const long st = (long)gridDim.x * 256;
  float4_t s = {0, 0, 0, 0};
#pragma unroll 4
  for (; i < n4; i += st) { const float4_t v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  if (s.x == 1.2345e-30f) b[0] = s;
}
__global__ void init(double *val, long m) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  for (long k = i; k < m; k += st) val[k] = 1.0 + (k % 5) * 0.25;
}
template <class K, class... A>
float timeit(K k, dim3 g, dim3 b, A... a) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  CK(hipGetLastError()); CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, g, b, 0, 0, a...);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 10;
}
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512;
  const long n = (long)N * N * N;
  double *val, *y;
  CK(hipMalloc(&val, 8 * n * NZ + 65536)); CK(hipMalloc(&y, 8 * n + 65536));
  hipLaunchKernelGGL(init, dim3(8192), dim3(256), 0, 0, val, n * NZ); CK(hipDeviceSynchronize());
  const dim3 g((unsigned)((n + RB - 1) / RB)), b(RB);
  auto rep = [&](const char *name, float ms, double bytes_per_row) { printf("%-52s %8.3f ms  %6.0f GB/s moved\n", name, ms, bytes_per_row * n / ms / 1e6); };
  rep("read 56 B/row, no store", timeit(mix<0>, g, b, val, y, n), 56);
  rep("read 56 + store 8 B/row (8 B per lane)", timeit(mix<1>, g, b, val, y, n), 64);
  rep("read 56 + non-temporal store 8 B/row", timeit(mix<2>, g, b, val, y, n), 64);
  rep("read 56 + store 8 B/row as 16 B from even lanes", timeit(mix<3>, g, b, val, y, n), 64);
  rep("non-temporal read 56 + store 8 B/row", timeit(mix<4>, g, b, val, y, n), 64);
  rep("read 56 + store 8, 4 row blocks per workgroup", timeit(mix_g<4>, dim3((g.x + 3) / 4), b, val, y, n), 64);
  rep("read 56 + store 8, 16 row blocks per workgroup", timeit(mix_g<16>, dim3((g.x + 15) / 16), b, val, y, n), 64);
  rep("read 56 + store 8 B/row (again)", timeit(mix<1>, g, b, val, y, n), 64);
  const long n4 = n * NZ / 2 / 2;       // half of val → other half: 3.76 GB each way
  rep("float4 copy (half of val to the other half)", timeit(copy4, dim3(256 * 8), dim3(256), (const float4_t *)val, (float4_t *)(val + n4 * 2), n4), 16.0 * n4 * 2 / n);
  rep("float4 read of val", timeit(read4, dim3(256 * 8), dim3(256), (const float4_t *)val, (float4_t *)y, n * NZ / 2), 56);
  return 0;
}
