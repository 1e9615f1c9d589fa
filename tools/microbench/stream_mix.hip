// Microbenchmark (dev tool, not product code): how fast can a wave64 kernel stream the CSR
// val(f64)+col(i32) mix on MI355X, by load width and structure, with and without the x gather?
// build: hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int2_t __attribute__((ext_vector_type(2)));
typedef double double2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W, bool NT, bool GATHER, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const double *__restrict__ val, const int *__restrict__ col, const double *__restrict__ x,
                                                     long nnz, double *__restrict__ out) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * W;
  const long stride = (long)gridDim.x * blockDim.x * W;
  double acc = 0.0;
#pragma unroll UNROLL
  for (; i + W <= nnz; i += stride) {
    if (W == 1) {
      double v = NT ? __builtin_nontemporal_load(val + i) : val[i];
      int c = NT ? __builtin_nontemporal_load(col + i) : col[i];
      acc += GATHER ? v * x[c] : v + (double)c;
    } else if (W == 2) {
      double2_t v = NT ? __builtin_nontemporal_load((const double2_t *)(val + i)) : *(const double2_t *)(val + i);
      int2_t c = NT ? __builtin_nontemporal_load((const int2_t *)(col + i)) : *(const int2_t *)(col + i);
      acc += GATHER ? v.x * x[c.x] + v.y * x[c.y] : v.x + v.y + (double)(c.x + c.y);
    } else {
      double2_t v0 = NT ? __builtin_nontemporal_load((const double2_t *)(val + i)) : *(const double2_t *)(val + i);
      double2_t v1 = NT ? __builtin_nontemporal_load((const double2_t *)(val + i + 2)) : *(const double2_t *)(val + i + 2);
      int4_t c = NT ? __builtin_nontemporal_load((const int4_t *)(col + i)) : *(const int4_t *)(col + i);
      acc += GATHER ? v0.x * x[c.x] + v0.y * x[c.y] + v1.x * x[c.z] + v1.y * x[c.w] : v0.x + v0.y + v1.x + v1.y + (double)(c.x + c.y + c.z + c.w);
    }
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

__global__ void init_kernel(double *val, int *col, long nnz, int N) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  const long n = (long)N * N * N, N2 = (long)N * N;
  for (; i < nnz; i += stride) {
    val[i] = 1.0 + (i & 7);
    long row = i / 7; int q = (int)(i % 7);   // 7-point pattern: e-N2,e-N,e-1,e,e+1,e+N,e+N2 (clamped)
    long off[7] = {-N2, -N, -1, 0, 1, N, N2};
    long c = row + off[q]; if (c < 0) c = 0; if (c >= n) c = n - 1;
    col[i] = (int)c;
  }
}

template <int W, bool NT, bool G, int U>
void run(const char *name, const double *val, const int *col, const double *x, long nnz, double *out, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<W, NT, G, U>), dim3(blocks), dim3(256), 0, 0, val, col, x, nnz, out);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<W, NT, G, U>), dim3(blocks), dim3(256), 0, 0, val, col, x, nnz, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-34s blocks %7d  %8.3f ms  %7.0f GB/s (12 B/nnz)\n", name, blocks, ms, 12.0 * nnz / ms / 1e6);
}

int main(int argc, char **argv) {
  int N = argc > 1 ? atoi(argv[1]) : 512;
  long n = (long)N * N * N, nnz = 7 * n;
  double *val, *x, *out; int *col;
  CK(hipMalloc(&val, 8 * nnz + 64)); CK(hipMalloc(&col, 4 * nnz + 64)); CK(hipMalloc(&x, 8 * n)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(init_kernel, dim3(8192), dim3(256), 0, 0, val, col, nnz, N);
  CK(hipMemset(x, 0, 8 * n)); CK(hipDeviceSynchronize());
  for (int blocks : {2048, 4096, 8192, 65536}) {
    run<1, false, false, 8>("w1 plain", val, col, x, nnz, out, blocks);
    run<1, true, false, 8>("w1 nt", val, col, x, nnz, out, blocks);
    run<2, false, false, 4>("w2 plain", val, col, x, nnz, out, blocks);
    run<2, true, false, 4>("w2 nt", val, col, x, nnz, out, blocks);
    run<4, false, false, 4>("w4 plain", val, col, x, nnz, out, blocks);
    run<4, true, false, 4>("w4 nt", val, col, x, nnz, out, blocks);
    run<4, true, false, 8>("w4 nt unroll8", val, col, x, nnz, out, blocks);
    run<1, true, true, 8>("w1 nt + gather x", val, col, x, nnz, out, blocks);
    run<2, true, true, 4>("w2 nt + gather x", val, col, x, nnz, out, blocks);
    run<4, true, true, 4>("w4 nt + gather x", val, col, x, nnz, out, blocks);
  }
  return 0;
}
