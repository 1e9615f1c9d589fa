// spmv_pipe.hip — microbenchmark (dev tool, not product code): does the row-block SpMV structure gain from keeping the NEXT row
// block's value slice in flight (LDS-DMA, `global_load_lds_dwordx4`, double-buffered LDS) while the current block is summed?
// Operator: 7 entries per row, columns row + {−N², −N, −1, 0, 1, N, N²} (clamped), values streamed from HBM: the shape of the
// pattern-coded fine-level kernel without its tables.  Variants:
//   A  one 256-row block per workgroup: stage through registers → LDS, barrier, gather + sum (the shipped structure)
//   C  G consecutive blocks per workgroup, same staging, no prefetch
//   B  G consecutive blocks per workgroup, LDS-DMA into the other LDS buffer one block ahead
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o spmv_pipe spmv_pipe.hip ; run: ./spmv_pipe [N=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int RB = 256, NZ = 7, SLICE = RB * NZ;    // 1792 doubles = 14336 B = 14 pieces of 1 KiB

template <bool GATHER = true>
__device__ __forceinline__ double row_sum(const double *__restrict__ vals, int tid, long row, long n, long N, long N2, const double *__restrict__ x) {
  const long off[NZ] = {-N2, -N, -1, 0, 1, N, N2};
  double xv[NZ], s = 0.0;
#pragma unroll
  for (int q = 0; q < NZ; ++q) { long c = row + off[q]; c = c < 0 ? 0 : (c >= n ? n - 1 : c); xv[q] = GATHER ? x[c] : 1.0 + q; }
#pragma unroll
  for (int q = 0; q < NZ; ++q) s += vals[tid * NZ + q] * xv[q];
  return s;
}

// MODE 1: no x gather; 2: no y store (one conditional store); 3: no value stream (LDS holds garbage) — which part costs what
template <int MODE>
__global__ __launch_bounds__(RB) void spmv_part(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  __shared__ double vals[SLICE];
  const int tid = threadIdx.x;
  const long N2 = (long)N * N, r0 = (long)blockIdx.x * RB;
  if (r0 >= n) return;
  const double *src = val + r0 * NZ;
  if (MODE != 3) {
#pragma unroll 4
    for (int c = tid; c < SLICE / 2; c += RB) *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(src + 2 * c);
  } else if (tid == 0) vals[0] = 1.0;
  __syncthreads();
  const double s = MODE == 1 ? row_sum<false>(vals, tid, r0 + tid, n, N, N2, x) : row_sum<true>(vals, tid, r0 + tid, n, N, N2, x);
  if (MODE != 2 || s == 1.2345e-300) y[r0 + tid] = s;
}

// T: values stored block-transposed (ELL inside a 256-row block: val_T[block][q][t]) — every lane loads its own entries with
// coalesced 8-byte loads, no LDS, no barrier: a wave has all 14 loads of its rows in flight at once
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void spmv_ell(const double *__restrict__ valT, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  const long N2 = (long)N * N;
  const long row = (long)blockIdx.x * (64 * WAVES) + threadIdx.x;
  if (row >= n) return;
  const long blk = row / RB; const int t = (int)(row % RB);
  const double *src = valT + blk * SLICE + t;
  const long off[NZ] = {-N2, -N, -1, 0, 1, N, N2};
  double v[NZ], xv[NZ], s = 0.0;
#pragma unroll
  for (int q = 0; q < NZ; ++q) v[q] = src[q * RB];
#pragma unroll
  for (int q = 0; q < NZ; ++q) { long c = row + off[q]; c = c < 0 ? 0 : (c >= n ? n - 1 : c); xv[q] = x[c]; }
#pragma unroll
  for (int q = 0; q < NZ; ++q) s += v[q] * xv[q];
  y[row] = s;
}

// X: like A, but the five contiguous runs of x a 256-row block reads (e−N², e−N, e−1..e+1, e+N, e+N²) are staged in LDS with wide
// coalesced loads instead of 7 eight-byte gathers per row (interior blocks only; boundary blocks gather)
__global__ __launch_bounds__(RB) void spmv_xlds(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  __shared__ double vals[SLICE];
  __shared__ double xs[5][RB + 2];
  const int tid = threadIdx.x;
  const long N2 = (long)N * N, r0 = (long)blockIdx.x * RB;
  if (r0 >= n) return;
  const double *src = val + r0 * NZ;
#pragma unroll 4
  for (int c = tid; c < SLICE / 2; c += RB) *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(src + 2 * c);
  const bool interior = r0 - N2 >= 0 && r0 + RB + N2 <= n;      // block-uniform
  if (interior) {
    const long d[4] = {-N2, -N, N, N2};
    if (tid < RB / 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<double2_t *>(&xs[r < 2 ? r : r + 1][2 * tid]) = *reinterpret_cast<const double2_t *>(x + r0 + d[r] + 2 * tid);
    } else {
      const int t = tid - RB / 2;                                // the middle run x[r0−1 .. r0+256]: 258 doubles, 8-byte loads
      xs[2][t] = x[r0 - 1 + t]; xs[2][t + RB / 2] = x[r0 - 1 + t + RB / 2];
      if (t < 2) xs[2][RB + t] = x[r0 - 1 + RB + t];
    }
  }
  __syncthreads();
  double s = 0.0;
  if (interior) {
    const double xv[NZ] = {xs[0][tid], xs[1][tid], xs[2][tid], xs[2][tid + 1], xs[2][tid + 2], xs[3][tid], xs[4][tid]};
#pragma unroll
    for (int q = 0; q < NZ; ++q) s += vals[tid * NZ + q] * xv[q];
  } else s = row_sum(vals, tid, r0 + tid, n, N, N2, x);
  y[r0 + tid] = s;
}

// Z: like A, but the x gathers are issued BEFORE the barrier, beside the loads of the value slice (possible here because the
// columns follow from the row number; in the product they come from the block's pattern table in LDS): what would ONE memory latency
// per workgroup instead of two be worth?
__global__ __launch_bounds__(RB) void spmv_early(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  __shared__ double vals[SLICE];
  const int tid = threadIdx.x;
  const long N2 = (long)N * N, r0 = (long)blockIdx.x * RB;
  if (r0 >= n) return;
  const double *src = val + r0 * NZ;
  const long row = r0 + tid;
  const long off[NZ] = {-N2, -N, -1, 0, 1, N, N2};
  double2_t rr[4]; double xv[NZ];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int c = tid + k * RB; if (c < SLICE / 2) rr[k] = *reinterpret_cast<const double2_t *>(src + 2 * c); }
#pragma unroll
  for (int q = 0; q < NZ; ++q) { long c = row + off[q]; c = c < 0 ? 0 : (c >= n ? n - 1 : c); xv[q] = x[c]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int c = tid + k * RB; if (c < SLICE / 2) *reinterpret_cast<double2_t *>(vals + 2 * c) = rr[k]; }
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < NZ; ++q) s += vals[tid * NZ + q] * xv[q];
  y[row] = s;
}

// W: like A, but every wave stages only the value slice of ITS 64 rows (3.5 KiB, contiguous) and nobody waits for the other waves:
// no workgroup barrier, only the wave's own LDS ordering
__global__ __launch_bounds__(RB) void spmv_wave(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  __shared__ double vals[SLICE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long N2 = (long)N * N, r0 = (long)blockIdx.x * RB;
  if (r0 >= n) return;
  constexpr int WS = 64 * NZ;                       // doubles per wave slice (448), 224 pairs
  const double *src = val + (r0 + 64 * wave) * NZ;
  double *dst = vals + wave * WS;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < WS / 2) *reinterpret_cast<double2_t *>(dst + 2 * c) = *reinterpret_cast<const double2_t *>(src + 2 * c);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  y[r0 + tid] = row_sum(dst, lane, r0 + tid, n, N, N2, x);
}

template <int G>
__global__ __launch_bounds__(RB) void spmv_reg(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  __shared__ double vals[SLICE];
  const int tid = threadIdx.x;
  const long N2 = (long)N * N;
  for (int g = 0; g < G; ++g) {
    const long blk = (long)blockIdx.x * G + g, r0 = blk * RB;
    if (r0 >= n) return;
    if (g) __syncthreads();
    const double *src = val + r0 * NZ;
#pragma unroll 4
    for (int c = tid; c < SLICE / 2; c += RB) *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(src + 2 * c);
    __syncthreads();
    y[r0 + tid] = row_sum(vals, tid, r0 + tid, n, N, N2, x);
  }
}

// one wave copies pieces (1 KiB each: 64 lanes × 16 B) p = wave, wave+4, … of the 14-piece slice into LDS, no VGPR destination
__device__ __forceinline__ void issue_slice(const double *__restrict__ src, double *lds, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int piece = wave + 4 * p;
    if (piece < SLICE * 8 / 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 128 + lane * 2),
                                       (__attribute__((address_space(3))) void *)(lds + piece * 128), 16, 0, 0);
  }
}

template <int G>
__global__ __launch_bounds__(RB) void spmv_glds(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  __shared__ double vals[2][SLICE];
  const int tid = threadIdx.x;
  const long N2 = (long)N * N;
  const long first = (long)blockIdx.x * G;
  if (first * RB >= n) return;
  issue_slice(val + first * RB * NZ, vals[0], tid);
  for (int g = 0; g < G; ++g) {
    const long r0 = (first + g) * RB;
    if (r0 >= n) return;
    __syncthreads();                               // fence: vmcnt(0) (slice g landed) + barrier (everyone is done with the other buffer)
    if (g + 1 < G && r0 + RB < n) issue_slice(val + (r0 + RB) * NZ, vals[(g + 1) & 1], tid);
    y[r0 + tid] = row_sum(vals[g & 1], tid, r0 + tid, n, N, N2, x);
  }
}

// R2: two row blocks per workgroup, every lane owns row t of both: one barrier per 512 rows, the 14 gathers of a lane's two rows in flight together
__global__ __launch_bounds__(RB) void spmv_two(const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y, long n, int N) {
  __shared__ double vals[2 * SLICE];
  const int tid = threadIdx.x;
  const long N2 = (long)N * N, r0 = (long)blockIdx.x * 2 * RB;
  if (r0 >= n) return;
  const double *src = val + r0 * NZ;
#pragma unroll 7
  for (int c = tid; c < SLICE; c += RB) *reinterpret_cast<double2_t *>(vals + 2 * c) = *reinterpret_cast<const double2_t *>(src + 2 * c);
  __syncthreads();
  const long off[NZ] = {-N2, -N, -1, 0, 1, N, N2};
  const long ra = r0 + tid, rb = r0 + RB + tid;
  double xa[NZ], xb[NZ], sa = 0.0, sb = 0.0;
#pragma unroll
  for (int q = 0; q < NZ; ++q) { long c = ra + off[q]; c = c < 0 ? 0 : (c >= n ? n - 1 : c); xa[q] = x[c]; }
#pragma unroll
  for (int q = 0; q < NZ; ++q) { long c = rb + off[q]; c = c < 0 ? 0 : (c >= n ? n - 1 : c); xb[q] = x[c]; }
#pragma unroll
  for (int q = 0; q < NZ; ++q) { sa += vals[tid * NZ + q] * xa[q]; sb += vals[SLICE + tid * NZ + q] * xb[q]; }
  y[ra] = sa; y[rb] = sb;
}

template <class K>
void run(const char *name, K kernel, int rows_per_wg, const double *val, const double *x, double *y, long n, int N, int threads = RB) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (int)((n + rows_per_wg - 1) / rows_per_wg);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, val, x, y, n, N);
  CK(hipGetLastError());
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, 0, val, x, y, n, N);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  double chk = 0; CK(hipMemcpy(&chk, y + n / 2, 8, hipMemcpyDeviceToHost));
  printf("%-40s %8.3f ms  %6.0f GB/s (72 B/row)   y[n/2] = %.6f\n", name, ms, 72.0 * n / ms / 1e6, chk);
}

__global__ void init(double *val, double *x, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  for (long k = i; k < n * NZ; k += st) val[k] = 1.0 + (k % 5) * 0.25;
  for (long k = i; k < n; k += st) x[k] = 0.5 + (k % 3);
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512;
  const long n = (long)N * N * N;
  double *val, *x, *y;
  CK(hipMalloc(&val, 8 * n * NZ + 65536)); CK(hipMalloc(&x, 8 * n)); CK(hipMalloc(&y, 8 * n + 65536));
  hipLaunchKernelGGL(init, dim3(8192), dim3(256), 0, 0, val, x, n); CK(hipDeviceSynchronize());
  run("A  regs->LDS, 1 block/WG", spmv_reg<1>, RB, val, x, y, n, N);
  run("C  regs->LDS, 4 blocks/WG", spmv_reg<4>, 4 * RB, val, x, y, n, N);
  run("C  regs->LDS, 16 blocks/WG", spmv_reg<16>, 16 * RB, val, x, y, n, N);
  run("B  LDS-DMA one block ahead, 4 blocks/WG", spmv_glds<4>, 4 * RB, val, x, y, n, N);
  run("B  LDS-DMA one block ahead, 16 blocks/WG", spmv_glds<16>, 16 * RB, val, x, y, n, N);
  run("B  LDS-DMA one block ahead, 64 blocks/WG", spmv_glds<64>, 64 * RB, val, x, y, n, N);
  run("A  regs->LDS, 1 block/WG (again)", spmv_reg<1>, RB, val, x, y, n, N);
  run("T  block-ELL values, no LDS, 256 thr/WG", spmv_ell<4>, RB, val, x, y, n, N);
  run("T  block-ELL values, no LDS, 64 thr/WG", spmv_ell<1>, 64, val, x, y, n, N, 64);
  run("T  block-ELL values, no LDS, 512 thr/WG", spmv_ell<8>, 512, val, x, y, n, N, 512);
  run("X  A + x runs staged in LDS (16-B loads)", spmv_xlds, RB, val, x, y, n, N);
  run("W  wave-private slices, no workgroup barrier", spmv_wave, RB, val, x, y, n, N);
  run("A  regs->LDS, 1 block/WG (again)", spmv_reg<1>, RB, val, x, y, n, N);
  run("W  wave-private slices (again)", spmv_wave, RB, val, x, y, n, N);
  run("Z  gathers issued before the barrier", spmv_early, RB, val, x, y, n, N);
  run("A  regs->LDS, 1 block/WG (again)", spmv_reg<1>, RB, val, x, y, n, N);
  run("Z  gathers issued before the barrier (again)", spmv_early, RB, val, x, y, n, N);
  run("R2 two rows per lane, 512 rows/WG", spmv_two, 2 * RB, val, x, y, n, N);
  run("A  regs->LDS, 1 block/WG (again)", spmv_reg<1>, RB, val, x, y, n, N);
  run("R2 two rows per lane (again)", spmv_two, 2 * RB, val, x, y, n, N);
  run("A without the x gather   (64 B/row moved)", spmv_part<1>, RB, val, x, y, n, N);
  run("A without the y store    (64 B/row moved)", spmv_part<2>, RB, val, x, y, n, N);
  run("A without the val stream (16 B/row moved)", spmv_part<3>, RB, val, x, y, n, N);
  return 0;
}
