// latency/issue micro-benchmarks for the single-wave dependent chains of the solver (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__device__ __forceinline__ double lane_bcast(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out, long long* t, double x0, int mode) {
  __shared__ double sh[1024];
  double x = x0 + threadIdx.x * 1e-9, y = x0 * 0.5, acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const int lane = threadIdx.x & 63;
  long long c0 = clock64();
  if (mode == 0) {            // dependent f64 FMA chain
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, y, 1e-3);
  } else if (mode == 1) {     // 8 independent f64 FMA chains
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fma(acc[k], y, 1e-3);
  } else if (mode == 2) {     // readlane -> fma chain
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, lane_bcast(x, i & 63), 1e-3);
  } else if (mode == 3) {     // rsq chain
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x) + 1.0;
  } else if (mode == 4) {     // LDS write -> barrier -> read chain (block of 256)
#pragma unroll
    for (int i = 0; i < N; ++i) { sh[threadIdx.x] = x; __syncthreads(); x = sh[(threadIdx.x + 64) & 255] + 1.0; }
  } else if (mode == 5) {     // LDS write -> read same wave (no barrier)
#pragma unroll
    for (int i = 0; i < N; ++i) { sh[threadIdx.x] = x; x = sh[threadIdx.x ^ 1] + 1.0; }
  } else if (mode == 6) {     // dependent f32 FMA chain
    float f = (float)x, g = (float)y;
#pragma unroll
    for (int i = 0; i < N; ++i) f = fmaf(f, g, 1e-3f);
    x = f;
  } else if (mode == 7) {     // DPP-style wave shuffle chain
#pragma unroll
    for (int i = 0; i < N; ++i) x = x + __shfl_xor(x, 1);
  } else if (mode == 8) {     // barrier only
#pragma unroll
    for (int i = 0; i < N; ++i) { __syncthreads(); x += 1.0; }
  } else if (mode == 9) {     // independent v_readlane pairs + fma (the in-register update pattern)
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fma(-y, lane_bcast(y, (i * 8 + k) & 63), acc[k]);
  } else if (mode == 10) {    // dependent f64 mul chain
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * y;
  } else if (mode == 11) {    // dependent f64 add chain
#pragma unroll
    for (int i = 0; i < N; ++i) x = x + y;
  } else if (mode == 12) {    // 8 independent v_fmac_f64 with a DPP row_newbcast operand
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[k]) : "v"(x), "v"(y));
  } else if (mode == 13) {    // ds_read_b128 broadcast (uniform address) + 2 fma per read
    const double2* sp2 = reinterpret_cast<const double2*>(sh);
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
      for (int k = 0; k < 8; k += 2) { const double2 v = sp2[(i * 4 + k / 2) & 255]; acc[k] = fma(v.x, y, acc[k]); acc[k + 1] = fma(v.y, y, acc[k + 1]); }
  }
  long long c1 = clock64();
  for (int k = 0; k < 8; ++k) x += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}
int main() {
  double* out; long long* t; hipMalloc(&out, 1 << 20); hipMalloc(&t, 4096);
  const char* names[] = {"dependent f64 fma", "8 independent f64 fma chains", "readlane->fma chain", "v_rsq_f64 + add chain", "LDS write->barrier->read (256 thr)", "LDS write->read same wave",
                         "dependent f32 fma", "shfl_xor + add chain", "barrier + add (256 thr)", "independent readlane pair + fma", "dependent f64 mul", "dependent f64 add", "independent v_fmac_f64_dpp row_newbcast", "uniform ds_read_b128 + 2 fma (per fma)"};
  for (int threads : {64, 256, 512})
    for (int mode = 0; mode < 14; ++mode) {
      long long h = 0;
      for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, t, 1.0001, mode); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); }
      printf("threads %3d  %-38s %7.1f clocks per step\n", threads, names[mode], (double)h / N);
    }
  return 0;
}
