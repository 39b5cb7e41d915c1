#include <hip/hip_runtime.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(double& x, double& y) {
  v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void swap16(double& x, double& y) {
  v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}
__global__ void k(double* o) {
  double x = threadIdx.x, y = 100.0 + threadIdx.x;
  double a = x, b = y; swap32(a, b);
  double c = x, d = y; swap16(c, d);
  o[threadIdx.x] = a; o[64 + threadIdx.x] = b; o[128 + threadIdx.x] = c; o[192 + threadIdx.x] = d;
}
int main() {
  double* d; hipMalloc(&d, 256 * 8); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) { for (int i = 0; i < 64; i += 8) printf("%5.0f ", h[64 * r + i]); printf("\n"); }
  return 0;
}
