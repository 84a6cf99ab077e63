// Empirical operand/result layout of the fp64 MFMAs with B = 1: which lanes' A values does each output hold?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* out4x4, double* out16) {
  const int lane = threadIdx.x;
  // A value = 2^lane would overflow the mantissa for 64 lanes; use two runs: lane id encoded as 1 << (lane % 32) in hi/lo
  double a_lo = lane < 32 ? (double)(1ull << lane) : 0.0;
  double a_hi = lane >= 32 ? (double)(1ull << (lane - 32)) : 0.0;
  double d_lo = __builtin_amdgcn_mfma_f64_4x4x4f64(a_lo, 1.0, 0.0, 0, 0, 0);
  double d_hi = __builtin_amdgcn_mfma_f64_4x4x4f64(a_hi, 1.0, 0.0, 0, 0, 0);
  out4x4[lane * 2] = d_lo; out4x4[lane * 2 + 1] = d_hi;
  double4_t z = {0, 0, 0, 0};
  double4_t e_lo = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lo, 1.0, z, 0, 0, 0);
  double4_t e_hi = __builtin_amdgcn_mfma_f64_16x16x4f64(a_hi, 1.0, z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { out16[(lane * 4 + r) * 2] = e_lo[r]; out16[(lane * 4 + r) * 2 + 1] = e_hi[r]; }
}
int main() {
  double *a, *b; hipMalloc(&a, 64 * 2 * 8); hipMalloc(&b, 64 * 4 * 2 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b);
  double ha[128], hb[512]; hipMemcpy(ha, a, sizeof(ha), hipMemcpyDeviceToHost); hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost);
  printf("4x4x4 (B = 1): lane -> set of source lanes summed\n");
  for (int l = 0; l < 64; ++l) {
    unsigned long long lo = (unsigned long long)ha[2 * l], hi = (unsigned long long)ha[2 * l + 1];
    printf("lane %2d:", l);
    for (int s = 0; s < 32; ++s) if (lo >> s & 1) printf(" %d", s);
    for (int s = 0; s < 32; ++s) if (hi >> s & 1) printf(" %d", s + 32);
    printf("\n");
  }
  printf("16x16x4 (B = 1): lane, reg -> source lanes\n");
  for (int l = 0; l < 64; l += 1) for (int r = 0; r < 4; ++r) {
    if (l % 16 > 1) continue;
    unsigned long long lo = (unsigned long long)hb[(l * 4 + r) * 2], hi = (unsigned long long)hb[(l * 4 + r) * 2 + 1];
    printf("lane %2d reg %d:", l, r);
    for (int s = 0; s < 32; ++s) if (lo >> s & 1) printf(" %d", s);
    for (int s = 0; s < 32; ++s) if (hi >> s & 1) printf(" %d", s + 32);
    printf("\n");
  }
  return 0;
}
