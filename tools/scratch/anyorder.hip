// Does hipExtAnyOrderLaunch (dispatch packet without the barrier bit) let a kernel start while its predecessor in the
// SAME stream is still running on gfx950?  A spins for ~50 us; B records when it starts.
//   hipcc -O2 --offload-arch=gfx950 -o anyorder tools/scratch/anyorder.hip && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_spin(unsigned long long* out, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) out[0] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[1] = wall_clock64();
}
__global__ void k_mark(unsigned long long* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = wall_clock64();
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_spin, dim3(19), dim3(256), 0, s, d, 5000ull);        // 50 us of the 100 MHz clock
      hipExtLaunchKernelGGL(k_mark, dim3(3968), dim3(256), 0, s, nullptr, nullptr, flags, d);
      hipError_t e = hipGetLastError();
      hipStreamSynchronize(s);
      unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("flags %d: launch rc %d; spin ran %.1f us; the next kernel started %+.1f us relative to the spin's END (%s)\n", flags, (int)e,
             (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0, h[2] < h[1] ? "OVERLAPPED" : "after it");
    }
  }
  return 0;
}
