// What read bandwidth does MI355X HBM3E sustain for a pure streaming read (the ceiling of the weight-streaming GEMMs)?
// hipcc --offload-arch=gfx950 -O3 -o readbw readbw.hip && ./readbw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U, bool NT>
__global__ void __launch_bounds__(1024) rd(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
  // each workgroup streams a contiguous slab; each thread keeps U 16-byte loads in flight
  const size_t per = n16 / gridDim.x;
  const u32x4* q = p + (size_t)blockIdx.x * per;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i + (U - 1) * blockDim.x < per; i += (size_t)U * blockDim.x) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(q + i + u * blockDim.x) : q[i + u * blockDim.x];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *out = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U, bool NT>
static int run(const u32x4* buf, size_t n16, unsigned* out, int blocks, int threads) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    rd<U, NT><<<blocks, threads>>>(buf, n16, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("blocks=%5d threads=%4d U=%d nt=%d : %7.1f us  %.2f TB/s\n", blocks, threads, U, (int)NT, best * 1e3, n16 * 16.0 / best / 1e9);
  return 0;
}

int main() {
  const size_t bytes = (size_t)4 << 30;     // 4 GiB >> Infinity Cache
  u32x4* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 1, bytes));
  const size_t n16 = bytes / 16;
  for (int blocks : {256, 512, 1024, 2048, 4096})
    for (int threads : {256, 512, 1024}) {
      run<4, true>(buf, n16, out, blocks, threads);
      run<8, true>(buf, n16, out, blocks, threads);
    }
  run<4, false>(buf, n16, out, 2048, 512);
  run<8, false>(buf, n16, out, 2048, 512);
  run<16, true>(buf, n16, out, 1024, 512);
  return 0;
}
