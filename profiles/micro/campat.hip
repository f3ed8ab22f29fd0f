// Round 6: does the POWER-OF-TWO row-group stride of the K = 8192 matrices cost the prefill GEMM's o_proj / QKV their bandwidth?
// Pure reads in the prefill GEMM's access structure (csrc/gemm_pf.hip): grid (N / 128, S) workgroups of 4 waves, wave w streams the two
// row groups (tile * 8 + 2 w, + 1) over its split's k-tiles, 1 KiB per (row group, k-tile), 8 k-steps in flight per wave (two register
// batches), all workgroups start together and walk in lock-step.  Variants:
//   stride   row-group stride in KiB: 256 (today's fragment-major layout at K = 8192) against padded strides
//   rot      every workgroup starts its walk at its own k offset (and wraps): the same bytes, de-phased
// hipcc --offload-arch=gfx950 -O3 -o /tmp/campat profiles/micro/campat.hip && /tmp/campat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) rd(const u32x4* __restrict__ W, size_t rg_stride, int nk, int rot, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t g0 = ((size_t)blockIdx.x * WAVES + wave) * 2;
  const int kz0 = blockIdx.y * nk;
  const u32x4* p0 = W + g0 * rg_stride + lane;
  const u32x4* p1 = p0 + rg_stride;
  const int start = rot ? (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)nk) & ~7 : 0;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 a[8][2], b[8][2];
  auto ld = [&](u32x4 (&d)[8][2], int i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int k = start + i + u;
      if (k >= nk) k -= nk;
      d[u][0] = __builtin_nontemporal_load(p0 + ((size_t)(kz0 + k) << 6));
      d[u][1] = __builtin_nontemporal_load(p1 + ((size_t)(kz0 + k) << 6));
    }
  };
  auto use = [&](u32x4 (&d)[8][2]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc ^= d[u][0]; acc ^= d[u][1]; }
  };
  ld(a, 0);
  for (int i = 0; i < nk; i += 16) {
    if (i + 8 < nk) ld(b, i + 8);
    use(a);
    if (i + 16 < nk) ld(a, i + 16);
    if (i + 8 < nk) use(b);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *out = 1;
}

template <int WAVES>
static float run(const u32x4* W, size_t stride_kib, int tiles, int S, int nk, int rot, unsigned* out, int copies, size_t copy_chunks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rd<WAVES>, dim3(tiles, S), dim3(64 * WAVES), 0, 0, W + (i % copies) * copy_chunks, stride_kib * 64, nk, rot, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(rd<WAVES>, dim3(tiles, S), dim3(64 * WAVES), 0, 0, W + (i % copies) * copy_chunks, stride_kib * 64, nk, rot, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  // room for 4 copies of [8192 + pad rows groups] x up to 320 KiB per row group
  const size_t groups = 640, max_stride_kib = 320, copies = 4;
  const size_t copy_chunks = groups * max_stride_kib * 64;         // 16-byte chunks per copy
  u32x4* W;
  unsigned* out;
  if (hipMalloc(&W, copies * copy_chunks * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 4);
  hipMemset(W, 1, copies * copy_chunks * 16);
  struct { const char* name; int tiles, waves, S, nk; } shapes[] = {
      {"o_proj  [8192 x 8192]  w4 s4", 64, 4, 4, 64}, {"qkv     [10240 x 8192] w5 s4", 64, 5, 4, 64}, {"o_proj-like w8 s4 (256-row tiles)", 32, 8, 4, 64},
      {"o_proj  w4 s1 (whole K per workgroup: 64 wgs)", 64, 4, 1, 256}};
  for (auto& sh : shapes) {
    const double mb = (double)sh.tiles * sh.waves * 2 * sh.S * sh.nk * 1024 / 1e6;
    for (size_t stride : {256, 257, 258, 260, 264, 272, 288, 320}) {
      for (int rot = 0; rot < 2; ++rot) {
        float us = sh.waves == 4 ? run<4>(W, stride, sh.tiles, sh.S, sh.nk, rot, out, copies, copy_chunks)
                 : sh.waves == 5 ? run<5>(W, stride, sh.tiles, sh.S, sh.nk, rot, out, copies, copy_chunks)
                                 : run<8>(W, stride, sh.tiles, sh.S, sh.nk, rot, out, copies, copy_chunks);
        printf("%-48s %6.1f MB  stride %3zu KiB rot %d : %7.2f us  %5.2f TB/s\n", sh.name, mb, stride, rot, us, mb / us);
      }
    }
  }
  return 0;
}
