// Does the DRAM access PATTERN of the skinny GEMM cost bandwidth?  Same bytes, same workgroup / wave structure
// (N/16/NT workgroups x 8 waves, each wave streams its K-slice of NT row groups in 1 KiB tiles), two tile orders:
//   A: tiles ordered [row group][k-tile]            (today's fragment-major layout: a wave alternates between NT
//                                                   streams that are KT KiB apart)
//   B: tiles ordered [row-group block][k-tile][nt]  (a wave reads NT KiB contiguous per k-step, 1 contiguous run)
// hipcc --offload-arch=gfx950 -O3 -o readpat readpat.hip && ./readpat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// C: today's tile order, but the waves of a workgroup take k-tiles round-robin (wave w: kt = 2w, 2w+1, then + 2*nw ...), so the
//    workgroup as a whole walks each row group's K run linearly
template <int NT>
__global__ void __launch_bounds__(1024) rdc(const u32x4* __restrict__ W, int KT, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t tile0 = (size_t)blockIdx.x * NT;
  u32x4 acc = {0, 0, 0, 0};
  for (int kt = 2 * wave; kt < KT; kt += 2 * nw) {
    u32x4 v[2][NT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) v[u][nt] = __builtin_nontemporal_load(W + ((((tile0 + nt) * KT + kt + u) << 6) + lane));
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc ^= v[u][nt];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *out = 1;
}

template <int NT, bool PATB>
__global__ void __launch_bounds__(1024) rd(const u32x4* __restrict__ W, int KT, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int kt0 = KT * wave / nw, kt1 = KT * (wave + 1) / nw;
  const size_t tile0 = (size_t)blockIdx.x * NT;
  u32x4 acc = {0, 0, 0, 0};
  for (int kt = kt0; kt < kt1; kt += 2) {
    u32x4 v[2][NT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const size_t chunk = PATB ? (((tile0 * KT + (size_t)(kt + u) * NT + nt) << 6) + lane)
                                  : ((((tile0 + nt) * KT + kt + u) << 6) + lane);
        v[u][nt] = __builtin_nontemporal_load(W + chunk);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc ^= v[u][nt];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *out = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT>
static int runc(const u32x4* buf, int N, int K, int waves, unsigned* out, int copies, size_t stride16) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = N / 16 / NT, KT = K / 32;
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int c = 0; c < copies; ++c) rdc<NT><<<blocks, waves * 64>>>(buf + c * stride16, KT, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("N=%6d K=%6d NT=%d waves=%2d pattern C : %7.1f us/launch  %.2f TB/s\n", N, K, NT, waves, best * 1e3 / copies,
         (double)N * K * 2 * copies / best / 1e9);
  return 0;
}

template <int NT, bool PATB>
static int run(const u32x4* buf, int N, int K, int waves, unsigned* out, int copies, size_t stride16) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = N / 16 / NT, KT = K / 32;
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int c = 0; c < copies; ++c) rd<NT, PATB><<<blocks, waves * 64>>>(buf + c * stride16, KT, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("N=%6d K=%6d NT=%d waves=%2d pattern %c : %7.1f us/launch  %.2f TB/s\n", N, K, NT, waves, PATB ? 'B' : 'A',
         best * 1e3 / copies, (double)N * K * 2 * copies / best / 1e9);
  return 0;
}

int main() {
  const size_t bytes = (size_t)4 << 30;
  u32x4* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 1, bytes));
  struct { int N, K; } shapes[] = {{57344, 8192}, {8192, 28672}, {8192, 8192}, {10240, 8192}, {28672, 4096},
                                   {2048, 8192}, {16384, 2048}, {3072, 2048}, {2048, 2048}, {1280, 8192}, {7168, 8192}, {8192, 3584}};
  for (auto s : shapes) {
    const size_t mat16 = (size_t)s.N * s.K * 2 / 16;
    int copies = (int)(bytes / 16 / mat16);
    if (copies > 64) copies = 64;
    run<4, false>(buf, s.N, s.K, 8, out, copies, mat16);
    run<4, true>(buf, s.N, s.K, 8, out, copies, mat16);
    run<2, false>(buf, s.N, s.K, 16, out, copies, mat16);
    run<2, true>(buf, s.N, s.K, 16, out, copies, mat16);
    run<1, false>(buf, s.N, s.K, 8, out, copies, mat16);
    runc<4>(buf, s.N, s.K, 8, out, copies, mat16);
    runc<2>(buf, s.N, s.K, 16, out, copies, mat16);
    runc<2>(buf, s.N, s.K, 8, out, copies, mat16);
    runc<1>(buf, s.N, s.K, 8, out, copies, mat16);
    runc<1>(buf, s.N, s.K, 4, out, copies, mat16);
  }
  return 0;
}
