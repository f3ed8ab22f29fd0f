// Round 6 (end): can the resident segments (csrc/chain.hip, tree_segment.hip: ordinary launches whose 256 workgroups wait on each other)
// be launched COOPERATIVELY -- the runtime then guarantees co-residency or refuses the launch -- inside the hipGraphs the engine replays,
// and does a cooperative launch still run beside another stream's kernels?
//   1. hipLaunchCooperativeKernel under stream capture -> instantiate -> replay: error codes
//   2. a cooperative launch (256 x 512 threads, each workgroup spins ~40 us) on stream A beside an ordinary long kernel on stream B:
//      wall time of both together against each alone
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/coop_probe profiles/micro/coop_probe.hip && /tmp/coop_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(512) spin_kernel(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, 1u);
}

#define CK(x) do { hipError_t e_ = (x); printf("  %-58s -> %s\n", #x, hipGetErrorName(e_)); } while (0)

static float ms_between(hipEvent_t a, hipEvent_t b) { float ms = 0; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  unsigned* out;
  hipMalloc(&out, 4);
  hipMemset(out, 0, 4);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  unsigned long long ticks = 4000;          // 100 MHz clock: 40 us
  void* args[] = {&ticks, &out};
  int dev = 0, coop = 0;
  hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
  printf("cooperative launch supported: %d\n", coop);
  printf("1. plain cooperative launch\n");
  CK(hipLaunchCooperativeKernel((const void*)spin_kernel, dim3(256), dim3(512), args, 0, sa));
  CK(hipStreamSynchronize(sa));
  printf("2. under stream capture\n");
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
  CK(hipLaunchCooperativeKernel((const void*)spin_kernel, dim3(256), dim3(512), args, 0, sa));
  CK(hipStreamEndCapture(sa, &g));
  if (g) {
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    if (ge) {
      CK(hipGraphLaunch(ge, sa));
      CK(hipStreamSynchronize(sa));
    }
  }
  (void)hipGetLastError();
  printf("3. concurrency: cooperative on A beside an ordinary kernel on B (each spins 40 us; 256 + 128 workgroups of 512 threads)\n");
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {          // 0: A coop alone, 1: A coop + B ordinary, 2: A ordinary + B ordinary
    hipDeviceSynchronize();
    hipEventRecord(e0, sa);
    hipStreamWaitEvent(sb, e0, 0);
    for (int i = 0; i < 10; ++i) {
      if (mode == 2) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, sa, ticks, out);
      else hipLaunchCooperativeKernel((const void*)spin_kernel, dim3(256), dim3(512), args, 0, sa);
      if (mode >= 1) hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(512), 0, sb, ticks, out);
    }
    hipEventRecord(e1, sb);
    hipStreamWaitEvent(sa, e1, 0);
    hipEventRecord(e1, sa);
    hipDeviceSynchronize();
    printf("  mode %d: %.1f us per pair of launches\n", mode, ms_between(e0, e1) * 100.f);
  }
  return 0;
}
