// Microbenchmark: cost of a device-wide barrier inside a persistent kernel vs a kernel boundary in a hipGraph.
// hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += nblocks;
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);        // agent scope by default for global atomics
        long spins = 0;
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < epoch && ++spins < 100000000L) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) persistent(unsigned* counter, float* data, int rounds, int work) {
    unsigned epoch = 0;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < work; ++i) acc += data[(blockIdx.x * blockDim.x + threadIdx.x + i * 65536 + r * 7) & 0xFFFFF];
        grid_barrier(counter, gridDim.x, epoch);
    }
    if (acc == 12345.678f) data[0] = acc;
}

__global__ void __launch_bounds__(1024) tiny(float* data, int work, int r) {
    float acc = 0.f;
    for (int i = 0; i < work; ++i) acc += data[(blockIdx.x * blockDim.x + threadIdx.x + i * 65536 + r * 7) & 0xFFFFF];
    if (acc == 12345.678f) data[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    unsigned* counter; float* data;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&data, 4 << 20)); CK(hipMemset(data, 0, 4 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 200;
    for (int blocks : {256, 512}) for (int threads : {256, 1024}) for (int work : {0, 4}) {
        if (blocks * threads > 256 * 2048) continue;
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipEventRecord(e0, s));
            persistent<<<blocks, threads, 0, s>>>(counter, data, rounds, work);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("persistent blocks=%d threads=%d work=%d : %.2f us per round\n", blocks, threads, work, best * 1e3 / rounds);
        // same thing as a graph of `rounds` dependent tiny kernels
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int r = 0; r < rounds; ++r) tiny<<<blocks, threads, 0, s>>>(data, work, r);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("graph      blocks=%d threads=%d work=%d : %.2f us per kernel\n", blocks, threads, work, best * 1e3 / rounds);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
