// Throughput of device-scope 64-bit integer max atomics on a map-shaped table in global memory
// (run on the GPU box):  hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o tools/atomic_probe
// E envs x N pixels; pixel i of env e hits cell hash(i) % M of the env's table (M cells per env).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

__device__ inline unsigned hash32(unsigned h) { h *= 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return h; }

template <int MODE>   // 0: atomic max without return, 1: plain store (no atomic), 2: 4 pixels per thread
__global__ __launch_bounds__(256) void k_max(unsigned long long *table, int N, int M, int hot) {
    const int e = blockIdx.y;
    unsigned long long *t = table + (size_t)e * M;
    if (MODE == 2) {
        const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned h = hash32((unsigned)(i0 + k) + 0x9E37u * e);
            atomicMax(&t[(h >> 4) % (unsigned)hot], ((unsigned long long)h << 20) | (unsigned)(i0 + k));
        }
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const unsigned h = hash32((unsigned)i + 0x9E37u * e);
    const unsigned long long key = ((unsigned long long)h << 20) | (unsigned)i;
    unsigned long long *p = &t[(h >> 4) % (unsigned)hot];
    if (MODE == 0) atomicMax(p, key); else *p = key;
}

int main() {
    const int E = 16, N = 640 * 480, M = 480 * 480;
    unsigned long long *table;
    CHECK(hipMalloc(&table, (size_t)E * M * 8));
    CHECK(hipMemset(table, 0, (size_t)E * M * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %8.1f us  %7.1f G atomics/s\n", name, ms / 20 * 1e3f, (double)E * N / (ms / 20 * 1e-3) / 1e9);
    };
    for (int hot : {M, 16384}) {   // cells hit per env: the whole map, or the ~16 k cells a frame really touches
        printf("cells hit per env: %d\n", hot);
        run("atomicMax u64, 1 pixel / thread", [&] { hipLaunchKernelGGL(k_max<0>, dim3(N / 256, E), dim3(256), 0, 0, table, N, M, hot); });
        run("atomicMax u64, 4 pixels / thread", [&] { hipLaunchKernelGGL(k_max<2>, dim3(N / 1024, E), dim3(256), 0, 0, table, N, M, hot); });
        run("plain 8-byte store", [&] { hipLaunchKernelGGL(k_max<1>, dim3(N / 256, E), dim3(256), 0, 0, table, N, M, hot); });
    }
    CHECK(hipGetLastError());
    return 0;
}
