// How fast can one 1024-thread block per CU read the pixel records the geometry pass has just written?
// (run on the GPU box)   hipcc --offload-arch=gfx950 -O3 tools/rec_probe.hip -o tools/rec_probe
// 16 envs x 16 tables x 300 geometry blocks x 64 records: heights (8 B) and words (4 B) in two arrays,
// written by a 19200-block kernel, then read by 256 blocks (one per table) the way the winner kernel's
// phase a reads row 0: wave w of a block takes the geometry blocks w, w + 16, ..., 19 loads of each
// array per thread, all issued before the first use.  Variants: load flavour, pairs of records per lane
// (16-byte loads), two 512-thread blocks per CU, a 119 KB LDS array that keeps one block per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)
constexpr int kE = 16, kT = 16, kB = 300, kKeep = 19;
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_write(u64 *h, unsigned *w) {
    // geometry block b of env e writes its 64 records of each of the 16 tables
    const int e = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    for (int part = t >> 6; part < kT; part += 4) {
        const size_t at = (((size_t)e * kT + part) * kB + b) * 64 + (t & 63);
        h[at] = 0x4000000000000000ull + at * 977u;
        w[at] = (unsigned)at * 2654435761u;
    }
}

template <int NT, int PAIR, int THREADS, int LDSKB>
__global__ __launch_bounds__(THREADS) void k_read(const u64 *__restrict__ h, const unsigned *__restrict__ w, unsigned *out) {
    __shared__ unsigned table[LDSKB * 256];
    constexpr int kWaves = THREADS / 64;
    constexpr int kParts = 1024 / THREADS;          // blocks per table
    const int e = blockIdx.y, part = blockIdx.x / kParts, sub = blockIdx.x % kParts;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t tbase = ((size_t)e * kT + part) * kB * 64;
    u64 acc = 0;
    unsigned accw = 0;
    if (PAIR) {
        // a wave-load covers row 0 of two neighbouring geometry blocks: 16 bytes of heights per lane
        u64x2 hv[10];
        u32x2 wv[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int pair = wave + kWaves * (u * kParts + sub);
            const size_t at = tbase + (size_t)pair * 128 + 2 * lane;
            if (pair * 2 < kB) {
                if (NT) { hv[u] = __builtin_nontemporal_load((const u64x2 *)(h + at)); wv[u] = __builtin_nontemporal_load((const u32x2 *)(w + at)); }
                else { hv[u] = *(const u64x2 *)(h + at); wv[u] = *(const u32x2 *)(w + at); }
            } else { hv[u] = 0; wv[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) { acc += hv[u].x ^ hv[u].y; accw += wv[u].x ^ wv[u].y; }
    } else {
        u64 hv[kKeep];
        unsigned wv[kKeep];
#pragma unroll
        for (int u = 0; u < kKeep; ++u) {
            const int b = wave + kWaves * (u * kParts + sub);
            const size_t at = tbase + (size_t)b * 64 + lane;
            if (b < kB) {
                if (NT) { hv[u] = __builtin_nontemporal_load(h + at); wv[u] = __builtin_nontemporal_load(w + at); }
                else { hv[u] = h[at]; wv[u] = w[at]; }
            } else { hv[u] = 0; wv[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < kKeep; ++u) { acc += hv[u]; accw += wv[u]; }
    }
    table[(threadIdx.x * 33) % (LDSKB * 256)] = accw;
    __syncthreads();
    if (acc == 12345 && table[threadIdx.x] == 7) out[0] = accw;
}

template <int NT, int PAIR, int THREADS, int LDSKB>
static void run(const char *name, u64 *h, unsigned *w, unsigned *out) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float total = 0;
    const int reps = 20;
    for (int i = 0; i < reps + 2; ++i) {
        hipLaunchKernelGGL(k_write, dim3(kB, kE), dim3(256), 0, 0, h, w);
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k_read<NT, PAIR, THREADS, LDSKB>), dim3(kT * (1024 / THREADS), kE), dim3(THREADS), 0, 0, h, w, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (i >= 2) total += ms;
    }
    printf("%-64s %7.2f us  (%.2f TB/s)\n", name, total * 1e3 / reps, (double)kE * kT * kB * 64 * 12 / (total * 1e-3 / reps) / 1e12);
    fflush(stdout);
}

int main() {
    u64 *h; unsigned *w, *out;
    const size_t n = (size_t)kE * kT * kB * 64 + 1024;
    CHECK(hipMalloc(&h, n * 8)); CHECK(hipMalloc(&w, n * 4)); CHECK(hipMalloc(&out, 4096));
    run<0, 0, 1024, 116>("1024 threads, one record per lane, plain loads, 116 KB LDS", h, w, out);
    run<1, 0, 1024, 116>("1024 threads, one record per lane, nontemporal loads, 116 KB LDS", h, w, out);
    run<0, 1, 1024, 116>("1024 threads, two records per lane (16-byte loads), 116 KB LDS", h, w, out);
    run<1, 1, 1024, 116>("1024 threads, two records per lane, nontemporal, 116 KB LDS", h, w, out);
    run<0, 0, 512, 58>("2 x 512 threads per table, one record per lane, 58 KB LDS", h, w, out);
    run<0, 1, 512, 58>("2 x 512 threads per table, two records per lane, 58 KB LDS", h, w, out);
    run<0, 0, 256, 29>("4 x 256 threads per table, one record per lane, 29 KB LDS", h, w, out);
    run<0, 1, 256, 29>("4 x 256 threads per table, two records per lane, 29 KB LDS", h, w, out);
    return 0;
}
