// How fast can map rows be read, changed and written back, by lane mapping?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/row_probe.hip -o tools/row_probe
// Skeleton of the rule update kernel: every env has a list of cells in ascending order (17 thousand of the
// 24 thousand cells of a 160 x 150 region of a 480 x 480 map), a staged row per list slot (contiguous) and a
// map row per cell (C floats, cell-major).  A block takes 64 consecutive list slots.  Per slot: read the
// staged row and the map row, add, write the map row.  Mappings:
//   M=16  a 16-lane row of a wave per cell, lane <-> V floats (V = 1, 2), ceil(C / 16 V) steps; four cells
//         per wave-instruction, four batches per wave, NB batches in flight
//   M=32  32 lanes per cell, lane <-> four floats (loads that are 4-byte aligned only), two cells per wave-instruction
//   M=64  the whole wave per cell, lane <-> two floats, ceil(C / 128) steps, four cells in flight per wave
// and what is done: RD = 1 reads only, WR = 1 reads and writes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

constexpr int kE = 16, kCellsPerEnv = 480 * 480, kList = 16800;
typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));

template <int C, int M, int V, int NB, int WR, int NT = 0>
__global__ __launch_bounds__(256) void k_rows(const float *__restrict__ staged, float *map, const int *__restrict__ cells) {
    const int e = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int *list = cells + (size_t)e * kList;
    const float *st = staged + (size_t)e * kList * C;
    float *env_map = map + (size_t)e * kCellsPerEnv * C;
    const int s0 = blockIdx.x * 64 + wave * 16;      // this wave's 16 list slots
    if (s0 >= kList) return;
    float sink = 0.0f;
    if (M == 16) {
        constexpr int K = (C + 16 * V - 1) / (16 * V);
        const int sub = lane & 15, row = lane >> 4;
        typedef float VecF __attribute__((ext_vector_type(V)));
        for (int b0 = 0; b0 < 4; b0 += NB) {
            VecF p[NB][K], m[NB][K];
            int cell[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int s = s0 + 4 * (b0 + b) + row;
                const bool on = s < kList;
                cell[b] = on ? list[s] : 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = V * (sub + 16 * k), cs = c < C ? c : 0;
                    if (NT == 0) {
                        p[b][k] = *reinterpret_cast<const VecF *>(st + (size_t)(on ? s : 0) * C + cs);
                        m[b][k] = *reinterpret_cast<const VecF *>(env_map + (size_t)cell[b] * C + cs);
                    } else if (NT == 1) {
                        p[b][k] = __builtin_nontemporal_load(reinterpret_cast<const VecF *>(st + (size_t)(on ? s : 0) * C + cs));
                        m[b][k] = __builtin_nontemporal_load(reinterpret_cast<const VecF *>(env_map + (size_t)cell[b] * C + cs));
                    } else if constexpr (V == 2) {
                        // device-scope loads (sc1): served by L2, no line taken in L1
                        typedef unsigned long long u64;
                        const u64 pa = __hip_atomic_load(reinterpret_cast<const u64 *>(st + (size_t)(on ? s : 0) * C + cs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const u64 ma = __hip_atomic_load(reinterpret_cast<const u64 *>(env_map + (size_t)cell[b] * C + cs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        p[b][k] = __builtin_bit_cast(VecF, pa);
                        m[b][k] = __builtin_bit_cast(VecF, ma);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int s = s0 + 4 * (b0 + b) + row;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = V * (sub + 16 * k);
                    m[b][k] += p[b][k];
                    if (WR == 1) { if (s < kList && c < C) *reinterpret_cast<VecF *>(env_map + (size_t)cell[b] * C + c) = m[b][k]; }
                    else if (WR == 2) { if (s < kList && c < C) __builtin_nontemporal_store(m[b][k], reinterpret_cast<VecF *>(env_map + (size_t)cell[b] * C + c)); }
                    else if (WR == 3) {
                        if constexpr (V == 2) {
                            if (s < kList && c < C)
                                __hip_atomic_store(reinterpret_cast<unsigned long long *>(env_map + (size_t)cell[b] * C + c),
                                                   __builtin_bit_cast(unsigned long long, m[b][k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    else sink += m[b][k][0];
                }
            }
        }
    } else if (M == 32) {
        constexpr int K = (C + 127) / 128;
        const int sub = lane & 31, row = lane >> 5;
        for (int b0 = 0; b0 < 8; b0 += NB) {
            f4 p[NB][K], m[NB][K];
            int cell[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int s = s0 + 2 * (b0 + b) + row;
                const bool on = s < kList;
                cell[b] = on ? list[s] : 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = 4 * (sub + 32 * k), cs = c + 3 < C ? c : 0;
                    p[b][k] = *reinterpret_cast<const f4 *>(st + (size_t)(on ? s : 0) * C + cs);
                    m[b][k] = *reinterpret_cast<const f4 *>(env_map + (size_t)cell[b] * C + cs);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int s = s0 + 2 * (b0 + b) + row;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = 4 * (sub + 32 * k);
                    m[b][k] += p[b][k];
                    // (a row's last two floats, when C is no multiple of four, are left out: a probe)
                    if (WR) { if (s < kList && c + 3 < C) *reinterpret_cast<f4 *>(env_map + (size_t)cell[b] * C + c) = m[b][k]; }
                    else sink += m[b][k][0];
                }
            }
        }
    } else {
        constexpr int K = (C + 127) / 128;
        for (int b0 = 0; b0 < 16; b0 += NB) {
            f2 p[NB][K], m[NB][K];
            int cell[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int s = s0 + b0 + b;
                const bool on = s < kList;
                cell[b] = on ? list[s] : 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = 2 * (lane + 64 * k), cs = c < C ? c : 0;
                    p[b][k] = *reinterpret_cast<const f2 *>(st + (size_t)(on ? s : 0) * C + cs);
                    m[b][k] = *reinterpret_cast<const f2 *>(env_map + (size_t)cell[b] * C + cs);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int s = s0 + b0 + b;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = 2 * (lane + 64 * k);
                    m[b][k] += p[b][k];
                    if (WR) { if (s < kList && c < C) *reinterpret_cast<f2 *>(env_map + (size_t)cell[b] * C + c) = m[b][k]; }
                    else sink += m[b][k][0];
                }
            }
        }
    }
    if (!WR && sink == 12345.678f) map[0] = sink;
}

template <int C, int M, int V, int NB, int WR, int NT = 0>
static void run(const char *name, const float *staged, float *map, const int *cells) {
    dim3 grid((kList + 63) / 64, kE);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    // every repetition starts from cold caches: 1 GiB is written in between (the update kernel runs
    // behind a kernel that streams gigabytes), and only the kernel itself is timed
    static char *flush = nullptr;
    if (!flush) CHECK(hipMalloc(&flush, (size_t)1 << 30));
    const int reps = 10;
    float ms = 0;
    for (int i = 0; i < reps + 2; ++i) {
        CHECK(hipMemsetAsync(flush, i, (size_t)1 << 30, 0));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k_rows<C, M, V, NB, WR, NT>), grid, dim3(256), 0, 0, staged, map, cells);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float one = 0;
        CHECK(hipEventElapsedTime(&one, a, b));
        if (i >= 2) ms += one;
    }
    const double us = ms * 1e3 / reps;
    const double bytes = (double)kE * kList * C * 4 * (WR ? 3 : 2);   // (WR: 1 plain, 2 nontemporal, 3 device-scope stores)
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, bytes / us / 1e6);
    fflush(stdout);
}

static bool argc_short = false;
template <int C>
static void table() {
    float *staged, *map;
    int *cells;
    CHECK(hipMalloc(&staged, (size_t)kE * kList * C * 4));
    CHECK(hipMalloc(&map, (size_t)kE * kCellsPerEnv * C * 4));
    CHECK(hipMemset(staged, 0, (size_t)kE * kList * C * 4));
    CHECK(hipMemset(map, 0, (size_t)kE * kCellsPerEnv * C * 4));
    std::vector<int> h((size_t)kE * kList);
    std::mt19937 rng(1);
    for (int e = 0; e < kE; ++e) {
        std::vector<int> region;
        for (int r = 100; r < 260; ++r) for (int c = 100; c < 250; ++c) region.push_back(r * 480 + c);
        std::shuffle(region.begin(), region.end(), rng);
        region.resize(kList);
        std::sort(region.begin(), region.end());
        std::copy(region.begin(), region.end(), h.begin() + (size_t)e * kList);
    }
    CHECK(hipMalloc(&cells, h.size() * 4));
    CHECK(hipMemcpy(cells, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("C = %d\n", C);
#define RUN(M, V, NB, WR) run<C, M, V, NB, WR>("lanes/row " #M " V " #V " in flight " #NB " write " #WR, staged, map, cells)
#define RUNT(M, V, NB, WR, NT) run<C, M, V, NB, WR, NT>("lanes/row " #M " V " #V " in flight " #NB " write " #WR " loads nt/sc1 " #NT, staged, map, cells)
    RUNT(16, 2, 2, 0, 1); RUNT(16, 2, 2, 1, 1); RUNT(16, 2, 2, 2, 1); RUNT(16, 2, 2, 3, 1);
    RUNT(16, 2, 2, 2, 0); RUNT(16, 2, 2, 3, 0);
    if (argc_short) return;
    RUN(16, 1, 2, 0); RUN(16, 1, 2, 1);
    RUN(16, 2, 2, 0); RUN(16, 2, 2, 1);
    RUN(16, 2, 4, 0); RUN(16, 2, 4, 1);
    RUN(32, 4, 2, 0); RUN(32, 4, 2, 1);
    RUN(32, 4, 4, 0); RUN(32, 4, 4, 1);
    RUN(64, 2, 4, 0); RUN(64, 2, 4, 1);
    RUN(64, 2, 8, 0); RUN(64, 2, 8, 1);
    CHECK(hipFree(staged)); CHECK(hipFree(map)); CHECK(hipFree(cells));
}

int main(int argc, char **) {
    argc_short = argc > 1;
    table<150>();
    table<40>();
    return 0;
}
