// What do scattered map-row stores cost beside a streaming read?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/store_probe
// Skeleton of the streaming kernel's direct update: 19200 blocks of 256 threads, every block reads
// 40 KB of a class-major tensor (40 plane loads per thread) and then reads / writes the rows of ~14 map
// cells, one cell per 16-lane row (or per wave), cells dense in a region of the map but visited in
// random order.  Variants differ in the row stride of the map and in what is read and written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

constexpr int kE = 16, kN = 480 * 640, kC = 40, kCellsPerEnv = 480 * 480, kPerBlock = 14, kBlocksPerEnv = kN / 256;

// STRIDE: floats per map row.  RD: 0 none, 1 the 40 floats.  WR: 0 none, 1 the 40 floats (3 dword stores per
// 16-lane row), 2 the whole stride per 16-lane row (dword stores), 3 the whole stride as float4 per lane
// (STRIDE / 4 lanes of the row), 4 one wave per cell (STRIDE == 64: one dword per lane).
template <int STRIDE, int RD, int WR, int STREAM, int AUXW = 0, int AUXR = 2, int AUXO = -1>
__global__ __launch_bounds__(256) void k_probe(const float *__restrict__ in, float *__restrict__ out, float *map,
                                               const int *__restrict__ cells) {
    const int e = blockIdx.y, t = threadIdx.x, lane = t & 63, sub = t & 15;
    float acc = 0.0f;
    if (STREAM) {
        const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(in + (size_t)e * kC * kN), 0, kC * kN * 4, 0x00020000);
        float x[kC];
#pragma unroll
        for (int c = 0; c < kC; ++c)
            x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frame, (blockIdx.x * 256 + t) * 4, c * kN * 4, AUXR));
#pragma unroll
        for (int c = 0; c < kC; ++c) acc += x[c];
        if (AUXO < 0) {
            out[(size_t)e * kN + blockIdx.x * 256 + t] = acc;
        } else {
            // three output planes like stage 1's (semantic, uncertainty, confidence)
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(out, 0, 3 * kE * kN * 4, 0x00020000);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc + (float)k), ors,
                                                      ((k * kE + e) * kN + blockIdx.x * 256 + t) * 4, 0, AUXO);
        }
    }
    const int *mine = cells + ((size_t)e * kBlocksPerEnv + blockIdx.x) * 16;
    float *env_map = map + (size_t)e * kCellsPerEnv * STRIDE;
    if (WR == 4) {
        // one wave per cell, 4 cells per wave in turn
        for (int j = (t >> 6); j < kPerBlock; j += 4) {
            float *row = env_map + (size_t)mine[j] * STRIDE;
            float v = RD ? row[lane] : 0.0f;
            row[lane] = v + acc + 1.0f;
        }
        return;
    }
    // a 16-lane row per cell: 16 rows per block, kPerBlock of them live
    const int r = t >> 4;
    if (r >= kPerBlock) return;
    float *row = env_map + (size_t)mine[r] * STRIDE;
    float mv[4] = {0, 0, 0, 0};
    if (RD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) mv[k] = row[(sub + 16 * k) < kC ? sub + 16 * k : 0];
    }
    if (WR == 1 && AUXW) {
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(env_map, 0, kCellsPerEnv * STRIDE * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (sub + 16 * k < kC)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mv[k] + acc + 1.0f), mrs,
                                                      (mine[r] * STRIDE + sub + 16 * k) * 4, 0, AUXW);
    } else if (WR == 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) if (sub + 16 * k < kC) row[sub + 16 * k] = mv[k] + acc + 1.0f;
    } else if (WR == 2) {
#pragma unroll
        for (int k = 0; k < STRIDE / 16; ++k) row[sub + 16 * k] = mv[k & 3] + acc + 1.0f;
    } else if (WR == 3) {
        if (sub < STRIDE / 4) reinterpret_cast<float4 *>(row)[sub] = make_float4(mv[0] + acc, mv[1], mv[2], 1.0f);
    } else if (RD) {
        if (mv[0] + mv[1] + mv[2] == 12345.678f) out[0] = 1.0f;
    }
}

template <int STRIDE, int RD, int WR, int STREAM, int AUXW = 0, int AUXR = 2, int AUXO = -1>
static void run(const char *name, const float *in, float *out, float *map, const int *cells) {
    dim3 grid(kBlocksPerEnv, kE);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_probe<STRIDE, RD, WR, STREAM, AUXW, AUXR, AUXO>), grid, dim3(256), 0, 0, in, out, map, cells);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_probe<STRIDE, RD, WR, STREAM, AUXW, AUXR, AUXO>), grid, dim3(256), 0, 0, in, out, map, cells);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-58s %8.1f us\n", name, ms * 1e3 / reps);
    fflush(stdout);
}

int main(int argc, char **) {
    float *in, *out, *map;
    int *cells;
    const size_t n_in = (size_t)kE * kC * kN;
    CHECK(hipMalloc(&in, n_in * 4));
    CHECK(hipMalloc(&out, (size_t)3 * kE * kN * 4));
    CHECK(hipMalloc(&map, (size_t)kE * kCellsPerEnv * 64 * 4));
    CHECK(hipMemset(in, 0, n_in * 4));
    CHECK(hipMemset(map, 0, (size_t)kE * kCellsPerEnv * 64 * 4));
    // cells: a dense region of 160 x 150 cells per env, kPerBlock * blocks of them, visited in random order
    std::vector<int> h((size_t)kE * kBlocksPerEnv * 16, 0);
    std::mt19937 rng(1);
    for (int e = 0; e < kE; ++e) {
        std::vector<int> region;
        for (int r = 100; r < 260; ++r) for (int c = 100; c < 250; ++c) region.push_back(r * 480 + c);
        std::shuffle(region.begin(), region.end(), rng);
        region.resize((size_t)kBlocksPerEnv * kPerBlock);   // 16800 of 24000
        for (int b = 0; b < kBlocksPerEnv; ++b)
            for (int j = 0; j < kPerBlock; ++j) h[((size_t)e * kBlocksPerEnv + b) * 16 + j] = region[(size_t)b * kPerBlock + j];
    }
    CHECK(hipMalloc(&cells, h.size() * 4));
    CHECK(hipMemcpy(cells, h.data(), h.size() * 4, hipMemcpyHostToDevice));
#define RUN(S, R, W, ST) run<S, R, W, ST>("stride " #S " rd " #R " wr " #W " stream " #ST, in, out, map, cells)
#define RUNA(S, R, W, ST, AW, AR) run<S, R, W, ST, AW, AR>("stride " #S " rd " #R " wr " #W " stream " #ST " auxw " #AW " auxr " #AR, in, out, map, cells)
#define RUNO(W, AW, AO) run<40, W, W, 1, AW, 2, AO>("3 out planes, rmw " #W " auxw " #AW " auxo " #AO, in, out, map, cells)
#define RUNS(S, R, W, AW) run<S, R, W, 1, AW, 2, 16>("3 out planes (sc1), stride " #S " rd " #R " wr " #W " auxw " #AW, in, out, map, cells)
    if (argc > 1) {
        // second table: row strides and store widths beside the full stage-1 traffic
        RUNS(40, 0, 0, 0);
        RUNS(40, 1, 0, 0);
        RUNS(40, 1, 1, 0);
        RUNS(40, 0, 1, 0);
        RUNS(40, 1, 1, 2);
        RUNS(40, 1, 1, 16);
        RUNS(40, 1, 1, 17);
        RUNS(40, 1, 3, 0);
        RUNS(48, 1, 1, 0);
        RUNS(48, 1, 2, 0);
        RUNS(48, 1, 3, 0);
        RUNS(64, 1, 1, 0);
        RUNS(64, 1, 2, 0);
        RUNS(64, 1, 3, 0);
        RUNS(64, 1, 4, 0);
        return 0;
    }
    RUNA(40, 0, 0, 1, 0, 2);
    RUNO(0, 0, 0);
    RUNO(0, 0, 1);
    RUNO(0, 0, 2);
    RUNO(0, 0, 3);
    RUNO(0, 0, 16);
    RUNO(0, 0, 17);
    RUNO(0, 0, 18);
    RUNO(1, 0, 2);
    RUNO(1, 16, 2);
    RUNO(1, 16, 16);
    RUNO(1, 0, 16);
    RUNO(1, 16, 0);
    return 0;
}
