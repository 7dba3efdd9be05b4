// Read-bandwidth probe for the class-major streaming pattern (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/bw_probe
// Variants: linear float4 sweep; plane-strided reads (one pixel group per thread, C planes)
// with 1/2/4 pixels per thread, with and without the softmax arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

constexpr int C = 40;

__global__ __launch_bounds__(256) void k_linear(const float4 *in, float *out, size_t n4) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 q = in[i];
        acc += q.x + q.y + q.z + q.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// each block sweeps a contiguous chunk, 4 independent loads in flight per thread
__global__ __launch_bounds__(256) void k_linear_chunk(const float4 *in, float *out, size_t n4) {
    const size_t per = n4 / gridDim.x;
    const float4 *p = in + per * blockIdx.x;
    float acc = 0.0f;
    for (size_t i = threadIdx.x; i + 768 < per; i += 1024) {
        const float4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int PX, int MATH, int NT>
__global__ __launch_bounds__(256) void k_planes(const float *in, float *out, int N) {
    typedef float VecF __attribute__((ext_vector_type(PX)));
    const int e = blockIdx.y;
    const int pix = (blockIdx.x * 256 + threadIdx.x) * PX;
    if (pix >= N) return;
    const float *src = in + (size_t)e * C * N + pix;
    float x[PX][C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const VecF *ptr = reinterpret_cast<const VecF *>(src + (size_t)c * N);
        const VecF q = NT ? __builtin_nontemporal_load(ptr) : *ptr;
#pragma unroll
        for (int p = 0; p < PX; ++p) x[p][c] = q[p];
    }
    VecF r;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        if (MATH) {
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < C; ++c) { x[p][c] *= 0.3125f; m = fmaxf(m, x[p][c]); }
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) { x[p][c] = __expf(x[p][c] - m); s += x[p][c]; }
            const float inv = 1.0f / s;
            float h = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) { x[p][c] *= inv; h += x[p][c] * __logf(x[p][c] + 1e-6f); }
            r[p] = h;
        } else {
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) s += x[p][c];
            r[p] = s;
        }
    }
    *reinterpret_cast<VecF *>(out + (size_t)e * N + pix) = r;
}

template <typename F>
static float time_it(F launch, int reps = 20) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipGetLastError());
    return ms / reps * 1e3f;
}

int main() {
    const int E = 16, N = 640 * 480;
    const size_t n = (size_t)E * C * N;
    float *in, *out;
    CHECK(hipMalloc(&in, n * 4));
    CHECK(hipMalloc(&out, (size_t)E * N * 4));
    CHECK(hipMemset(in, 0, n * 4));
    const double gb = n * 4 / 1e9;
    float us;
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        us = time_it([&] { hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, (const float4 *)in, out, n / 4); });
        printf("linear float4 grid-stride %5d blocks: %7.1f us %7.1f GB/s\n", blocks, us, gb / us * 1e6);
    }
    for (int blocks : {1024, 2048, 4800}) {
        us = time_it([&] { hipLaunchKernelGGL(k_linear_chunk, dim3(blocks), dim3(256), 0, 0, (const float4 *)in, out, n / 4); });
        printf("linear float4 chunked     %5d blocks: %7.1f us %7.1f GB/s\n", blocks, us, gb / us * 1e6);
    }
#define RUN(PX, MATH, NT) \
    us = time_it([&] { hipLaunchKernelGGL((k_planes<PX, MATH, NT>), dim3(N / PX / 256, E), dim3(256), 0, 0, in, out, N); }); \
    printf("planes PX=%d math=%d nt=%d: %7.1f us %7.1f GB/s\n", PX, MATH, NT, us, gb / us * 1e6);
    RUN(1, 0, 0) RUN(2, 0, 0) RUN(4, 0, 0)
    RUN(1, 0, 1) RUN(2, 0, 1) RUN(4, 0, 1)
    RUN(1, 1, 0) RUN(2, 1, 0) RUN(4, 1, 0)
    RUN(1, 1, 1) RUN(2, 1, 1) RUN(4, 1, 1)
    return 0;
}
