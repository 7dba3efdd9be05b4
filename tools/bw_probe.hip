// Read-bandwidth probe for the class-major streaming pattern (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/bw_probe
// Variants: linear float4 sweep; plane-strided reads (one pixel group per thread, C planes)
// with 1/2/4 pixels per thread, with and without the softmax arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)


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

// logits-like input: hashed values in [-6, 6)
__global__ void k_fill(float *p, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h >> 8) * (12.0f / 16777216.0f) - 6.0f;
    }
}

template <int C, int PX, int MATH, int NT>
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

// The engine's stage-1 arithmetic step by step (FLAGS bits): 1 = buffer loads with scalar plane
// offsets, 2 = the reference-rounding softmax (exp2 / log2 builtins, no contraction) with argmax,
// 4 = three per-pixel outputs (nontemporal) instead of one, 8 = class table lookup (global gather)
template <int C, int FLAGS>
__device__ __forceinline__ void real_body(const float *in, float *out, int *out_i, float *out2, const int *lut, int N, float inv_t) {
    const int e = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= N) return;
    float x[C];
    if (FLAGS & 1) {
        const int plane_bytes = N * 4;
        const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(in + ((size_t)e * C) * N), 0, C * plane_bytes, 0x00020000);
#pragma unroll
        for (int c = 0; c < C; ++c)
            x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frame, pix * 4, c * plane_bytes, 2));
    } else {
        const float *src = in + (size_t)e * C * N + pix;
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = __builtin_nontemporal_load(src + (size_t)c * N);
    }
    float h = 0.0f, conf = 0.0f;
    int bi = 0;
    if (FLAGS & 16) {
        // the same roundings two classes at a time (v_pk_mul / v_pk_add / v_pk_fma), two partial sums
#pragma clang fp contract(off)
        typedef float f2 __attribute__((ext_vector_type(2)));
        static_assert(C % 2 == 0, "pairs");
        f2 xv[C / 2];
        float m = -INFINITY;
#pragma unroll
        for (int q = 0; q < C / 2; ++q) {
            xv[q] = f2{x[2 * q], x[2 * q + 1]} * f2{inv_t, inv_t};
            m = fmaxf(m, fmaxf(xv[q].x, xv[q].y));
        }
        f2 s2 = {0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < C / 2; ++q) {
            const f2 d = (xv[q] - f2{m, m}) * f2{1.44269504088896341f, 1.44269504088896341f};
            xv[q] = f2{__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
            s2 += xv[q];
        }
        const float inv_s = 1.0f / (s2.x + s2.y);
        f2 h2 = {0.0f, 0.0f};
#pragma unroll
        for (int q = C / 2 - 1; q >= 0; --q) {
            xv[q] *= f2{inv_s, inv_s};
            const f2 a = xv[q] + f2{1e-6f, 1e-6f};
            const f2 l = {__builtin_amdgcn_logf(a.x), __builtin_amdgcn_logf(a.y)};
            h2 = __builtin_elementwise_fma(xv[q], l, h2);
            bi = xv[q].y == inv_s ? 2 * q + 1 : bi;
            bi = xv[q].x == inv_s ? 2 * q : bi;
        }
        h = h2.x + h2.y;
        conf = inv_s;
    } else if (FLAGS & 2) {
#pragma clang fp contract(off)
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) { x[c] *= inv_t; m = fmaxf(m, x[c]); }
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) { x[c] = __builtin_amdgcn_exp2f((x[c] - m) * 1.44269504088896341f); s += x[c]; }
        const float inv_s = 1.0f / s;
#pragma unroll
        for (int c = C - 1; c >= 0; --c) {
            x[c] *= inv_s;
            h = __builtin_fmaf(x[c], __builtin_amdgcn_logf(x[c] + 1e-6f), h);
            bi = x[c] == inv_s ? c : bi;
        }
        conf = inv_s;
    } else {
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) { x[c] *= 0.3125f; m = fmaxf(m, x[c]); }
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) { x[c] = __expf(x[c] - m); s += x[c]; }
        const float inv = 1.0f / s;
#pragma unroll
        for (int c = 0; c < C; ++c) { x[c] *= inv; h += x[c] * __logf(x[c] + 1e-6f); }
        conf = inv;
    }
    if (FLAGS & 8) bi = lut[bi];
    const size_t o = (size_t)e * N + pix;
    if (FLAGS & 4) {
        __builtin_nontemporal_store(h, out + o);
        __builtin_nontemporal_store(bi, out_i + o);
        __builtin_nontemporal_store(conf, out2 + o);
    } else {
        out[o] = h + conf + (float)bi;
    }
}

template <int C, int FLAGS, int W>
__global__ __launch_bounds__(256, W) void k_real(const float *in, float *out, int *out_i, float *out2, const int *lut, int N, float inv_t) {
    real_body<C, FLAGS>(in, out, out_i, out2, lut, N, inv_t);
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

template <int C>
static void run_planes(int E) {
    const int N = 640 * 480;
    const size_t n = (size_t)E * C * N;
    float *in, *out;
    CHECK(hipMalloc(&in, n * 4));
    CHECK(hipMalloc(&out, (size_t)E * N * 4));
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, in, n);
    CHECK(hipDeviceSynchronize());
    const double gb = n * 4 / 1e9;
    float us;
#define RUN(PX, MATH, NT) \
    us = time_it([&] { hipLaunchKernelGGL((k_planes<C, PX, MATH, NT>), dim3(N / PX / 256, E), dim3(256), 0, 0, in, out, N); }); \
    printf("C=%d planes PX=%d math=%d nt=%d: %7.1f us %7.1f GB/s\n", C, PX, MATH, NT, us, gb / us * 1e6);
    RUN(1, 0, 1) RUN(1, 1, 1)
#undef RUN
    int *out_i, *lut; float *out2;
    CHECK(hipMalloc(&out_i, (size_t)E * N * 4)); CHECK(hipMalloc(&out2, (size_t)E * N * 4)); CHECK(hipMalloc(&lut, 1024)); CHECK(hipMemset(lut, 0, 1024));
#define REAL(F, W) \
    us = time_it([&] { hipLaunchKernelGGL((k_real<C, F, W>), dim3(N / 256, E), dim3(256), 0, 0, in, out, out_i, out2, lut, N, 0.3125f); }); \
    printf("C=%d real flags=%2d minwaves=%d: %7.1f us %7.1f GB/s\n", C, F, W, us, gb / us * 1e6);
    REAL(0, 2) REAL(2, 2) REAL(7, 2) REAL(15, 2) REAL(15, 3) REAL(29, 2) REAL(29, 3)
#undef REAL
    CHECK(hipFree(in)); CHECK(hipFree(out));
}

int main() {
    run_planes<150>(16);
    run_planes<40>(16);
    return 0;
}
