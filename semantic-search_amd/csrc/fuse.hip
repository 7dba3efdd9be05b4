// Stage 1 (temperature softmax + entropy + confidence, infer.py:97-118 / 302-327) fused with
// stage 4 for the probability rules (avg_prob / w_avg_prob / w2_avg_prob, mapper.py:517-569;
// log_odds, mapper.py:570-603): ONE pass over the class-major [C][H][W] tensor.
//
// Each thread owns one pixel and keeps its C values in a private LDS column (stride 257
// floats: conflict-free both when a thread walks its column and when a wave reads one
// column across classes).  Pixels that won a map cell (target[pix] >= 0, written by the
// winner kernel) are then handed to whole waves: lanes run over the classes, so the
// read-modify-write of the cell's C contiguous floats in the fused map is coalesced and the
// per-cell entropy / argmax are wavefront reductions.
#include "ssmap_internal.h"

namespace ssmap {

constexpr int kTile = 256;
constexpr int kStride = kTile + 1;

__device__ inline float wave_sum(float x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// numpy argmax order: NaN beats everything, ties go to the lower index
__device__ inline bool arg_better(float a, int ia, float b, int ib) {
    const bool an = a != a, bn = b != b;
    if (an != bn) return an;
    if (an) return ia < ib;
    return a > b || (a == b && ia < ib);
}
__device__ inline void wave_argmax(float &val, int &idx) {
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(val, o);
        const int oi = __shfl_xor(idx, o);
        if (arg_better(ov, oi, val, idx)) { val = ov; idx = oi; }
    }
}

// scipy.special.entr: -x log x, 0 at 0, -inf below 0
__device__ inline float entr(float x) {
    if (x > 0.0f) return -x * __logf(x);
    return x == 0.0f ? 0.0f : (x != x ? x : -INFINITY);
}

__device__ inline int occupancy_override(int id, float h, int use_height, int n_schema) {
    if (!use_height || n_schema != 9) return id;
    if (id == 0 || id == 1 || id == 2) {
        if (h > 0.1f) return 2;
        if (h != -INFINITY && h < 0.1f) return 1;
    }
    return id;
}

__device__ inline void commit_cell(const ssmap_env_maps &mp, const uint8_t *palette, int cell, int id) {
    mp.semantic_map[cell] = (uint32_t)id;
    const uint8_t *rgb = palette + 3 * id;
    uint8_t *dst = mp.semantic_map_color + 3ll * cell;
    dst[0] = rgb[0]; dst[1] = rgb[1]; dst[2] = rgb[2];
}

constexpr int kMaxPerLane = 3;   // classes per lane: C <= 192 (the LDS tile caps C below that)

// MODE 0: stage 1 only.  MODE 1: avg family.  MODE 2: log-odds.
template <int MODE>
__global__ __launch_bounds__(kTile) void k_stream(View v, const float *__restrict__ in, int from_logits,
                                                  float inv_t, int32_t *__restrict__ sem_out,
                                                  float *__restrict__ unc_out, float *__restrict__ maxp_out,
                                                  float *__restrict__ probs_out,
                                                  const int32_t *__restrict__ gt) {
    extern __shared__ float tile[];   // [C][kStride]
    __shared__ int s_wave_cnt[4];
    __shared__ int s_win_t[kTile];
    __shared__ int s_win_cell[kTile];

    const int e = blockIdx.y;
    const int t = threadIdx.x;
    const int pix = blockIdx.x * kTile + t;
    const bool active = pix < v.N;
    const int C = v.C;
    const int64_t fbase = (int64_t)e * v.N;
    const float *src = in + ((int64_t)e * C) * v.N + pix;
    float *col = tile + t;

    if (active) {
        for (int c = 0; c < C; ++c) col[c * kStride] = src[(int64_t)c * v.N];
    }
    if (from_logits && active) {
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float x = col[c * kStride] * inv_t;
            col[c * kStride] = x;
            m = fmaxf(m, x);
        }
        float s = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float ev = __expf(col[c * kStride] - m);
            col[c * kStride] = ev;
            s += ev;
        }
        float hsum = 0.0f, best = -1.0f;
        int bi = 0;
        const float inv_s = 1.0f / s;
        for (int c = 0; c < C; ++c) {
            const float p = col[c * kStride] * inv_s;
            col[c * kStride] = p;
            hsum += p * __logf(p + 1e-6f);
            if (p > best) { best = p; bi = c; }
        }
        const int sid = v.lut[bi];
        float conf = best;
        if (sid >= 3 && sid <= 8) {
            // probability mass of every model class that maps to this goal id (infer.py:109-118)
            const int g = sid - 3;
            const int n = v.pool_count[g];
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) acc += col[v.pool_ids[g * SSMAP_MAX_POOL + k] * kStride];
            if (n > 0) conf = acc;
        }
        if (sem_out) sem_out[fbase + pix] = sid;
        if (unc_out) unc_out[fbase + pix] = -hsum / __logf((float)C);
        if (maxp_out) maxp_out[fbase + pix] = conf;
        if (probs_out) {
            float *dst = probs_out + ((int64_t)e * C) * v.N + pix;
            for (int c = 0; c < C; ++c) dst[(int64_t)c * v.N] = col[c * kStride];
        }
    }
    if (MODE == 0) return;

    // ---- hand the winner pixels of this tile to the waves
    const int tgt = active ? v.target[fbase + pix] : -1;
    const bool win = tgt >= 0;
    const unsigned long long ball = __ballot(win);
    const int lane = t & 63, wave = t >> 6;
    if (lane == 0) s_wave_cnt[wave] = __popcll(ball);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) before += s_wave_cnt[w];
        total += s_wave_cnt[w];
    }
    if (win) {
        const int rank = before + __popcll(ball & ((1ull << lane) - 1ull));
        s_win_t[rank] = t;
        s_win_cell[rank] = tgt;
    }
    __syncthreads();

    const ssmap_env_maps &mp = v.envs[e].maps;
    const PoseBlock &pb = v.poses[e];
    const float log_c_np = logf((float)C);

    for (int w = wave; w < total; w += 4) {
        const int tw = s_win_t[w];
        const int cell = s_win_cell[w];
        const int wpix = blockIdx.x * kTile + tw;
        float *mrow = mp.prob_map + (int64_t)cell * C;

        // height map and ground-truth map (mapper.py:338-341, 361-363); every lane computes,
        // lane 0 stores
        const double hw = unorder_f64(v.horder[fbase + wpix]);
        const float hnew = fmaxf(mp.height_map[cell], (float)(hw - pb.agent[1]));

        float p[kMaxPerLane], mv[kMaxPerLane];
#pragma unroll
        for (int j = 0; j < kMaxPerLane; ++j) {
            const int c = lane + 64 * j;
            p[j] = c < C ? tile[c * kStride + tw] : 0.0f;
            mv[j] = c < C ? mrow[c] : 0.0f;
        }
        float best = -INFINITY;
        int bi = 0x7FFFFFFF;

        if (MODE == 1) {
            float wgt = 1.0f;
            if (v.rule != SSMAP_RULE_AVG_PROB) {
                float sp = 0.0f;
#pragma unroll
                for (int j = 0; j < kMaxPerLane; ++j) sp += p[j];
                sp = wave_sum(sp);
                float hs = 0.0f;
#pragma unroll
                for (int j = 0; j < kMaxPerLane; ++j) hs += (lane + 64 * j < C) ? entr(p[j] / sp) : 0.0f;
                const float ent = wave_sum(hs) / log_c_np;
                wgt = v.rule == SSMAP_RULE_W_AVG_PROB ? 1.0f - ent : 1.0f / ent;
            }
            const float wn = mp.weight_map[cell] + wgt;
            const float rcp = 1.0f / wn;
            float sm = 0.0f;
#pragma unroll
            for (int j = 0; j < kMaxPerLane; ++j) {
                const int c = lane + 64 * j;
                if (c < C) {
                    const float d = rcp * (p[j] - mv[j]);
                    mv[j] = mv[j] + d * wgt;
                    mrow[c] = mv[j];
                    sm += mv[j];
                }
            }
            sm = wave_sum(sm);
            float hs = 0.0f;
#pragma unroll
            for (int j = 0; j < kMaxPerLane; ++j) {
                const int c = lane + 64 * j;
                if (c < C) {
                    hs += entr(mv[j] / sm);
                    if (arg_better(mv[j], c, best, bi)) { best = mv[j]; bi = c; }
                }
            }
            const float ent = wave_sum(hs) / log_c_np;
            if (lane == 0) {
                mp.weight_map[cell] = wn;
                mp.unc_map[cell] = ent;
            }
        } else {
            const float p0 = tile[tw];
#pragma unroll
            for (int j = 0; j < kMaxPerLane; ++j) {
                const int c = lane + 64 * j;
                if (c < C) {
                    mv[j] = mv[j] + __logf(p[j] / p0);
                    mrow[c] = mv[j];
                    if (arg_better(mv[j], c, best, bi)) { best = mv[j]; bi = c; }
                }
            }
        }
        wave_argmax(best, bi);
        if (lane == 0) {
            mp.height_map[cell] = hnew;
            if (gt) mp.gt_map[cell] = (uint32_t)gt[fbase + wpix];
            const int id = occupancy_override(v.lut[bi], hnew, v.use_height, v.n_schema);
            commit_cell(mp, v.palette, cell, id);
        }
    }
}

void launch_stream(const View &v, const ssmap_frames &f, bool fuse, float *probs_out, hipStream_t s) {
    const float *in = f.logits ? f.logits : f.probs;
    const int from_logits = f.logits ? 1 : 0;
    const float inv_t = 1.0f / f.temperature;
    dim3 grid((v.N + kTile - 1) / kTile, v.E);
    const size_t lds = (size_t)v.C * kStride * sizeof(float);
    int mode = 0;
    if (fuse) mode = v.rule == SSMAP_RULE_LOG_ODDS ? 2 : 1;
    auto go = [&](auto kern) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(kTile), lds, s, v, in, from_logits, inv_t, f.semantic,
                           f.semantic_unc, f.semantic_max_prob, probs_out, f.semantic_gt);
    };
    if (mode == 0) go(k_stream<0>);
    else if (mode == 1) go(k_stream<1>);
    else go(k_stream<2>);
}

}  // namespace ssmap
