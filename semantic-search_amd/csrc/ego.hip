// Egocentric observation maps (mapper.py:876-998): one thread per output pixel gathers straight
// from the global maps -- crop, goal recolour, nearest-neighbour rotation, agent disk, auxiliary
// squares, centre crop and resize are folded into the address computation of each tap, so no
// intermediate image exists and no map leaves the GPU.
//
// The OpenCV steps are restated exactly as published for 4.x:
//   warpAffine / INTER_NEAREST   inverse matrix in float64 (host), source coordinate =
//                                (rint((M1 y + M2) 1024) + 512 + rint(M0 x 1024)) >> 10
//   resize / INTER_AREA shrink   separable area table, float32, horizontal pass first
//   resize / INTER_AREA enlarge  bilinear kernel with area-mode coefficients, 11-bit weights,
//                                8-bit vertical pass
//   resize / INTER_NEAREST       min(floor(d * scale), size - 1)
// Compiled with -ffp-contract=off: the fixed-point coordinates and the float32 sums must round
// like the separate multiply and add of the CPU code.
#include "ssmap_internal.h"

namespace ssmap {

// area-table taps per output pixel and axis: the kernel is built for 8 (shrink factors below 7,
// the usual crops) and for 32 (up to the host's limit of kEgoTapLimit taps)

// What rot_pixel needs besides the tables: kept in LDS / registers by the kernel.
struct EgoView {
    const ssmap_env_maps *mp;
    const EgoPose *pose;        // LDS copy
    const EgoTables *tab;       // global; only its scalars are read per tap
    const int32_t *aux_points;  // LDS copy of tab->aux_points
    const int32_t *agent_hw;    // LDS copy of tab->agent_hw
    const uint8_t *palette;     // LDS copy of the first n_palette palette rows
    int n_palette;
    bool has_aux;
    int side, origin, aux_half, aux_reach, n_aux, agent_radius;
};

// Fixed-point source coordinate of warpAffine: the row terms (x0, y0) and the column terms
// (ad, bd) are separable, so a pixel's taps evaluate them once per tap row / tap column.
__device__ inline void warp_row_terms(const double *m, int Y, int &x0, int &y0) {
    x0 = (int)__builtin_rint((m[1] * (double)Y + m[2]) * 1024.0) + 512;
    y0 = (int)__builtin_rint((m[4] * (double)Y + m[5]) * 1024.0) + 512;
}
__device__ inline void warp_col_terms(const double *m, int X, int &ad, int &bd) {
    ad = (int)__builtin_rint(m[0] * (double)X * 1024.0);
    bd = (int)__builtin_rint(m[3] * (double)X * 1024.0);
}
// Source pixel of the rotated image -> global map cell.  false: outside the crop (border value).
__device__ inline bool warp_cell(const EgoView &ev, int x0, int y0, int ad, int bd, int &cell, bool &in_map) {
    int xs = (x0 + ad) >> 10, ys = (y0 + bd) >> 10;
    xs = xs < -32768 ? -32768 : (xs > 32767 ? 32767 : xs);
    ys = ys < -32768 ? -32768 : (ys > 32767 ? 32767 : ys);
    if (xs < 0 || xs >= ev.side || ys < 0 || ys >= ev.side) return false;   // constant border
    const long long r = (long long)ev.pose->row0 + ys, c = (long long)ev.pose->col0 + xs;
    in_map = r >= 0 && r < ev.mp->map_h && c >= 0 && c < ev.mp->map_w;      // zero padding outside
    cell = in_map ? (int)(r * ev.mp->map_w + c) : 0;
    return true;
}

// Rotated, overlaid image of one env at column X, row Y (both in [0, side)).
__device__ inline void rot_pixel(const EgoView &ev, int X, int Y, int x0, int y0, int ad, int bd, int (&rgb)[3]) {
    const EgoTables &t = *ev.tab;
    const int dy = Y - ev.origin, dx = X - ev.origin;
    const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
    // auxiliary squares are drawn last (mapper.py:934-935): the highest bin that covers the pixel wins
    if (ev.has_aux && ady <= ev.aux_reach && adx <= ev.aux_reach) {
        const int half = ev.aux_half;
        for (int j = ev.n_aux - 1; j >= 0; --j) {
            const int p0 = ev.aux_points[2 * j] + ev.origin, p1 = ev.aux_points[2 * j + 1] + ev.origin;
            if (Y >= p0 - half && Y < p0 + half && X >= p1 - half && X < p1 + half) {
                const uint8_t *c = ev.pose->aux_color + 3 * j;
                rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
                return;
            }
        }
    }
    // agent disk (mapper.py:932, 1041-1051)
    if (ady <= ev.agent_radius && ev.agent_hw[ady] >= 0 && adx <= ev.agent_hw[ady]) {
        rgb[0] = t.arrow_color[0]; rgb[1] = t.arrow_color[1]; rgb[2] = t.arrow_color[2];
        return;
    }
    int cell;
    bool in_map;
    rgb[0] = rgb[1] = rgb[2] = 0;
    if (!warp_cell(ev, x0, y0, ad, bd, cell, in_map) || !in_map) return;
    const int id = (int)ev.mp->semantic_map[cell];
    if (id == ev.pose->goal_id) {                                           // mapper.py:920
        rgb[0] = t.active_goal_color[0]; rgb[1] = t.active_goal_color[1]; rgb[2] = t.active_goal_color[2];
    } else if (id < ev.n_palette) {
        // semantic_map_color[cell] is palette[semantic_map[cell]] after every write of the mapper
        // (mapper.py:728-733, 772, 833): one gather instead of four
        const uint8_t *c = ev.palette + 3 * id;
        rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
    } else {
        const uint8_t *c = ev.mp->semantic_map_color + 3ll * cell;
        rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
    }
}

__device__ inline uint8_t round_u8(float v) {
    const float r = __builtin_rintf(v);
    return (uint8_t)(r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r));
}

// grid: (ceil(out_w * out_h / 256), 2 maps, envs of this launch)
template <int kEgoMaxTaps>
__global__ __launch_bounds__(256) void k_ego_maps(View v, EgoPack pack, int env0, const EgoTables *tables,
                                                  uint8_t *ego_small, uint8_t *ego_large, float *unc_small,
                                                  float *unc_large) {
    const EgoTables &t = *tables;
    const int which = blockIdx.y;                  // 0 = small crop, 1 = large crop
    const int e = env0 + blockIdx.z;
    // this env's pose goes through LDS: pointing into the by-value kernel argument would make every
    // thread copy the whole pack into private memory
    __shared__ EgoPose s_pose;
    __shared__ int32_t s_aux[2 * SSMAP_MAX_AUX];
    __shared__ int32_t s_hw[kEgoMaxRadius + 1];
    __shared__ uint8_t s_pal[3 * 64];
    if (threadIdx.x == 0) s_pose = pack.p[blockIdx.z];
    if (threadIdx.x < 2 * SSMAP_MAX_AUX) s_aux[threadIdx.x] = t.aux_points[threadIdx.x];
    if (threadIdx.x <= kEgoMaxRadius) s_hw[threadIdx.x] = t.agent_hw[threadIdx.x];
    const int n_pal = min(v.palette_rows, 64);
    if (threadIdx.x < 3 * n_pal) s_pal[threadIdx.x] = v.palette[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.out_w * t.out_h) return;
    const int dy = i / t.out_w, dx = i - dy * t.out_w;
    const EgoAxis &ax = t.axis[which][0], &ay = t.axis[which][1];
    EgoView ev;
    const ssmap_env_maps maps = v.envs[e].maps;   // by value: no pointer fetch in front of every gather
    ev.mp = &maps;
    ev.pose = &s_pose;
    ev.tab = tables;
    ev.aux_points = s_aux;
    ev.agent_hw = s_hw;
    ev.palette = s_pal;
    ev.n_palette = n_pal;
    ev.has_aux = s_pose.has_aux != 0;
    ev.side = t.side;
    ev.origin = t.origin;
    ev.aux_half = t.aux_half;
    ev.aux_reach = t.aux_reach;
    ev.n_aux = t.n_aux;
    ev.agent_radius = t.agent_radius;
    const double *m = s_pose.m;
    const int off = t.crop_off[which];             // centre crop of the rotated image (mapper.py:937-946)
    uint8_t *dst = (which ? ego_large : ego_small) + ((int64_t)e * t.out_h * t.out_w + i) * 3;

    if (t.mode[which] != kEgoLinear) {
        // column taps of this output pixel: source column, weight, fixed-point column terms
        const int nx = ax.count[dx], ny = ay.count[dy];
        int sxk[kEgoMaxTaps], adk[kEgoMaxTaps], bdk[kEgoMaxTaps];
        float alk[kEgoMaxTaps];
#pragma unroll
        for (int k = 0; k < kEgoMaxTaps; ++k) {
            const bool on = k < nx;
            sxk[k] = on ? t.tap_src[ax.first[dx] + k] : 0;
            alk[k] = on ? t.tap_w[ax.first[dx] + k] : 0.0f;
            warp_col_terms(m, off + sxk[k], adk[k], bdk[k]);
        }
        float sum[3] = {0.0f, 0.0f, 0.0f};
        for (int j = 0; j < ny; ++j) {
            const int sy = t.tap_src[ay.first[dy] + j];
            const float beta = t.tap_w[ay.first[dy] + j];
            int x0, y0;
            warp_row_terms(m, off + sy, x0, y0);
            float buf[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < kEgoMaxTaps; ++k) {
                if (k < nx) {
                    int rgb[3];
                    rot_pixel(ev, off + sxk[k], off + sy, x0, y0, adk[k], bdk[k], rgb);
#pragma unroll
                    for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)rgb[c] * alk[k];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) sum[c] = sum[c] + beta * buf[c];
        }
        if (t.mode[which] == kEgoBox) {
            // integer factors: the taps have weight one, so `sum` is the block's exact integer sum.
            // 2 x 2 blocks are rounded half up in integers, the others through sum * (1 / area)
            const bool two = t.box_area[which] == 4 && nx == 2;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                dst[c] = two ? (uint8_t)(((int)sum[c] + 2) >> 2) : round_u8(sum[c] * t.box_scale[which]);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c] = round_u8(sum[c]);
        }
    } else {
        // bilinear kernel, area-mode coefficients: two source columns (the second pinned to the
        // last column), two source rows (the second clipped), 11-bit weights
        const int sx0 = ax.lin_src[dx], sy0 = ay.lin_src[dy];
        const int sx1 = min(sx0 + 1, ax.ssize - 1), sy1 = min(sy0 + 1, ay.ssize - 1);
        const int a0 = ax.lin_w0[dx], a1 = ax.lin_w1[dx], b0 = ay.lin_w0[dy], b1 = ay.lin_w1[dy];
        int x00, y00, x01, y01, ad0, bd0, ad1, bd1;
        warp_row_terms(m, off + sy0, x00, y00);
        warp_row_terms(m, off + sy1, x01, y01);
        warp_col_terms(m, off + sx0, ad0, bd0);
        warp_col_terms(m, off + sx1, ad1, bd1);
        int p00[3], p01[3], p10[3], p11[3];
        rot_pixel(ev, off + sx0, off + sy0, x00, y00, ad0, bd0, p00);
        rot_pixel(ev, off + sx1, off + sy0, x00, y00, ad1, bd1, p01);
        rot_pixel(ev, off + sx0, off + sy1, x01, y01, ad0, bd0, p10);
        rot_pixel(ev, off + sx1, off + sy1, x01, y01, ad1, bd1, p11);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int r0 = p00[c] * a0 + p01[c] * a1, r1 = p10[c] * a0 + p11[c] * a1;
            const int o = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            dst[c] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
        }
    }

    float *udst = which ? unc_large : unc_small;
    if (udst && ev.mp->unc_map) {
        // nearest resize of the centre crop of the rotated uncertainty (mapper.py:959-994)
        const int sx = min((int)floor((double)dx * ax.scale), ax.ssize - 1);
        const int sy = min((int)floor((double)dy * ay.scale), ay.ssize - 1);
        int x0, y0, ad, bd, cell;
        warp_row_terms(m, off + sy, x0, y0);
        warp_col_terms(m, off + sx, ad, bd);
        bool in_map;
        float u = -0.5f;                                                    // warp border
        if (warp_cell(ev, x0, y0, ad, bd, cell, in_map)) u = in_map ? ev.mp->unc_map[cell] : 0.0f;
        udst[(int64_t)e * t.out_h * t.out_w + i] = u;
    }
}

void launch_ego_maps(const View &v, const EgoPose *host_poses, const EgoTables *d_tables, const EgoTables &h_tables,
                     uint8_t *ego_small, uint8_t *ego_large, float *unc_small, float *unc_large, hipStream_t s) {
    const int n_pix = h_tables.out_w * h_tables.out_h;
    for (int env0 = 0; env0 < v.E; env0 += kEgoPack) {
        const int n = v.E - env0 < kEgoPack ? v.E - env0 : kEgoPack;
        EgoPack pack;
        for (int i = 0; i < n; ++i) pack.p[i] = host_poses[env0 + i];
        const dim3 grid((n_pix + 255) / 256, 2, n);
        if (h_tables.max_taps <= 8)
            hipLaunchKernelGGL((k_ego_maps<8>), grid, dim3(256), 0, s, v, pack, env0, d_tables,
                               ego_small, ego_large, unc_small, unc_large);
        else
            hipLaunchKernelGGL((k_ego_maps<kEgoTapLimit>), grid, dim3(256), 0, s, v, pack, env0, d_tables,
                               ego_small, ego_large, unc_small, unc_large);
    }
}

}  // namespace ssmap
