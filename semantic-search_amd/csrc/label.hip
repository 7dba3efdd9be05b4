// Stage 4 for the label rules and stage 5 (found decision).  One thread per update cell,
// read from the unordered (pixel, cell) list the winner kernel wrote.
//   direct / argmax (+filter)   mapper.py:343-359, 412-447
//   hv / prob_hv                mapper.py:604-727
//   goal_decay (+filter)        mapper.py:364-411 with _erode_goal_2d, mapper.py:1114-1161
//   found decision              mapper.py:735-821
// Compiled with -ffp-contract=off (the view-distance gate is float64 and must round like
// numpy's sqrt(dx*dx + dz*dz)).
#include "ssmap_internal.h"
#include "rows.h"

namespace ssmap {

constexpr int kListBlocks = 128;     // blocks per env for list / bbox sweeps (one cell per thread for typical frames)
constexpr int kGdLost = 0x80;        // goal_decay local map: pixel lost its wrapped cell to a later one

__device__ inline int occupancy_override(int id, float h, int use_height, int n_schema) {
    if (!use_height || n_schema != 9) return id;
    if (id == 0 || id == 1 || id == 2) {
        if (h > 0.1f) return 2;
        if (h != -INFINITY && h < 0.1f) return 1;
    }
    return id;
}

__device__ inline void paint(const GMaps &mp, const uint8_t *palette, int cell, int id) {
    const uint8_t *rgb = palette + 3 * id;
    const uint8_t r = rgb[0], g = rgb[1], b = rgb[2];   // all three before the first store (which could alias them)
    gptr<uint8_t> dst = mp.semantic_map_color + 3ll * cell;
    dst[0] = r; dst[1] = g; dst[2] = b;
}

__device__ inline float raise_height(const View &v, const GMaps &mp, const PoseBlock &pb,
                                     int64_t fbase, unsigned slot, int cell) {
    const double h = unorder_f64(v.wheight[fbase + slot]);
    const float hn = fmaxf(mp.height_map[cell], (float)(h - pb.agent[1]));
    mp.height_map[cell] = hn;
    return hn;
}

// numpy's maximum / max: a NaN on either side wins
__device__ inline float nan_max(float a, float b) {
    return a != a ? a : (b != b ? b : fmaxf(a, b));
}

// The goal statistics of one update cell (mapper.py:449-469) and, for the stubborn rule, the
// naive-Bayes verdict on them (mapper.py:502-515): returns false when the model says "not the goal".
// `pr` points at the pixel's probability in class plane 0; planes are N floats apart.
__device__ inline bool goal_statistics(const View &v, const GMaps &mp, const float *pr, int cell, int goal) {
    const int g = goal - 3;
    const int n_pool = v.pool_count[g];
    const int32_t *pool = v.pool_ids + g * SSMAP_MAX_POOL;
    // sum of the goal classes' probabilities in list order, float32 (numpy reduces the gathered
    // columns one after the other)
    float pv[SSMAP_MAX_POOL];
#pragma unroll
    for (int k = 0; k < SSMAP_MAX_POOL; ++k) pv[k] = k < n_pool ? pr[(int64_t)pool[k] * v.N] : 0.0f;
    float mass = 0.0f;
#pragma unroll
    for (int k = 0; k < SSMAP_MAX_POOL; ++k)
        if (k < n_pool) mass = mass + pv[k];
    // largest probability among the other classes
    unsigned long long in_pool[4] = {0ull, 0ull, 0ull, 0ull};
    for (int k = 0; k < n_pool; ++k) in_pool[pool[k] >> 6] |= 1ull << (pool[k] & 63);
    float other = 0.0f;
    bool first = true;
    for (int c = 0; c < v.C; ++c) {
        if ((in_pool[c >> 6] >> (c & 63)) & 1ull) continue;
        const float p = pr[(int64_t)c * v.N];
        other = first ? p : nan_max(other, p);
        first = false;
    }
    const uint32_t views = mp.total_view[cell] + 1u;
    const float cum = mp.cumulative_conf[cell] + mass;
    const float mx = nan_max(mp.max_conf[cell], mass);
    const float mo = nan_max(mp.max_other_conf[cell], other);
    mp.total_view[cell] = views;
    mp.cumulative_conf[cell] = cum;
    mp.max_conf[cell] = mx;
    mp.max_other_conf[cell] = mo;
    // while the training rows are being collected the model is not consulted (mapper.py:474-515)
    if (v.rule != SSMAP_RULE_STUBBORN || v.prm.nb_stats) return true;
    const NbModel &m = v.nb[g];
    const double x[4] = {(double)views, (double)cum, (double)mx, (double)mo};
    double best = 0.0;
    int label = 1;
    for (int k = 0; k < m.n_classes; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double d = x[j] - m.theta[k][j];
            s = s + (d * d) / m.var[k][j];
        }
        const double jll = m.log_prior[k] + (m.konst[k] - 0.5 * s);
        // numpy argmax: the first largest, a NaN counting as largest
        if (k == 0 || jll > best || (jll != jll && best == best)) { best = jll; label = m.classes[k]; }
    }
    return label != 0;
}

__global__ __launch_bounds__(256) void k_label(View v, const int32_t *__restrict__ semantic,
                                               const float *__restrict__ unc,
                                               const float *__restrict__ maxp,
                                               const int32_t *__restrict__ gt) {
    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    const FrameHeader *hdr = v.hdr + e;
    const unsigned n = hdr->n_winners;
    const GMaps mp(v.envs[e].maps);   // by value: the fields stay in registers across the stores below
    const PoseBlock pb = v.poses[e];
    const int goal = pb.goal_id;
    const int2 *list = reinterpret_cast<const int2 *>(v.winners) + fbase;
    const bool hv = v.rule == SSMAP_RULE_HV || v.rule == SSMAP_RULE_PROB_HV;

    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 pc = list[i];
        const uint32_t pw = (uint32_t)pc.x;
        if (hv && i == 0) v.overlay_count[v.cur_overlay][e] = n;
        if (pw & kLostBit) {
            if (hv) v.overlay[v.cur_overlay][fbase + i] = -1;
            continue;
        }
        const int pix = (int)(pw & kPixMask);
        const int cell = pc.y;
        // every gather of this cell is issued before the first store (the stores could alias the
        // loads as far as the compiler knows, and would otherwise chain the round trips)
        const float h_old = mp.height_map[cell];
        const unsigned long long hk = v.wheight[fbase + i];   // height image of this list slot's pixel
        int id = semantic[fbase + pix];
        const int raw = id;
        const int gtv = v.rule != SSMAP_RULE_DIRECT ? gt[fbase + pix] : 0;
        const float unc_v = unc ? unc[fbase + pix] : 0.0f;
        const float maxp_v = maxp ? maxp[fbase + pix] : 0.0f;
        float views = 0.0f, hits = 0.0f;
        int bg = 0;
        if (hv) {
            views = mp.views[cell];
            hits = mp.hits[cell];
            bg = (int)mp.background_map[cell];
        }
        const float views_old = views, hits_old = hits;
        const int bg_old = bg;
        // height map (mapper.py:338-341)
        const float h = fmaxf(h_old, (float)(unorder_f64(hk) - pb.agent[1]));
        mp.height_map[cell] = h;

        if (v.rule == SSMAP_RULE_DIRECT) {
            id = occupancy_override(id, h, v.use_height, v.n_schema);
            mp.semantic_map[cell] = (uint32_t)id;
            paint(mp, v.palette, cell, id);
            continue;
        }
        mp.gt_map[cell] = (uint32_t)gtv;

        if (v.rule == SSMAP_RULE_ARGMAX || v.rule == SSMAP_RULE_STUBBORN) {
            if (unc) mp.unc_map[cell] = unc_v;
            if (v.prm.use_filter && id == goal && (1.0f - unc_v) < v.prm.filter_goal_prob) id = 0;
            id = occupancy_override(id, h, v.use_height, v.n_schema);
            if (v.nb_probs) {   // stubborn, or argmax collecting the statistics
                const bool keep = goal_statistics(v, mp, v.nb_probs + (int64_t)e * v.C * v.N + pix, cell, goal);
                if (!keep && id == goal) id = 2;
            }
            mp.semantic_map[cell] = (uint32_t)id;
            paint(mp, v.palette, cell, id);
            continue;
        }
        if (hv) {
            const bool weighted = v.rule == SSMAP_RULE_PROB_HV;
            if (weighted) mp.unc_map[cell] = unc_v;
            // Up to four update pixels can share a cell through numpy's negative-index wrap, one
            // per sign quadrant (resolve_wrapped in the winner kernel recorded them; this thread has the highest
            // pixel, the others were dropped from the list).  numpy applies every statement of
            // the rule to all of them with "the last one wins" among the entries that statement
            // selects, so the pixels are walked in ascending order and each statement keeps the
            // last value it saw (mapper.py:630-666, 689-727).
            int others[3], other_quad[3], n_others = 0;
            if (hdr->n_neg) {
                const int4 qs = *reinterpret_cast<const int4 *>(v.quad + e * v.quad_stride + 4ll * cell);
                const int q4[4] = {qs.x, qs.y, qs.z, qs.w};
                for (int q = 0; q < 4; ++q)
                    if (q4[q] != 0 && q4[q] != pix + 1) {
                        int at = n_others++;
                        while (at > 0 && others[at - 1] > q4[q] - 1) {   // ascending pixel order
                            others[at] = others[at - 1]; other_quad[at] = other_quad[at - 1]; --at;
                        }
                        others[at] = q4[q] - 1; other_quad[at] = q;
                    }
            }
            bool any_gated = false;
            for (int k = 0; k <= n_others; ++k) {
                const bool self = k == n_others;                      // this thread's pixel comes last
                const int p = self ? pix : others[k];
                const int raw_k = self ? raw : semantic[fbase + p];
                const float maxp_k = self ? maxp_v : (maxp ? maxp[fbase + p] : 0.0f);
                const bool row_neg = self ? (pw & kRowNegBit) != 0 : (other_quad[k] & 2) != 0;
                const bool col_neg = self ? (pw & kColNegBit) != 0 : (other_quad[k] & 1) != 0;
                int id_k = raw_k;
                if (!weighted && v.prm.use_filter && id_k == goal && maxp_k < v.prm.filter_goal_prob) id_k = 0;
                id_k = occupancy_override(id_k, h, v.use_height, v.n_schema);
                if (id_k != goal) bg = id_k;
                bool gated = true;
                if (v.prm.hv_limit_obj_view_dist > 0.0) {
                    // map2world of the UNWRAPPED cell (mapper.py:639-644, 865-874)
                    int row = cell / mp.map_w, col = cell - row * mp.map_w;
                    if (row_neg) row -= mp.map_h;
                    if (col_neg) col -= mp.map_w;
                    const double dx = ((double)col * v.p2m + mp.lower[0]) - pb.agent[0];
                    const double dz = ((double)row * v.p2m + mp.lower[2]) - pb.agent[2];
                    gated = sqrt(dx * dx + dz * dz) < v.prm.hv_limit_obj_view_dist;
                }
                const float add = weighted ? maxp_k : 1.0f;
                if (!weighted || gated) views = views_old + add;      // buffered +=: from the old value
                if (gated && raw_k == goal) hits = hits_old + add;
                any_gated |= gated;
            }
            if (bg != bg_old) mp.background_map[cell] = (uint32_t)bg;
            if (views != views_old) mp.views[cell] = views;
            if (hits != hits_old) mp.hits[cell] = hits;
            const bool show = any_gated && hits > 0.0f && views != INFINITY;
            const int shown = show ? goal : bg;
            mp.semantic_map[cell] = (uint32_t)shown;
            paint(mp, v.palette, cell, shown);
            // slot i of the overlay list: no counter, so no same-address atomics
            v.overlay[v.cur_overlay][fbase + i] = show ? cell : -1;
            continue;
        }
    }
}

// ---------------------------------------------------------------- goal_decay
__device__ inline bool gd_box(const FrameHeader *hdr, int &r0, int &c0, int &bh, int &bw) {
    if (hdr->wrmin > hdr->wrmax) return false;
    r0 = hdr->wrmin; c0 = hdr->wcmin;
    bh = hdr->wrmax - r0 + 1; bw = hdr->wcmax - c0 + 1;
    return true;
}

__global__ void k_gd_clear(View v) {
    const int e = blockIdx.y;
    int r0, c0, bh, bw;
    if (!gd_box(v.hdr + e, r0, c0, bh, bw)) return;
    uint8_t *local = v.local_ids + (int64_t)e * v.local_stride;
    const int64_t n = (int64_t)bh * bw;
    for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) local[i] = 0;
}

__global__ void k_gd_scatter(View v, const int32_t *__restrict__ semantic, const float *__restrict__ unc,
                             const int32_t *__restrict__ gt) {
    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    FrameHeader *hdr = v.hdr + e;
    int r0, c0, bh, bw;
    if (!gd_box(hdr, r0, c0, bh, bw)) return;
    const unsigned n = hdr->n_winners;
    const GMaps mp(v.envs[e].maps);   // by value: the fields stay in registers across the stores below
    const PoseBlock pb = v.poses[e];
    const int goal = pb.goal_id;
    uint8_t *local = v.local_ids + (int64_t)e * v.local_stride;
    const int2 *list = reinterpret_cast<const int2 *>(v.winners) + fbase;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 pc = list[i];
        const uint32_t pw = (uint32_t)pc.x;
        const bool lost = (pw & kLostBit) != 0;   // a later pixel writes the same wrapped cell
        const int pix = (int)(pw & kPixMask);
        const int cell = pc.y;
        // every gather before the first store (which could alias them as far as the compiler knows)
        int id = semantic[fbase + pix];
        const int gtv = gt[fbase + pix];
        const float h_old = mp.height_map[cell];
        const unsigned long long hk = v.wheight[fbase + i];
        if (!lost) {
            mp.height_map[cell] = fmaxf(h_old, (float)(unorder_f64(hk) - pb.agent[1]));
            mp.gt_map[cell] = (uint32_t)gtv;
        }
        // the uncertainty of the few goal pixels only: one more gather per pixel would cost more
        // than the second round trip of these
        if (v.prm.use_filter && id == goal && (1.0f - unc[fbase + pix]) < v.prm.filter_goal_prob) id = 0;
        int row = cell / mp.map_w, col = cell - row * mp.map_w;
        if (pw & kRowNegBit) row -= mp.map_h;
        if (pw & kColNegBit) col -= mp.map_w;
        // the local map is indexed by UNWRAPPED cells, so wrapped twins stay distinct here
        local[(int64_t)(row - r0) * bw + (col - c0)] = (uint8_t)((id + 1) | (lost ? kGdLost : 0));
        if (id == goal) hdr->any_goal = 1u;
    }
}

// erosion of the goal mask by an n x n box anchored at n/2 with the outside counted as set
__global__ void k_gd_erode(View v, int size) {
    const int e = blockIdx.y;
    FrameHeader *hdr = v.hdr + e;
    int r0, c0, bh, bw;
    if (!gd_box(hdr, r0, c0, bh, bw) || !hdr->any_goal) return;
    const uint8_t goal1 = (uint8_t)(v.poses[e].goal_id + 1);
    const uint8_t *local = v.local_ids + (int64_t)e * v.local_stride;
    uint8_t *er = v.local_eroded + (int64_t)e * v.local_stride;
    const int a = size / 2;
    const int64_t n = (int64_t)bh * bw;
    for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / bw), c = (int)(i - (int64_t)r * bw);
        bool keep = true;
        for (int dr = -a; dr < size - a && keep; ++dr) {
            const int rr = r + dr;
            if (rr < 0 || rr >= bh) continue;
            for (int dc = -a; dc < size - a; ++dc) {
                const int cc = c + dc;
                if (cc < 0 || cc >= bw) continue;
                if ((local[(int64_t)rr * bw + cc] & (kGdLost - 1)) != goal1) { keep = false; break; }
            }
        }
        er[i] = keep ? 1 : 0;
        if (keep) hdr->any_eroded = 1u;
    }
}

// Is local cell (r, c) part of the rewritten update set (mapper.py:1157), and with which id?
__device__ inline bool gd_member(const uint8_t *local, const uint8_t *er, int bh, int bw, int r, int c,
                                 int lo, int hi, int goal, int &id) {
    const int stored = local[(int64_t)r * bw + c] & (kGdLost - 1);
    id = stored ? stored - 1 : 0;
    if (id != goal && id != 0) return true;
    if (!er) return false;   // nothing survived the erosion: the dilated mask is empty
    for (int dr = -lo; dr <= hi; ++dr) {
        const int rr = r + dr;
        if (rr < 0 || rr >= bh) continue;
        for (int dc = -lo; dc <= hi; ++dc) {
            const int cc = c + dc;
            if (cc < 0 || cc >= bw) continue;
            if (er[(int64_t)rr * bw + cc]) return true;
        }
    }
    return false;
}

__global__ void k_gd_apply(View v, int dil) {
    const int e = blockIdx.y;
    const FrameHeader *hdr = v.hdr + e;
    int r0, c0, bh, bw;
    if (!gd_box(hdr, r0, c0, bh, bw)) return;
    const bool any_goal = hdr->any_goal != 0;
    const GMaps mp(v.envs[e].maps);   // by value: the fields stay in registers across the stores below
    const int goal = v.poses[e].goal_id;
    const uint8_t *local = v.local_ids + (int64_t)e * v.local_stride;
    const uint8_t *er = hdr->any_eroded ? v.local_eroded + (int64_t)e * v.local_stride : nullptr;
    const int lo = (dil - 1) / 2, hi = (dil - 1) - lo;
    const int Hm = mp.map_h, Wm = mp.map_w;
    const bool may_wrap = r0 < 0 || c0 < 0;
    const int64_t n = (int64_t)bh * bw;
    for (int64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / bw), c = (int)(i - (int64_t)r * bw);
        int id;
        bool has_goal, has_other;
        if (any_goal) {
            if (!gd_member(local, er, bh, bw, r, c, lo, hi, goal, id)) continue;
            has_goal = id == goal;
            has_other = !has_goal;
            if (may_wrap) {
                // Twins: members whose unwrapped index differs by the map size land on the same
                // map cell.  numpy applies "+= 1" once if any twin is a goal cell, then
                // "*= decay" once if any twin is not, and the semantic id of the LAST twin in
                // row-major order of the local map (mapper.py:391-396, 730).
                const int R = r + r0, Cc = c + c0;
                bool last = true;
                for (int k = 1; k < 4; ++k) {
                    const int tr = (k & 2) ? (R < 0 ? R + Hm : R - Hm) : R;
                    const int tc = (k & 1) ? (Cc < 0 ? Cc + Wm : Cc - Wm) : Cc;
                    const int lr = tr - r0, lc = tc - c0;
                    if (lr < 0 || lr >= bh || lc < 0 || lc >= bw) continue;
                    int tid;
                    if (!gd_member(local, er, bh, bw, lr, lc, lo, hi, goal, tid)) continue;
                    if (tid == goal) has_goal = true; else has_other = true;
                    if (tr > R || (tr == R && tc > Cc)) last = false;
                }
                if (!last) continue;
            }
        } else {
            const int stored = local[i];
            if (stored == 0 || (stored & kGdLost)) continue;   // only the last pixel of a wrapped cell writes
            id = stored - 1;
            has_goal = false;
            has_other = true;
        }
        int row = r + r0, col = c + c0;
        if (row < 0) row += Hm;
        if (col < 0) col += Wm;
        const int cell = row * Wm + col;
        float g = mp.goal_map[cell];
        if (has_goal) g = g + 1.0f;
        if (has_other) g = g * v.prm.goal_decay;
        mp.goal_map[cell] = g;
        if (id == goal) id = 0;
        if (g > v.prm.goal_mark) id = goal;
        id = occupancy_override(id, mp.height_map[cell], v.use_height, v.n_schema);
        mp.semantic_map[cell] = (uint32_t)id;
        paint(mp, v.palette, cell, id);
    }
}

// ---------------------------------------------------------------- stage 5
// The filled disk of radius r around the robot cell (per-row half widths = the raster of
// OpenCV's filled circle, mapper.py:737-743) is swept 64 cells at a time by each wave, one
// cell per lane for the cheap tests.  For the probability rules the surviving goal cells of
// the block are collected in LDS and their entropies are then computed sixteen at a time,
// one cell per 16-lane row with the lanes over the classes.
constexpr int kFoundBlocks = 24;
constexpr int kFoundThreads = 256;

__device__ inline float entr_f(float x) {
    return x > 0.0f ? -x * logf(x) : (x == 0.0f ? 0.0f : (x != x ? x : -INFINITY));
}

__global__ __launch_bounds__(kFoundThreads) void k_found(View v) {
    __shared__ int s_cand[kFoundThreads];
    __shared__ unsigned int s_n;
    const int e = blockIdx.y;
    const GMaps mp(v.envs[e].maps);   // by value: the fields stay in registers across the stores below
    const PoseBlock pb = v.poses[e];
    if (pb.skip || (v.hdr[e].status & SSMAP_ENV_INDEX_ERROR)) return;   // no frame for this env, or one that raised: no decision, no rejection side effects
    const int goal = pb.goal_id;
    const int R = v.disk_radius, side = 2 * R + 1;
    const int C = v.C;
    const bool hv = v.rule == SSMAP_RULE_HV || v.rule == SSMAP_RULE_PROB_HV;
    const bool avg = v.rule >= SSMAP_RULE_AVG_PROB && v.rule <= SSMAP_RULE_W2_AVG_PROB;
    const bool by_entropy = avg || v.rule == SSMAP_RULE_LOG_ODDS;
    const float log_c = C > 0 ? logf((float)C) : 1.0f;
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int row_id = threadIdx.x >> 4;                 // 16 rows of 16 lanes per block
    const int n_cells = side * side;
    bool found = false;
    // every pass takes blockDim cells per block; the trip count is the same for all threads
    for (int base = blockIdx.x * kFoundThreads; base < n_cells; base += gridDim.x * kFoundThreads) {
        const int i = base + threadIdx.x;
        int cell = -1;
        if (i < n_cells) {
            const int dr = i / side - R, dc = i % side - R;
            const int half = v.disk_hw[dr < 0 ? -dr : dr];
            const int row = pb.robot_row + dr, col = pb.robot_col + dc;
            if (half >= 0 && dc >= -half && dc <= half && row >= 0 && row < mp.map_h && col >= 0 && col < mp.map_w)
                cell = row * mp.map_w + col;
        }
        if (hv) {
            if (cell < 0) continue;
            const float views = mp.views[cell], hits = mp.hits[cell];
            if (!(views > v.prm.hv_min_views && hits > 0.0f) || views == INFINITY) continue;
            if (hits / views > v.prm.hv_ratio) {
                found = true;
            } else {   // rejected for good (mapper.py:763-772)
                mp.views[cell] = INFINITY;
                const int bg = (int)mp.background_map[cell];
                mp.semantic_map[cell] = (uint32_t)bg;
                paint(mp, v.palette, cell, bg);
            }
            continue;
        }
        const bool cand = cell >= 0 && (int)mp.semantic_map[cell] == goal;
        if (!by_entropy) {
            found |= cand;
            continue;
        }
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const unsigned long long ball = __ballot(cand);
        unsigned wbase = 0;
        if (lane == 0 && ball) wbase = atomicAdd(&s_n, (unsigned)__popcll(ball));
        wbase = __shfl(wbase, 0);
        if (cand) s_cand[wbase + (unsigned)__popcll(ball & ((1ull << lane) - 1ull))] = cell;
        __syncthreads();
        const unsigned n_cand = s_n;
        for (unsigned k0 = 0; k0 < n_cand; k0 += 16) {     // same trip count for the whole block
            const unsigned k = k0 + row_id;
            const bool on = k < n_cand;
            const gptr<float> m = mp.prob_map + (int64_t)(on ? s_cand[k] : 0) * C;
            // the cell's values of this lane (classes sub, sub + 16, ...) once, all loads in flight
            // together; the probability rules take at most 155 classes
            constexpr int kPer = 10;
            float mv[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int c = sub + 16 * q;
                mv[q] = m[c < C ? c : 0];
            }
            float ent;
            if (avg) {
                float s = 0.0f;
#pragma unroll
                for (int q = 0; q < kPer; ++q) s += (on && sub + 16 * q < C) ? mv[q] : 0.0f;
                s = row_sum(s);
                float hs = 0.0f;
#pragma unroll
                for (int q = 0; q < kPer; ++q) hs += (on && sub + 16 * q < C) ? entr_f(mv[q] / s) : 0.0f;
                ent = row_sum(hs) / log_c;
            } else {   // softmax of the log-odds first (mapper.py:805-812)
                float mx = -INFINITY;
#pragma unroll
                for (int q = 0; q < kPer; ++q)
                    if (sub + 16 * q < C) mx = fmaxf(mx, on ? mv[q] : 0.0f);
                mx = row_max(mx);
                float s = 0.0f;
#pragma unroll
                for (int q = 0; q < kPer; ++q) s += (on && sub + 16 * q < C) ? expf(mv[q] - mx) : 0.0f;
                s = row_sum(s);
                float s2 = 0.0f;
#pragma unroll
                for (int q = 0; q < kPer; ++q) s2 += (on && sub + 16 * q < C) ? expf(mv[q] - mx) / s : 0.0f;
                s2 = row_sum(s2);
                float hs = 0.0f;
#pragma unroll
                for (int q = 0; q < kPer; ++q)
                    hs += (on && sub + 16 * q < C) ? entr_f((expf(mv[q] - mx) / s) / s2) : 0.0f;
                ent = row_sum(hs) / log_c;
            }
            if (on && ent < v.prm.entropy_threshold) found = true;
        }
        __syncthreads();
    }
    if (found) v.hdr[e].found = 1;
    // goal_decay: the local map of the update box goes back to zero for the next step (a launch less
    // than clearing it in front of the scatter)
    if (v.rule == SSMAP_RULE_GOAL_DECAY && v.local_clean) {
        int r0, c0, bh, bw;
        if (gd_box(v.hdr + e, r0, c0, bh, bw)) {
            uint8_t *local = v.local_ids + (int64_t)e * v.local_stride;
            const int64_t n = (int64_t)bh * bw;
            // (four bytes a thread where the box allows; the buffer is 256-byte aligned per env)
            for (int64_t i = 4 * (int64_t)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += 4 * (int64_t)gridDim.x * blockDim.x) {
                if (i + 4 <= n) *reinterpret_cast<uint32_t *>(local + i) = 0u;
                else for (int64_t k = i; k < n; ++k) local[k] = 0;
            }
        }
    }
    // clear the wrapped-cell quadrant slots of this frame
    if (blockIdx.x == 0 && v.hdr[e].n_neg) {
        const unsigned n = v.hdr[e].n_winners;
        const int2 *list = reinterpret_cast<const int2 *>(v.winners) + (int64_t)e * v.N;
        int32_t *quad = v.quad + e * v.quad_stride;
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
            *reinterpret_cast<int4 *>(quad + 4ll * list[i].y) = make_int4(0, 0, 0, 0);
    }
}

void launch_label_rules(const View &v, const ssmap_frames &f, hipStream_t s) {
    dim3 grid(kListBlocks, v.E), block(256);
    if (v.rule == SSMAP_RULE_GOAL_DECAY) {
        const int erode = (int)ceil(0.1 / v.p2m), dil = (int)ceil(0.2 / v.p2m);
        // (the engine's own scratch is zero between steps: the found kernel wipes what a step wrote)
        if (!v.local_clean) hipLaunchKernelGGL(k_gd_clear, grid, block, 0, s, v);
        hipLaunchKernelGGL(k_gd_scatter, grid, block, 0, s, v, f.semantic, f.semantic_unc, f.semantic_gt);
        hipLaunchKernelGGL(k_gd_erode, grid, block, 0, s, v, erode);
        hipLaunchKernelGGL(k_gd_apply, grid, block, 0, s, v, dil);
        return;
    }
    // (hv rules: last frame's goal overlay was undone by the winner kernel)
    hipLaunchKernelGGL(k_label, grid, block, 0, s, v, f.semantic, f.semantic_unc, f.semantic_max_prob, f.semantic_gt);
}

// Path overlay (mapper.py:831-833): every cell of the trail shows the path id.
__global__ void k_path_paint(View v) {
    const int e = blockIdx.y;
    const unsigned n = v.path_count[e];
    const GMaps mp(v.envs[e].maps);   // by value: the fields stay in registers across the stores below
    const int32_t *cells = v.path_cells + (int64_t)e * v.path_cap;
    const int id = v.n_schema + 1;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cell = cells[i];
        mp.semantic_map[cell] = (uint32_t)id;
        paint(mp, v.palette, cell, id);
    }
}

void launch_path_paint(const View &v, uint32_t max_count, hipStream_t s) {
    if (max_count == 0) return;
    const unsigned blocks = (max_count + 255u) / 256u;
    hipLaunchKernelGGL(k_path_paint, dim3(blocks < 64u ? blocks : 64u, v.E), dim3(256), 0, s, v);
}

// Does the ground-truth map hold the goal id inside the found disk around the robot cell
// (what & 1), anywhere (what & 2)?  (env.py:581-604.)  One block per env; out[2 e], out[2 e + 1].
__global__ __launch_bounds__(1024) void k_gt_goal(View v, int what, int32_t *__restrict__ out) {
    const int e = blockIdx.x;
    const GMaps mp(v.envs[e].maps);
    const PoseBlock &pb = v.poses[e];
    const uint32_t goal = (uint32_t)pb.goal_id;
    // without probabilistic mapping the reference's gt_map is its semantic_map (mapper.py:257)
    const gptr<uint32_t> gt = mp.gt_map ? mp.gt_map : mp.semantic_map;
    int in_disk = 0, anywhere = 0;
    if (what & 2) {
        const int n = mp.map_h * mp.map_w;
        for (int i = threadIdx.x; i < n; i += blockDim.x) anywhere |= gt[i] == goal;
    }
    if (what & 1) {
        const int R = v.disk_radius, side = 2 * R + 1;
        for (int i = threadIdx.x; i < side * side; i += blockDim.x) {
            const int dr = i / side - R, dc = i % side - R;
            const int half = v.disk_hw[dr < 0 ? -dr : dr];
            const int row = pb.robot_row + dr, col = pb.robot_col + dc;
            if (half >= 0 && dc >= -half && dc <= half && row >= 0 && row < mp.map_h && col >= 0 && col < mp.map_w)
                in_disk |= gt[row * mp.map_w + col] == goal;
        }
    }
    in_disk = __syncthreads_or(in_disk);
    anywhere = __syncthreads_or(anywhere);
    if (threadIdx.x == 0) { out[2 * e] = in_disk ? 1 : 0; out[2 * e + 1] = anywhere ? 1 : 0; }
}

void launch_gt_goal(const View &v, int what, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_gt_goal, dim3(v.E), dim3(1024), 0, s, v, what, out);
}

void launch_found(const View &v, hipStream_t s) {
    hipLaunchKernelGGL(k_found, dim3(kFoundBlocks, v.E), dim3(kFoundThreads), 0, s, v);
}

}  // namespace ssmap
