// Stage 1 (temperature softmax + entropy + confidence, infer.py:97-118 / 302-327) fused with
// stage 4 for the probability rules (avg_prob / w_avg_prob / w2_avg_prob, mapper.py:517-569;
// log_odds, mapper.py:570-603): ONE pass over the class-major [C][H][W] tensor.
//
// Each thread owns one pixel.  Its C values are loaded with C independent coalesced loads
// (lane <-> pixel, so one wave-load reads 256 contiguous bytes of one class plane) and kept
//   * in registers when C is one of the compiled sizes (kernel k_stream_reg), or
//   * in a private LDS column of stride 257 floats otherwise (kernel k_stream_lds).
// Pixels that own a map cell this step (target[pix] = their slot in the frame's update list, set by
// the winner kernel; -1 for the others) write their probability vector to row `slot` of a compact
// [K][C] buffer (vector stores under the owners' execution mask; staging the rows through LDS
// for fully coalesced writes measured the same); the streaming pass never waits on map memory.
// k_update_prob then gives every update cell to a 16-lane row of a wave, lanes over the classes:
// the read-modify-write of the cell's C contiguous floats in the fused map is coalesced and the
// per-cell entropy / argmax are DPP row reductions.
#include "ssmap_internal.h"
#include "rows.h"
#include <cstring>

namespace ssmap {

#ifndef SSMAP_SCHED_GROUP
#define SSMAP_SCHED_GROUP 4
#endif
#ifndef SSMAP_WPE0
#define SSMAP_WPE0 1
#endif
#ifndef SSMAP_WPE150
#define SSMAP_WPE150 2
#endif
// Stage 1 alone (narrow class counts): at most this many waves per SIMD.  Measured at 40 classes, 16 envs:
// 8 waves 143.0 us, 5 waves 141.9, 4 waves 140.7, 3 waves 141.0 (64 envs: 585.8, 579.2, 573.8, 566.8):
// fewer streams in flight per CU read the class planes a little faster.
#ifndef SSMAP_WPE0MAX
#define SSMAP_WPE0MAX 3
#endif
#ifndef SSMAP_WPEDMAX
#define SSMAP_WPEDMAX 8
#endif
#ifndef SSMAP_WPED
#define SSMAP_WPED 1
#endif
constexpr int kTile = 256;
constexpr int kRegTile = 256;      // block of the register kernel (64 and 1024 threads measured slower)
constexpr int kStride = kTile + 1;
constexpr int kStageRows = 8;      // owner rows a wave of the register kernel stages per pass

// One term of an entropy sum in bits: -x log2(x) with the hardware logarithm (v_log_f32, one ulp),
// 0 at 0, -inf below 0, NaN for NaN (scipy.special.entr).  The logarithm is taken of max(x, smallest
// normal): the same product for a normal x, 126 x (below 2e-36, like its true term) for a denormal one,
// which the instruction does not take.  With the accurate logf and a division per class the update
// of 150-class rows was bound by the vector unit, not by memory; three selects less per term are
// still 10 us of the 150-class step.
__device__ inline float entr_bits(float x) {
    const float t = -x * __builtin_amdgcn_logf(x < 1.17549435e-38f ? 1.17549435e-38f : x);   // (NaN: the product carries it)
    return x < 0.0f ? -INFINITY : t;
}

// Natural logarithm through the hardware base-2 logarithm (v_log_f32, one ulp), about two ulp in
// all; a denormal argument is scaled first (the instruction does not take denormals).  log(0) =
// -inf, log(x < 0) = NaN, NaN and +inf pass through.
__device__ inline float log_hw(float x) {
    const bool tiny = x < 1.17549435e-38f;
    const float l2 = __builtin_amdgcn_logf(tiny ? x * 4294967296.0f : x) - (tiny ? 32.0f : 0.0f);
    return l2 * 0.693147180559945309f;
}

__device__ inline int occupancy_override(int id, float h, int use_height, int n_schema) {
    if (!use_height || n_schema != 9) return id;
    if (id == 0 || id == 1 || id == 2) {
        if (h > 0.1f) return 2;
        if (h != -INFINITY && h < 0.1f) return 1;
    }
    return id;
}

__device__ inline void commit_cell(const GMaps &mp, const uint8_t *palette, int cell, int id) {
    mp.semantic_map[cell] = (uint32_t)id;
    const uint8_t *rgb = palette + 3 * id;
    const uint8_t r = rgb[0], g = rgb[1], b = rgb[2];   // all three before the first store (which could alias them)
    gptr<uint8_t> dst = mp.semantic_map_color + 3ll * cell;
    dst[0] = r; dst[1] = g; dst[2] = b;
}

struct StreamArgs {
    const float *in;
    int from_logits;
    float inv_t;
    int pool_trivial;        // every goal id pools at most one class: confidence = max prob
    int32_t *sem_out;
    float *unc_out, *maxp_out, *probs_out;
    const int32_t *gt;
};

// ---------------------------------------------------------------- one map cell
// Rule update of one map cell by one 16-lane row: lane `sub` owns the classes V*(sub+16k) ..
// +V-1 (k < K); p = this frame's probabilities, mv = the cell's fused row (updated in place and
// written back to mrow).  Rows without a live cell (`on` false) still take part in the
// row-local DPP steps.  Out: argmax class, and for the average family the new weight and the
// normalised entropy of the fused row.
// MODE 1: avg family (mapper.py:528-569), MODE 2: log-odds (mapper.py:574-603).
// THROUGH: the row is written with device-scope (write-through) stores; for the direct update, whose
// scattered row stores run beside the streaming reads (V == 1 only).
template <typename VecF>
__device__ __forceinline__ void store_row(gptr<float> at, const VecF &val, bool through) {
    if constexpr (sizeof(VecF) == 4) {
        if (through) { __hip_atomic_store(at, val[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    }
    *(gptr<VecF>)at = val;
}
template <int MODE, int V, int K, bool THROUGH = false, typename VecF>
__device__ __forceinline__ void update_row(const View &v, int C, float log2_c, bool on, int sub, gptr<float> mrow,
                                           const VecF (&p)[K], VecF (&mv)[K], float w_old,
                                           int &bi, float &wn, float &ent_out) {
    // every product and sum rounds on its own, as numpy's array statements do (mapper.py:540-548):
    // with unit weights the fused average then equals the reference's bit for bit
#pragma clang fp contract(off)
    float best = -INFINITY;
    bi = 0x7FFFFFFF;
    ent_out = 0.0f;
    wn = 0.0f;
    if (MODE == 1) {
        float wgt = 1.0f;
        if (v.rule != SSMAP_RULE_AVG_PROB) {
            float sp = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < V; ++j) sp += p[k][j];
            sp = row_sum(sp);
            const float rsp = 1.0f / sp;     // one division per row: p * rsp is within an ulp of p / sp
            float hs = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < V; ++j) hs += (on && V * (sub + 16 * k) < C) ? entr_bits(p[k][j] * rsp) : 0.0f;
            const float ent = row_sum(hs) / log2_c;
            wgt = v.rule == SSMAP_RULE_W_AVG_PROB ? 1.0f - ent : 1.0f / ent;
        }
        wn = w_old + wgt;
        const float rcp = 1.0f / wn;
        float sm = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = V * (sub + 16 * k);
            if (on && c < C) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float d = rcp * (p[k][j] - mv[k][j]);
                    mv[k][j] = mv[k][j] + d * wgt;
                    sm += mv[k][j];
                }
                if (!(v.exp & 32)) store_row(mrow + c, mv[k], THROUGH);
            }
        }
        sm = row_sum(sm);
        const float rsm = 1.0f / sm;
        float hs = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = V * (sub + 16 * k);
            if (on && c < C) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    hs += entr_bits(mv[k][j] * rsm);
                    { const bool take = arg_better(mv[k][j], c + j, best, bi); best = take ? mv[k][j] : best; bi = take ? c + j : bi; }
                }
            }
        }
        ent_out = row_sum(hs) / log2_c;
    } else {
        // class 0 of the observation, broadcast over the row (the other lanes add zeros); one division
        // per row: p * r0 is within an ulp of p / p0 and has its zeros, infinities and NaNs.  A p0
        // so small that its reciprocal would overflow is scaled by 2^64 first, and p with it (exact).
        const float p0 = row_sum(sub == 0 ? p[0][0] : 0.0f);
        const float sc = fabsf(p0) < 7.9e-31f ? 18446744073709551616.0f : 1.0f;
        const float r0 = 1.0f / (p0 * sc);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = V * (sub + 16 * k);
            if (on && c < C) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float q = (p[k][j] * sc) * r0;
                    mv[k][j] = mv[k][j] + log_hw(q);
                    { const bool take = arg_better(mv[k][j], c + j, best, bi); best = take ? mv[k][j] : best; bi = take ? c + j : bi; }
                }
                if (!(v.exp & 32)) store_row(mrow + c, mv[k], THROUGH);
            }
        }
    }
    row_argmax(best, bi);
}

// Classes that the goal lists of the reference's SegFormer wrapper name (infer.py:278-292: chair 19,
// 30, 75; bed 7; plant 17; toilet 65; tv_monitor 143, 130, 141, 89; sofa 23), as a bit per class.
__host__ __device__ constexpr uint32_t ade_pool_word(int w) {
    constexpr int ids[] = {19, 30, 75, 7, 17, 65, 143, 130, 141, 89, 23};
    uint32_t m = 0;
    for (int id : ids) if ((id >> 5) == w) m |= 1u << (id & 31);
    return m;
}
constexpr uint32_t kAdePoolMask[kClassWords] = {
    ade_pool_word(0), ade_pool_word(1), ade_pool_word(2), ade_pool_word(3), ade_pool_word(4), 0, 0, 0};

// The env's maps as the direct update writes them: the pointers once, in registers (read through
// the struct they would be fetched again after every store, which could alias it).
struct DirectMaps {
    gptr<float> prob, weight, unc;
    gptr<uint32_t> sem;
    gptr<uint8_t> color;
    __device__ __forceinline__ void take(const ssmap_env_maps &mp) {
        prob = as_global(mp.prob_map); weight = as_global(mp.weight_map);
        unc = as_global(mp.unc_map); sem = as_global(mp.semantic_map);
        color = as_global(mp.semantic_map_color);
    }
};

// One 16-lane row finishes its cell: p = the owner pixel's probabilities, mv = the cell's fused
// row (lane `sub` holds the classes sub + 16k), cellw = the owner's word (the cell, and what the
// occupancy override asks of the cell's new height: the winner kernel has raised the height map
// and written the ground-truth map already).  update_row, then label and colour.
template <int UMODE, int CT, int V, int K, typename VecF>
__device__ __forceinline__ void direct_cell(const View &v, const DirectMaps &m, int n_cls, float log2_c, bool on, int lane, int cellw,
                                            float w_old, const VecF (&p)[K], VecF (&mv)[K],
                                            const int (&lut_reg)[(CT + 63) / 64], int pal_reg) {
    const int sub = lane & 15;
    const int cell = cellw & (int)kCellMask;
    const gptr<float> mrow = m.prob + (int64_t)cell * n_cls;
    int bi;
    float wn, ent_out;
    update_row<UMODE, V, K>(v, n_cls, log2_c, on, sub, mrow, p, mv, w_old, bi, wn, ent_out);
    int sid = __builtin_amdgcn_ds_bpermute((bi & 63) << 2, lut_reg[0]);
#pragma unroll
    for (int k = 1; k < (CT + 63) / 64; ++k) {
        const int other = __builtin_amdgcn_ds_bpermute((bi & 63) << 2, lut_reg[k]);
        sid = (bi >> 6) == k ? other : sid;
    }
    // occupancy override (mapper.py:556-569) from the two bits the winner kernel took of the new height
    int id = sid;
    if (v.use_height && v.n_schema == 9 && (unsigned)sid <= 2u)
        id = (cellw & (int)kCellHighBit) ? 2 : ((cellw & (int)kCellLowBit) ? 1 : sid);
    const int rgb = __builtin_amdgcn_ds_bpermute((id & 63) << 2, pal_reg);
    // the cell's scalars leave through two store instructions: lanes 1, 2 and 4 of the row write one
    // 32-bit map each, lanes 5..7 the colour bytes (after the row reductions every lane of the
    // row holds the same values)
    if (on && !(v.exp & 16)) {
        gptr<int> dst = nullptr;
        int val = 0;
        if (sub == 1 && UMODE == 1) { dst = (gptr<int>)(m.weight + cell); val = __builtin_bit_cast(int, wn); }
        if (sub == 2 && UMODE == 1) { dst = (gptr<int>)(m.unc + cell); val = __builtin_bit_cast(int, ent_out); }
        if (sub == 4) { dst = (gptr<int>)(m.sem + cell); val = id; }
        if (dst) *dst = val;
        if (sub >= 5 && sub <= 7) m.color[3ll * cell + (sub - 5)] = (uint8_t)(rgb >> (8 * (sub - 5)));
    }
}

// ---------------------------------------------------------------- direct update
// The streaming kernel updates the map cells its pixels own by itself: no probability rows leave
// the kernel and no update kernel runs.  Per wave:
//   request  as soon as the per-pixel words are there (the class planes are still in flight): the
//            owners' words (cell + the two height bits the winner kernel left) are exchanged through
//            LDS, the block's owners pooled, then every 16-lane row asks for one owner's map row (a
//            class per lane and step) and, for the average family, the cell's weight, for the first
//            kDirectPre owners of the block;
//   update   after the softmax: the owners park their exponentials (and their scale) in the block's
//            LDS rows, every 16-lane row updates its cell exactly as k_update_prob does (update_row)
//            and writes the map row, weight / entropy (average family), label and colour.  Height
//            map and ground-truth map are the winner kernel's.  Owners beyond kDirectPre (rare)
//            follow in further passes that load at that point.
// Two block barriers in the request, one per pass in the update.
constexpr int kDirectPre = 32;   // owners per block (and pass) whose map rows are requested ahead
struct DirectShared {
    int count[kRegTile / 64];
    int cell[kDirectPre];
    float scale[kDirectPre];   // average family: what turns an owner's parked row into probabilities
};
// CT: classes the kernel is compiled for; EXACT: the frame has exactly CT classes, otherwise v.C of them
// (at most CT; the kernel's class registers beyond v.C hold zeros)
template <int CT, bool EXACT = true>
struct DirectRows {
    __device__ __forceinline__ static int classes(const View &v) { return EXACT ? CT : v.C; }
    // one class per lane and step (16 lanes x K steps): 40 classes are 3 elements a lane where
    // float4 units would be 4 (of which 10 lanes in 16 hold data).  The owners of the whole block
    // are pooled: batches of four cells are dealt to the block's waves in turn, so that the rows of
    // a batch are full (a typical block of 256 pixels owns 14 cells: four batches, one per wave,
    // where the waves on their own would run five or six)
#ifndef SSMAP_DIRECT_WIDE_V
#define SSMAP_DIRECT_WIDE_V 2
#endif
    // (wide rows: two classes per lane and step, half the load and store instructions per cell)
    static constexpr int V = (EXACT && CT > 64 && CT % SSMAP_DIRECT_WIDE_V == 0) ? SSMAP_DIRECT_WIDE_V : 1;   // (a row of v.C classes may start on any word)
    static constexpr int K = (CT + 16 * V - 1) / (16 * V), NB = kDirectPre / 4 / (kRegTile / 64);
    static constexpr int SV = CT % 4 == 0 ? 4 : (CT % 2 == 0 ? 2 : 1);   // widest vector an owner parks its row with
    typedef float VecF __attribute__((ext_vector_type(V)));
    typedef float VecS __attribute__((ext_vector_type(SV)));
    VecF mv[NB][K];
    float w_old[NB];      // average family: the cell's weight so far
    int cell[NB];         // the owner's word: cell and height bits (kCellMask, kCellHighBit, kCellLowBit)
    bool on[NB];
    int n_blk, brank;     // owners in the block; this lane's number among them (if it owns a cell)
    bool owner;
    DirectMaps maps;

    // row r of the pass = batch r / 4; batch j goes to wave j % 4 as its (j / 4)-th
    __device__ __forceinline__ int my_row(int t, int b) const {
        return 4 * ((t >> 6) + (kRegTile / 64) * b) + ((t & 63) >> 4);
    }
    // the owners r0 .. r0 + kDirectPre - 1 of the block leave their word in LDS, every 16-lane row
    // picks up the owner it will work on (block barriers inside)
    __device__ __forceinline__ void exchange(DirectShared &sh, const View &v, int t, int cellw, int r0) {
        if (owner && brank >= r0 && brank < r0 + kDirectPre) sh.cell[brank - r0] = cellw;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int r = my_row(t, b);
            on[b] = r0 + r < n_blk;
            cell[b] = (on[b] && !(v.exp & 8)) ? sh.cell[r] : 0;
        }
        // (the slots are written again only by a later pass, behind the barriers of this pass's update)
    }
    // every lane loads, from a safe address where it has nothing to do (cell 0, class 0, first pixel)
    __device__ __forceinline__ void load_rows(const View &v, int t) {
        const int sub = t & 15;
        const int n_cls = classes(v);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const gptr<const float> mrow = maps.prob + (int64_t)(cell[b] & (int)kCellMask) * n_cls;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int c = V * (sub + 16 * k);
                mv[b][k] = *(gptr<const VecF>)(mrow + (c < n_cls ? c : 0));
            }
        }
    }
    // the only per-cell scalar left to read: the average family's weight (every lane of the row
    // asks for the same word)
    __device__ __forceinline__ void load_side(bool avg) {
#pragma unroll
        for (int b = 0; b < NB; ++b) w_old[b] = avg ? maps.weight[cell[b] & (int)kCellMask] : 0.0f;
    }
    __device__ __forceinline__ void request(DirectShared &sh, const View &v, int e, int t, int cellw, bool avg) {
        // the env's map pointers once, in registers (read through the struct they would be fetched
        // again after every store, which could alias it)
        maps.take(v.envs[e].maps);
        s_cellw = cellw;
        owner = cellw >= 0;
        const unsigned long long owners = __ballot(owner);
        const int wave = t >> 6, lane = t & 63;
        if (lane == 0) sh.count[wave] = __popcll(owners);
        __syncthreads();
        int base = 0;
        n_blk = 0;
#pragma unroll
        for (int w = 0; w < kRegTile / 64; ++w) {
            const int c = sh.count[w];
            base += w < wave ? c : 0;
            n_blk += c;
        }
        brank = base + __popcll(owners & ((1ull << lane) - 1ull));
        exchange(sh, v, t, cellw, 0);
        load_side(avg);
        load_rows(v, t);
    }
    template <int UMODE>
    __device__ __forceinline__ void update(DirectShared &sh, const View &v, int t, const float (&x)[CT], float row_scale,
                                           const int (&lut_reg)[(CT + 63) / 64], int pal_reg) {
        __shared__ __attribute__((aligned(16))) float s_rows[kDirectPre][CT];
        if (n_blk == 0) return;   // the same for the whole block
        if (v.exp & 1) return;
        const int lane = t & 63, sub = lane & 15;
        const int n_cls = classes(v);
        const float log2_c = log2f((float)n_cls);
        for (int r0 = 0; r0 < n_blk; r0 += kDirectPre) {
            if (r0 > 0) {   // rare: more owners in the block than were requested ahead
                exchange(sh, v, t, s_cellw, r0);
                load_side(UMODE == 1);
                load_rows(v, t);
            }
            if (owner && brank >= r0 && brank < r0 + kDirectPre) {
                float *row = s_rows[brank - r0];
#pragma unroll
                for (int q = 0; q < CT / SV; ++q) {
                    VecS val;
#pragma unroll
                    for (int j = 0; j < SV; ++j) val[j] = x[SV * q + j];
                    *reinterpret_cast<VecS *>(row + SV * q) = val;
                }
                if (UMODE == 1) sh.scale[brank - r0] = row_scale;
            }
            __syncthreads();
            if (v.exp & 2) continue;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int r = my_row(t, b);
                if (r0 + (r & ~3) >= n_blk) break;   // wave-uniform: no cell in this batch
                if (v.exp & 4) on[b] = false;
                VecF p[K];
                // the average family takes probabilities: the parked row times the owner's scale (the
                // same product the streaming part forms where it normalises); log-odds are ratios
                // within the row and take it as it is
                const float rs = UMODE == 1 ? sh.scale[on[b] ? r : 0] : 1.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = V * (sub + 16 * k);
                    p[k] = *reinterpret_cast<const VecF *>(&s_rows[on[b] ? r : 0][c < CT ? c : 0]);
                    const bool used = on[b] && c < n_cls;
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        if (UMODE == 1) p[k][j] *= rs;
                        p[k][j] = used ? p[k][j] : 0.0f;
                        mv[b][k][j] = used ? mv[b][k][j] : 0.0f;
                    }
                }
                direct_cell<UMODE, CT, V, K>(v, maps, n_cls, log2_c, on[b], lane, cell[b], (UMODE == 1 && on[b]) ? w_old[b] : 0.0f,
                                          p, mv[b], lut_reg, pal_reg);
            }
            if (r0 + kDirectPre < n_blk) __syncthreads();   // another pass reuses the LDS rows (the same for the whole block)
        }
    }
    int s_cellw;   // the lane's cell word, kept for the passes beyond the first
};

// ---------------------------------------------------------------- register kernel
// One pixel per thread.  The C class planes of the env's frame are read through one buffer
// descriptor: lane offset (pixel) in a vector register, plane offset in a scalar register, so
// the C loads need no per-lane 64-bit address arithmetic.  Loads are nontemporal: the frame is
// read once and should not push the map lines out of L2.  (Two or four pixels per thread with
// wider loads measured slower: the larger register footprint costs more occupancy than the
// wider requests gain.)
template <int CT, int MODE, int POOL, bool EXACT>
__device__ __forceinline__ void stream_reg_body(const View &v, const StreamArgs &a) {
    extern __shared__ float s_pool[];   // POOL: [pool_slots][kRegTile], a column per thread
    const int e = blockIdx.y;
    const int t = threadIdx.x;
    const int pix = blockIdx.x * kRegTile + t;
    const bool active = pix < v.N;
    const int64_t fbase = (int64_t)e * v.N;
    const int boff = active ? pix * 4 : 0;
    const int plane_bytes = v.N * 4;          // C * N * 4 < 2^31 is checked by the host
    // EXACT: the frame has CT classes.  Otherwise it has v.C <= CT of them: the planes beyond read as
    // zero (they lie outside the buffer descriptor) and, for logits, count as -inf: exponential 0,
    // no part in the maximum, the sums, the arg-max or the entropy
    static_assert(EXACT || MODE == 0 || MODE >= 3, "staged rows are written by the exact variants and the generic kernel");
    const int n_cls = EXACT ? CT : v.C;
    const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.in + ((int64_t)e * n_cls) * v.N), 0, n_cls * plane_bytes, 0x00020000);

    // the class -> schema table, spread over the lanes (lane l holds entries l, l + 64, ...): asked
    // for in front of the class planes and looked up with a lane permute at the end, so that no
    // wave ends on a dependent gather through the saturated memory pipeline
    constexpr bool DIRECT = MODE >= 3;            // the owners' map cells are updated here (modes 3, 4)
    constexpr int UMODE = DIRECT ? MODE - 2 : MODE;   // 1: average family, 2: log-odds
    constexpr bool NORM = MODE == 1 || MODE == 2;     // staged rows: the owners write probabilities
    constexpr int kLutRegs = (CT + 63) / 64;
    int lut_reg[kLutRegs];
    int cellw = -1, pal_reg = 0, slot = -1;
    float x[CT];
    // The per-pixel word first (the owners' map rows are requested as soon as it is there), then the
    // class planes, then the small tables: nothing that hangs on a chain of scalar loads (the skipped
    // flag of the env's pose) stands between the block's start and its plane loads.
    if (DIRECT) {
        const __amdgpu_buffer_rsrc_t cells = __builtin_amdgcn_make_buffer_rsrc(v.ocell + fbase, 0, v.N * 4, 0x00020000);
        cellw = __builtin_amdgcn_raw_buffer_load_b32(cells, active ? boff : -1, 0, 2 /* nt */);
    }
    if (MODE != 0 && !DIRECT) {
        const __amdgpu_buffer_rsrc_t slots = __builtin_amdgcn_make_buffer_rsrc(v.target + fbase, 0, v.N * 4, 0x00020000);
        slot = __builtin_amdgcn_raw_buffer_load_b32(slots, active ? boff : -1, 0, 2 /* nt */);
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
        x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frame, boff, c * plane_bytes, 2 /* nt */));
    __builtin_amdgcn_sched_barrier(0);
    if (a.from_logits || DIRECT) {
#pragma unroll
        for (int k = 0; k < kLutRegs; ++k) {
            const int idx = (t & 63) + 64 * k;
            lut_reg[k] = v.lut[idx < n_cls ? idx : 0];
        }
    }
    if (DIRECT) {
        const int row = (t & 63) < v.palette_rows ? (t & 63) : 0;
        const uint8_t *rgb = v.palette + 3 * row;
        pal_reg = (int)rgb[0] | ((int)rgb[1] << 8) | ((int)rgb[2] << 16);
    }
    // (an env without a frame in this batch keeps last step's words: it owns and stages nothing now)
    if (!active || v.poses[e].skip) { cellw = -1; slot = -1; }
    float row_scale = 1.0f;   // what turns x into probabilities (1 / sum while x holds the exponentials)
    DirectRows<DIRECT ? CT : 4, EXACT> dr;
    __shared__ DirectShared dsh;
    if constexpr (DIRECT) dr.request(dsh, v, e, t, cellw, UMODE == 1);
    if (a.from_logits) {
        // no implicit fused multiply-add in here: the roundings are those of the reference's
        // y = x / T, y - max(y)
#pragma clang fp contract(off)
        // softmax(y), y = x * (1 / T) rounded to float32 like the reference's scaled logits, so that
        // logits of any magnitude give the reference's differences y - max(y); the exponential in
        // base 2: exp(d) = exp2(d log2(e)), exactly 1 at the max
        float m = -INFINITY;
        if constexpr (!EXACT) {
#pragma unroll
            for (int c = 0; c < CT; ++c) x[c] = c < n_cls ? x[c] : -INFINITY;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) { x[c] *= a.inv_t; m = fmaxf(m, x[c]); }
        // arg-max on the way: y - max is +0 exactly at the classes that reach the max and negative
        // (sign bit set) at the others, so the unsigned minimum of bits(y - max) | c is the lowest
        // class at the max, like argmax's tie rule (an or and half a three-way minimum per class)
        float s = 0.0f;
        uint32_t key = 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float d = x[c] - m;
            key = min(key, __builtin_bit_cast(uint32_t, d) | (uint32_t)c);
            x[c] = __builtin_amdgcn_exp2f(d * 1.44269504088896341f);
            s += x[c];
            // (the tag chain is pinned here: it feeds only the label store, and sunk into that branch
            // it would keep every difference alive beside the exponentials)
            if (c % SSMAP_SCHED_GROUP == SSMAP_SCHED_GROUP - 1) { asm volatile("" : "+v"(key)); __builtin_amdgcn_sched_barrier(0); }
        }
        const float inv_s = 1.0f / s;   // the class at the max has exp2(0) = 1: the largest probability is inv_s itself
        // (no class at the max: NaN logits; class 0 like an argmax over NaNs)
        const int bi = (key >> 31) ? 0 : min((int)(key & 0xFFu), n_cls - 1);
        float h_bits;   // entropy in bits
        if constexpr (NORM || POOL) {
            // the probabilities are taken for every pixel (staged rows, pooled confidence): formed in place
            float hsum = 0.0f;
#pragma unroll
            for (int c = CT - 1; c >= 0; --c) {
                x[c] *= inv_s;
                hsum = __builtin_fmaf(x[c], __builtin_amdgcn_logf(x[c] + 1e-6f), hsum);
                if (c % SSMAP_SCHED_GROUP == 0) { asm volatile("" : "+v"(hsum)); __builtin_amdgcn_sched_barrier(0); }
            }
            h_bits = -hsum;
        } else {
            // from the exponentials as they are: with p = e / s,
            //   -sum p log2(p + eps) = log2(s) - (1 / s) sum e log2(e + eps s),
            // so no probability is formed per pixel; the direct update scales its few owner rows
            const float eps_s = 1e-6f * s;
            float hsum = 0.0f;
#pragma unroll
            for (int c = CT - 1; c >= 0; --c) {
                hsum = __builtin_fmaf(x[c], __builtin_amdgcn_logf(x[c] + eps_s), hsum);
                // (pinned like the tag chain: sunk into the branch of its store it would keep the
                // exponentials alive beside whatever follows)
                if (c % SSMAP_SCHED_GROUP == 0) { asm volatile("" : "+v"(hsum)); __builtin_amdgcn_sched_barrier(0); }
            }
            h_bits = __builtin_amdgcn_logf(s) - hsum * inv_s;
            if (a.probs_out) {
#pragma unroll
                for (int c = 0; c < CT; ++c) x[c] *= inv_s;
            } else {
                row_scale = inv_s;
            }
        }
        int sid = __builtin_amdgcn_ds_bpermute((bi & 63) << 2, lut_reg[0]);
#pragma unroll
        for (int k = 1; k < kLutRegs; ++k) {
            const int other = __builtin_amdgcn_ds_bpermute((bi & 63) << 2, lut_reg[k]);
            sid = (bi >> 6) == k ? other : sid;
        }
        float conf = inv_s;
        if (POOL) {
            // probability mass of the goal's listed model classes, added in list order like the
            // reference's probs[ids].sum(axis=0) (infer.py:109-118, 318-327); an empty list gives 0.
            // Every class that some goal pools parks its probability in the thread's LDS column
            // (membership is a scalar bit test: no vector work for the other classes; slots are
            // numbered in the order of this walk); a goal lane then adds its list's slots in list
            // order.  The slot numbers and list lengths travel by value with the kernel arguments.
            if (__ballot(sid >= 3 && sid <= 8)) {
                int pj = 0;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    // POOL 2: the lists are those of the reference's SegFormer wrapper (checked by the
                    // launcher), so membership is known when the loop is unrolled; otherwise one
                    // scalar test and branch per class
                    const bool member = POOL == 2 ? ((kAdePoolMask[c >> 5] >> (c & 31)) & 1u) != 0u
                                                  : ((v.pool_mask[c >> 5] >> (c & 31)) & 1u) != 0u;
                    if (member) { s_pool[pj * kRegTile + t] = x[c]; ++pj; }
                }
                for (int g = 0; g < 6; ++g) {
                    if (!__ballot(sid == g + 3)) continue;
                    const int n = (int)((v.pool_pack[12] >> (4 * g)) & 0xFu);
                    float acc = 0.0f;
                    for (int k = 0; k < n; ++k) {
                        const int at = g * SSMAP_MAX_POOL + k;
                        const int slot_k = (int)((v.pool_pack[at >> 2] >> (8 * (at & 3))) & 0xFFu);
                        const float pv = s_pool[slot_k * kRegTile + t];
                        acc = k == 0 ? pv : acc + pv;
                    }
                    conf = sid == g + 3 ? acc : conf;
                }
            }
        }
        if (active) {
            const float unc = h_bits * (0.693147180559945309f / logf((float)n_cls));
            // device-scope (write-through) stores: beside the streaming reads they measured 2-3 us
            // faster than plain or nontemporal ones (tools/store_probe.hip)
            // Stage 1 alone (the label rules' steps): device-scope (write-through) stores, 6 us faster
            // at 40 classes than nontemporal ones beside the streaming reads; with a rule update in
            // the kernel, or 150 classes, the nontemporal ones are (3-10 us)
            if constexpr (MODE == 0 && CT <= 64) {
                if (a.sem_out) __hip_atomic_store(a.sem_out + fbase + pix, sid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.unc_out) __hip_atomic_store(a.unc_out + fbase + pix, unc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.maxp_out) __hip_atomic_store(a.maxp_out + fbase + pix, conf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (a.sem_out) __builtin_nontemporal_store(sid, a.sem_out + fbase + pix);
                if (a.unc_out) __builtin_nontemporal_store(unc, a.unc_out + fbase + pix);
                if (a.maxp_out) __builtin_nontemporal_store(conf, a.maxp_out + fbase + pix);
            }
            if (a.probs_out) {
                float *dst = a.probs_out + ((int64_t)e * n_cls) * v.N + pix;
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    if (EXACT || c < n_cls) dst[(int64_t)c * v.N] = x[c];
            }
        }
    }
    if (MODE == 0) return;
    if constexpr (DIRECT) {
        dr.template update<UMODE>(dsh, v, t, x, row_scale, lut_reg, pal_reg);
        return;
    }

    // The owners' rows leave the wave through LDS: the few owner lanes (3 to 4 of 64 on a typical
    // frame) park their C probabilities in the wave's staging rows, then consecutive lanes write
    // consecutive vectors of those rows, so that a wave issues one or two store instructions
    // instead of C / 4 under a sparse execution mask (a vector store occupies the memory pipeline
    // for its full width whatever the mask).  Wave-local: no block barrier, the LDS accesses of one
    // wave are executed in order.
    constexpr int V = CT % 4 == 0 ? 4 : (CT % 2 == 0 ? 2 : 1);
    constexpr int U = CT / V;
    typedef float VecT __attribute__((ext_vector_type(V)));
    constexpr int kLdsRows = (CT % 4 == 0 && CT <= 64) ? kStageRows : 1;
    __shared__ __attribute__((aligned(16))) float s_rows[kRegTile / 64][kLdsRows][CT];
    __shared__ int s_slot[kRegTile / 64][kLdsRows];
    if (!(CT % 4 == 0 && CT <= 64)) {
        // wide rows: the staging rows would cost more than they save (150 classes measured 6 %
        // slower through LDS, rows padded to 16 bytes and written as float4 2 % slower); the owner
        // writes its own row with the widest vector that divides the row
        if (slot >= 0) {
            float *dst = v.wprob + (fbase + slot) * v.wprob_stride;
#pragma unroll
            for (int q = 0; q < U; ++q) {
                VecT val;
#pragma unroll
                for (int j = 0; j < V; ++j) val[j] = x[V * q + j];
                *reinterpret_cast<VecT *>(dst + V * q) = val;
            }
        }
        return;
    }
    const unsigned long long owners = __ballot(slot >= 0);
    if (owners == 0ull) return;
    const int wave = t >> 6, lane = t & 63;
    const int n_own = __popcll(owners);
    const int rank = __popcll(owners & ((1ull << lane) - 1ull));
    for (int r0 = 0; r0 < n_own; r0 += kStageRows) {
        if (slot >= 0 && rank >= r0 && rank < r0 + kStageRows) {
            float *row = s_rows[wave][rank - r0];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                VecT val;
#pragma unroll
                for (int j = 0; j < V; ++j) val[j] = x[V * q + j];
                *reinterpret_cast<VecT *>(row + V * q) = val;
            }
            s_slot[wave][rank - r0] = slot;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int rows = min(kStageRows, n_own - r0);
        for (int j = lane; j < rows * U; j += 64) {
            const int k = j / U, q = j - k * U;
            const VecT val = *reinterpret_cast<const VecT *>(&s_rows[wave][k][V * q]);
            *reinterpret_cast<VecT *>(v.wprob + (fbase + s_slot[wave][k]) * v.wprob_stride + V * q) = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// Stage 1 alone is fastest with few waves per SIMD (three at 40 classes); narrower kernels take more of
// them, so that about as many plane loads stay in flight.
constexpr int stage1_max_waves(int ct) { return ct >= 40 ? SSMAP_WPE0MAX : (120 / ct > 8 ? 8 : 120 / ct); }
#define SSMAP_STREAM_BOUNDS(CT, MODE) __launch_bounds__(kRegTile) __attribute__((amdgpu_waves_per_eu( \
    CT > 128 ? SSMAP_WPE150 : (MODE >= 3 ? SSMAP_WPED : SSMAP_WPE0), \
    CT > 128 ? SSMAP_WPE150 : (MODE == 0 ? stage1_max_waves(CT) : (MODE >= 3 ? SSMAP_WPEDMAX : 8)))))
// the frame has exactly CT classes
template <int CT, int MODE, int POOL>
__global__ SSMAP_STREAM_BOUNDS(CT, MODE) void k_stream_reg(View v, StreamArgs a) {
    stream_reg_body<CT, MODE, POOL, true>(v, a);
}
// the frame has v.C <= CT classes
template <int CT, int MODE, int POOL>
__global__ SSMAP_STREAM_BOUNDS(CT, MODE) void k_stream_reg_upto(View v, StreamArgs a) {
    stream_reg_body<CT, MODE, POOL, false>(v, a);
}

// ---------------------------------------------------------------- generic (LDS column) kernel
// Class counts without a compiled register variant: a thread's classes live in a private LDS column
// (stride 257 floats: conflict-free both down a column and across the columns of a wave).  The
// arithmetic is the register kernel's (scaled logits differenced against their maximum, base-2
// exponential and logarithm, arg-max from the tagged differences, entropy from the unnormalised
// exponentials), so both paths round alike.  Owners' rows leave through their wave: the lanes read
// the owner's column across the classes and write the row in one piece.
template <int MODE>
__global__ __launch_bounds__(kTile) void k_stream_lds(View v, StreamArgs a) {
    extern __shared__ float tile[];   // [C][kStride]

    const int e = blockIdx.y;
    const int t = threadIdx.x;
    const int pix = blockIdx.x * kTile + t;
    const bool active = pix < v.N;
    const int C = v.C;
    const int64_t fbase = (int64_t)e * v.N;
    const float *src = a.in + ((int64_t)e * C) * v.N + (active ? pix : 0);
    float *col = tile + t;
    const float pre = a.from_logits ? a.inv_t : 1.0f;   // (x * 1.0f is x)

    float m = -INFINITY;
    for (int c0 = 0; c0 < C; c0 += 8) {   // 8 loads in flight per thread
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = (c0 + j < C) ? src[(int64_t)(c0 + j) * v.N] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < C) {
                const float y = tmp[j] * pre;
                col[(c0 + j) * kStride] = y;
                m = fmaxf(m, y);
            }
    }
    float row_scale = 1.0f;   // what turns the column into probabilities
    if (a.from_logits) {
#pragma clang fp contract(off)
        float s = 0.0f;
        uint32_t key = 0xFFFFFFFFu;
        for (int c = 0; c < C; ++c) {
            const float d = col[c * kStride] - m;
            key = min(key, __builtin_bit_cast(uint32_t, d) | (uint32_t)c);
            const float ev = __builtin_amdgcn_exp2f(d * 1.44269504088896341f);
            col[c * kStride] = ev;
            s += ev;
        }
        const float inv_s = 1.0f / s;
        const int bi = (key >> 31) ? 0 : min((int)(key & 0xFFu), C - 1);
        const float eps_s = 1e-6f * s;
        float hsum = 0.0f;
        for (int c = C - 1; c >= 0; --c) {
            const float ev = col[c * kStride];
            hsum = __builtin_fmaf(ev, __builtin_amdgcn_logf(ev + eps_s), hsum);
        }
        const float h_bits = __builtin_amdgcn_logf(s) - hsum * inv_s;
        const int sid = v.lut[bi];
        float conf = inv_s;
        if (!a.pool_trivial && sid >= 3 && sid <= 8) {
            const int g = sid - 3;
            const int n = v.pool_count[g];
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) {
                const float pv = col[v.pool_ids[g * SSMAP_MAX_POOL + k] * kStride] * inv_s;
                acc = k == 0 ? pv : acc + pv;
            }
            conf = acc;   // an empty list sums to 0 (infer.py:116)
        }
        if (active) {
            if (a.sem_out) a.sem_out[fbase + pix] = sid;
            if (a.unc_out) a.unc_out[fbase + pix] = h_bits * (0.693147180559945309f / logf((float)C));
            if (a.maxp_out) a.maxp_out[fbase + pix] = conf;
            if (a.probs_out) {
                float *dst = a.probs_out + ((int64_t)e * C) * v.N + pix;
                for (int c = 0; c < C; ++c) dst[(int64_t)c * v.N] = col[c * kStride] * inv_s;
            }
        }
        row_scale = inv_s;
    }
    if (MODE == 0) return;
    const int slot = (active && !v.poses[e].skip) ? v.target[fbase + pix] : -1;
    unsigned long long owners = __ballot(slot >= 0);
    if (owners == 0ull) return;
    // (the wave's own LDS accesses are executed in order: what an owner wrote above is there)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int lane = t & 63, wave_col = t & ~63;
    while (owners) {
        const int src_lane = __ffsll((long long)owners) - 1;
        owners &= owners - 1ull;
        const int s_own = __shfl(slot, src_lane);
        const float rs = __shfl(row_scale, src_lane);
        float *dst = v.wprob + (fbase + s_own) * v.wprob_stride;
        const float *ocol = tile + wave_col + src_lane;
        // (nontemporal: 1.3 us per step at 21 classes; the register kernels' rows measured slower that way)
        for (int c = lane; c < C; c += 64) __builtin_nontemporal_store(ocol[c * kStride] * rs, &dst[c]);
    }
}

// ---------------------------------------------------------------- rule update
// Stage 4 for the probability rules.  A block takes 64 consecutive list slots per pass.
//   rows: a wave works on four update cells at a time, one per 16-lane row; lane `sub` of a
//         row owns the classes V*(sub+16k) .. +V-1 (k < K), so a cell's C contiguous floats
//         move with K vector loads / stores per row and the per-cell sums and argmax are
//         16-lane DPP reductions.
//   side: the per-cell scalars (height, weight, uncertainty, label, colour, ground truth) are
//         handled with lane <-> list slot, one array per wave: list slots are in cell order,
//         so these are mostly consecutive addresses instead of four lanes per instruction.
// MODE 1: avg family (mapper.py:528-569), MODE 2: log-odds (mapper.py:574-603).
template <int MODE, int V, int K>
__global__ __launch_bounds__(256) void k_update_prob(View v, const int32_t *__restrict__ gt) {
    // every product and sum rounds on its own, as numpy's array statements do (mapper.py:540-548):
    // with unit weights the fused average then equals the reference's bit for bit
#pragma clang fp contract(off)
    typedef float VecF __attribute__((ext_vector_type(V)));
    constexpr int NB = V * K <= 4 ? 4 : 2;   // row-batches in flight per wave (of the 4 a pass needs; all four of wide rows measured slower)
    constexpr unsigned kRun = 64;     // list slots per block pass
    __shared__ int s_bi[kRun];
    __shared__ float s_wn[kRun], s_ent[kRun];
    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    const unsigned n = v.hdr[e].n_winners;
    const int2 *list = reinterpret_cast<const int2 *>(v.winners) + fbase;
    // the env's map pointers once, in registers: read through the struct they would be fetched
    // again after every store (the stores could alias the struct), one more round trip in front
    // of the row loads
    const ssmap_env_maps &mp_ref = v.envs[e].maps;
    const gptr<float> m_prob = as_global(mp_ref.prob_map), m_height = as_global(mp_ref.height_map);
    const gptr<float> m_weight = as_global(mp_ref.weight_map), m_unc = as_global(mp_ref.unc_map);
    const gptr<uint32_t> m_gt = as_global(mp_ref.gt_map), m_sem = as_global(mp_ref.semantic_map);
    const gptr<uint8_t> m_color = as_global(mp_ref.semantic_map_color);
    const double agent_y = v.poses[e].agent[1];
    const int C = v.C;
    const float log2_c = log2f((float)C);
    const int lane = threadIdx.x & 63, sub = lane & 15, row = lane >> 4;
    const unsigned wave = threadIdx.x >> 6;
    for (unsigned r0 = blockIdx.x * kRun; r0 < n; r0 += gridDim.x * kRun) {
        // ---- side data of list slot r0 + lane (every wave: the new height feeds its stores)
        // Loads are issued for every lane, from a safe address where the lane has nothing to do
        // (slot 0, cell 0): a load inside a divergent branch is waited for at the end of that
        // branch, which would turn the independent loads below into one round trip each.
        const unsigned is = r0 + lane;
        const int2 ps = list[is < n ? is : 0u];
        const bool side_live = is < n && !((uint32_t)ps.x & kLostBit);
        const int side_cell = side_live ? ps.y : 0;
        const float h_old = m_height[side_cell];
        const unsigned long long hk_side = v.wheight[fbase + (is < n ? is : 0u)];
        int gtv = 0;
        if (gt) gtv = gt[fbase + (side_live ? (int)((uint32_t)ps.x & kPixMask) : 0)];
        const float hnew = fmaxf(h_old, (float)(unorder_f64(hk_side) - agent_y));
        // ---- rows
#pragma unroll 1
        for (unsigned b0 = 0; b0 < 4; b0 += NB) {
            VecF p[NB][K], mv[NB][K];
            float w_old[NB];
            int cell[NB];
            bool live[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                // the list entry of this row's slot sits in lane j of the side load above
                const int j = 16 * wave + 4 * (b0 + b) + row;
                const unsigned i = r0 + j;
                const int2 pc = make_int2(__shfl(ps.x, j), __shfl(ps.y, j));
                live[b] = i < n && !((uint32_t)pc.x & kLostBit);
                cell[b] = live[b] ? pc.y : 0;
                // unconditional loads (cell 0 / slot 0 / class 0 where there is nothing to do), the
                // unused values are zeroed afterwards
                const gptr<const float> mrow = m_prob + (int64_t)cell[b] * C;
                const float *prow = v.wprob + (fbase + (live[b] ? i : 0u)) * v.wprob_stride;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int c = V * (sub + 16 * k);
                    const int cs = c < C ? c : 0;
                    mv[b][k] = *(gptr<const VecF>)(mrow + cs);
                    p[b][k] = *reinterpret_cast<const VecF *>(prow + cs);
                }
                w_old[b] = MODE == 1 ? m_weight[cell[b]] : 0.0f;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool used = live[b] && V * (sub + 16 * k) < C;
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        mv[b][k][j] = used ? mv[b][k][j] : 0.0f;
                        p[b][k][j] = used ? p[b][k][j] : 0.0f;
                    }
                }
                if (!live[b]) w_old[b] = 0.0f;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const gptr<float> mrow = m_prob + (int64_t)cell[b] * C;
                int bi;
                float wn, ent_out;
                update_row<MODE, V, K>(v, C, log2_c, live[b], sub, mrow, p[b], mv[b], w_old[b], bi, wn, ent_out);
                if (sub == 0) {
                    const unsigned j = 16 * wave + 4 * (b0 + b) + row;
                    s_bi[j] = bi;
                    if (MODE == 1) { s_wn[j] = wn; s_ent[j] = ent_out; }
                }
            }
        }
        __syncthreads();
        // ---- side stores: height map and ground-truth map (mapper.py:338-341, 361-363), label, colour
        if (side_live) {
            const int cell = ps.y;
            const int id = occupancy_override(v.lut[s_bi[lane]], hnew, v.use_height, v.n_schema);
            if (wave == 0) {
                m_height[cell] = hnew;
                if (MODE == 1) m_weight[cell] = s_wn[lane];
            } else if (wave == 1) {
                if (gt) m_gt[cell] = (uint32_t)gtv;
                if (MODE == 1) m_unc[cell] = s_ent[lane];
            } else if (wave == 2) {
                m_sem[cell] = (uint32_t)id;
            } else {
                const uint8_t *rgb = v.palette + 3 * id;
                const uint8_t r = rgb[0], g = rgb[1], b = rgb[2];
                gptr<uint8_t> dst = m_color + 3ll * cell;
                dst[0] = r; dst[1] = g; dst[2] = b;
            }
        }
        __syncthreads();
    }
}

template <int CT, int POOL>
static void launch_reg_pool(int mode, hipStream_t s, const View &v, const StreamArgs &a) {
    dim3 grid((v.N + kRegTile - 1) / kRegTile, v.E);
    const size_t lds = (POOL ? (size_t)(v.pool_slots > 0 ? v.pool_slots : 1) * kRegTile * sizeof(float) : 0) + (size_t)v.exp_lds;
    if (mode == 0) hipLaunchKernelGGL((k_stream_reg<CT, 0, POOL>), grid, dim3(kRegTile), lds, s, v, a);
    else if (mode == 1) hipLaunchKernelGGL((k_stream_reg<CT, 1, POOL>), grid, dim3(kRegTile), lds, s, v, a);
    else if (mode == 2) hipLaunchKernelGGL((k_stream_reg<CT, 2, POOL>), grid, dim3(kRegTile), lds, s, v, a);
    else if constexpr (exact_class_count(CT)) {   // direct update of the owners' map cells
        if (mode == 3) hipLaunchKernelGGL((k_stream_reg<CT, 3, POOL>), grid, dim3(kRegTile), lds, s, v, a);
        else hipLaunchKernelGGL((k_stream_reg<CT, 4, POOL>), grid, dim3(kRegTile), lds, s, v, a);
    }
}
template <int CT>
static void launch_reg(int mode, hipStream_t s, const View &v, const StreamArgs &a) {
    // the pooled confidence needs its LDS columns only when some goal pools more than its own class
    if (a.from_logits && !a.pool_trivial) {
        if (CT == 150 && memcmp(v.pool_mask, kAdePoolMask, sizeof(kAdePoolMask)) == 0) launch_reg_pool<CT, CT == 150 ? 2 : 1>(mode, s, v, a);
        else launch_reg_pool<CT, 1>(mode, s, v, a);
    } else {
        launch_reg_pool<CT, 0>(mode, s, v, a);
    }
}
// Any other class count up to CT: the same kernel compiled for CT class registers, of which the
// frame's v.C are live (stage 1 alone, or with the direct update; staged rows stay with the
// generic kernel).
template <int CT>
static void launch_reg_upto(int mode, hipStream_t s, const View &v, const StreamArgs &a) {
    dim3 grid((v.N + kRegTile - 1) / kRegTile, v.E);
    const bool pool = a.from_logits && !a.pool_trivial;
    const size_t lds = (pool ? (size_t)(v.pool_slots > 0 ? v.pool_slots : 1) * kRegTile * sizeof(float) : 0) + (size_t)v.exp_lds;
#define SSMAP_GO(M) do { if (pool) hipLaunchKernelGGL((k_stream_reg_upto<CT, M, 1>), grid, dim3(kRegTile), lds, s, v, a); \
                         else hipLaunchKernelGGL((k_stream_reg_upto<CT, M, 0>), grid, dim3(kRegTile), lds, s, v, a); } while (0)
    if (mode == 0) SSMAP_GO(0);
    else if (mode == 3) SSMAP_GO(3);
    else SSMAP_GO(4);
#undef SSMAP_GO
}

void launch_update_prob(const View &v, const ssmap_frames &f, hipStream_t s) {
    // 64 list slots per block pass: 384 blocks take a typical frame's 15-25 thousand update cells
    // in one pass each (a grid just short of that makes the whole kernel wait for a second pass)
    dim3 grid(384, v.E), block(256);
    const bool lo = v.rule == SSMAP_RULE_LOG_ODDS;
    // widest vector that divides C (every cell row and staged row then starts on a vector boundary;
    // engine scratch is 256-byte aligned, bound maps are checked when they are bound)
    const int vec = v.C % 4 == 0 ? 4 : (v.C % 2 == 0 ? 2 : 1);
    const int k = (v.C + 16 * vec - 1) / (16 * vec);
#define GO(V, K) do { if (lo) hipLaunchKernelGGL((k_update_prob<2, V, K>), grid, block, 0, s, v, f.semantic_gt); \
                      else hipLaunchKernelGGL((k_update_prob<1, V, K>), grid, block, 0, s, v, f.semantic_gt); } while (0)
    if (vec == 4) { if (k <= 1) GO(4, 1); else GO(4, 3); }
    else if (vec == 2) { if (k <= 2) GO(2, 2); else GO(2, 5); }
    else { if (k <= 3) GO(1, 3); else GO(1, 10); }
#undef GO
}

void launch_stream(const View &v, const ssmap_frames &f, bool fuse, float *probs_out, hipStream_t s) {
    StreamArgs a;
    a.in = f.logits ? f.logits : f.probs;
    a.from_logits = f.logits ? 1 : 0;
    a.inv_t = f.logits ? 1.0f / f.temperature : 1.0f;
    a.pool_trivial = v.pool_trivial;
    a.sem_out = f.semantic;
    a.unc_out = f.semantic_unc;
    a.maxp_out = f.semantic_max_prob;
    a.probs_out = probs_out;
    a.gt = f.semantic_gt;
    int mode = 0;
    if (fuse) mode = (v.rule == SSMAP_RULE_LOG_ODDS ? 2 : 1) + (v.direct_update ? 2 : 0);
    // the register kernels address a whole frame with 32-bit buffer offsets
    if ((int64_t)v.C * v.N * 4 < (1ll << 31) && !v.force_generic) {
        switch (v.C) {
            // the class counts of the reference's segmentation heads (EMSANet 8/33/37/40/53/81,
            // SegFormer ADE20k 150); any other count streams through LDS below
            case 8: launch_reg<8>(mode, s, v, a); return;
            case 33: launch_reg<33>(mode, s, v, a); return;
            case 37: launch_reg<37>(mode, s, v, a); return;
            case 40: launch_reg<40>(mode, s, v, a); return;
            case 53: launch_reg<53>(mode, s, v, a); return;
            case 81: launch_reg<81>(mode, s, v, a); return;
            case 150: launch_reg<150>(mode, s, v, a); return;
            default: break;
        }
        // any other count: the register kernel of the next size up (modes 1 and 2, staged rows, are
        // the generic kernel's: SSMAP_DIRECT=0)
        if (mode == 0 || mode >= 3) {
            if (v.C <= 16) launch_reg_upto<16>(mode, s, v, a);
            else if (v.C <= 24) launch_reg_upto<24>(mode, s, v, a);
            else if (v.C <= 32) launch_reg_upto<32>(mode, s, v, a);
            else if (v.C <= 48) launch_reg_upto<48>(mode, s, v, a);
            else if (v.C <= 64) launch_reg_upto<64>(mode, s, v, a);
            else if (v.C <= 96) launch_reg_upto<96>(mode, s, v, a);
            else if (v.C <= 128) launch_reg_upto<128>(mode, s, v, a);
            else launch_reg_upto<SSMAP_MAX_CLASSES>(mode, s, v, a);
            return;
        }
    }
    dim3 grid((v.N + kTile - 1) / kTile, v.E);
    const size_t lds = (size_t)v.C * kStride * sizeof(float);
    auto go = [&](auto kern) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(kTile), lds, s, v, a);
    };
    if (mode == 0) go(k_stream_lds<0>);
    else if (mode == 1) go(k_stream_lds<1>);
    else go(k_stream_lds<2>);
}

}  // namespace ssmap
