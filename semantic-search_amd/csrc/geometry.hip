// Stages 2 and 3 of the fusion path for gfx950:
//   k_geometry    depth -> world point -> map cell + height, one pass over the depth image that
//                 leaves one record per live pixel in the winner table of its cell
//                 (mapper.py:279-283 sentinels, :18-32 pointcloud, :305-311 rotate/translate,
//                  :851-863 world2map, :319-329 masks)
//   k_winners     tallest point per cell, ties -> lowest pixel index (mapper.py:331), then
//                 the height / depth / bounds filter (mapper.py:333-335)
//
// This file is compiled with -ffp-contract=off: the float64 geometry must round exactly
// like the CPU statement of the same formulas (explicit fma only where written).
#include "ssmap_internal.h"
#include <cstring>

namespace ssmap {
#ifdef SSMAP_TIMELINE
// variant builds only: wall-clock stamps (10 ns ticks) of the winner kernel's phases, per block
__device__ unsigned long long g_timeline[4096][12];
#define SSMAP_STAMP(k) do { if (threadIdx.x == 0) g_timeline[blockIdx.y * gridDim.x + blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define SSMAP_STAMP(k) do { } while (0)
#endif

// Wave reductions of the geometry partials without LDS traffic: four DPP row rotations reduce
// inside each 16-lane row, four lane reads combine the rows.  The cell box travels as two words
// of packed 16-bit biased coordinates {row, col} and {~row, ~col}, so that one packed minimum
// reduces two bounds at once (max(x) = ~min(~x)).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk_min16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
template <int N>
__device__ inline uint32_t ror16(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x120 + N, 0xF, 0xF, false);
}
__device__ inline uint32_t wave_pk_min16(uint32_t x) {
    x = pk_min16(x, ror16<8>(x)); x = pk_min16(x, ror16<4>(x)); x = pk_min16(x, ror16<2>(x)); x = pk_min16(x, ror16<1>(x));
    const uint32_t a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const uint32_t c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return pk_min16(pk_min16(a, b), pk_min16(c, d));
}
__device__ inline uint32_t wave_min_u32(uint32_t x) {
    x = min(x, ror16<8>(x)); x = min(x, ror16<4>(x)); x = min(x, ror16<2>(x)); x = min(x, ror16<1>(x));
    const uint32_t a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const uint32_t c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return min(min(a, b), min(c, d));
}

__device__ inline int wave_min(int x) {
    for (int o = 32; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o));
    return x;
}
__device__ inline int wave_max(int x) {
    for (int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o));
    return x;
}
__device__ inline unsigned wave_minu(unsigned x) {
    for (int o = 32; o > 0; o >>= 1) x = min(x, (unsigned)__shfl_xor((int)x, o));
    return x;
}

// Correctly rounded a / b for a positive divisor known in advance (rb = RN(1/b) from the host,
// rb_wide = rb (1 + 2^-20)).  Two fused steps give a candidate q, the residual r = a - b q says
// how far it is from the quotient: a / b = q + r / b.  r is exact whenever q is one of the two
// doubles around a / b; then RN(q + r rb_wide) == q PROVES |r / b| < the distance from q to the
// edge of its rounding interval (r rb_wide exceeds r / b in magnitude: b rb >= 1 - 2^-53), i.e.
// q = RN(a / b).  A q further off has |r / b| of an ulp or more and fails the test.  `sure` is
// cleared when the proof fails; the caller then redoes the whole pixel with hardware divisions
// (one rare branch per pixel instead of one per division).  a = +0 gives +0 like a / b.
__device__ inline double div_known(double a, double b, double rb, double rb_wide, bool &sure) {
    const double q0 = a * rb;
    const double q = __builtin_fma(__builtin_fma(-b, q0, a), rb, q0);
    const double r = __builtin_fma(-b, q, a);
    sure &= __builtin_fma(r, rb_wide, q) == q;
    return q;
}

// floor(RN(d / p)) for a divisor known in advance: d * RN(1/p) is within 3.4e-16 relative of
// the rounded quotient, so away from the integers (guard band 1e-7, |q| < 2^24) both floors
// agree; inside the band `sure` is cleared.
__device__ inline double floor_div_known(double d, double rp, bool &sure) {
    const double qf = d * rp;
    const double fl = floor(qf);
    const double frac = qf - fl;
    sure &= (fabs(qf) < 16777216.0) & (frac > 1.0e-7) & (frac < 1.0 - 1.0e-7);
    return fl;
}

// Table of a cell in the winner selection (see the binning section below): independent of the
// frame's cell box, so the geometry kernel can already count its pixels per table.
constexpr int kGeomBinParts = kMaxParts;
__device__ inline int geom_cell_table(uint32_t id) {
    const int row_b = (int)((id >> kCoordBits) & kCoordMask), col_b = (int)(id & kCoordMask);
    return ((col_b >> 3) + row_b * 3) & (kGeomBinParts - 1);
}

// What the path needs to know about one pixel.
struct PixelGeom {
    uint32_t id;                // biased cell | kUpdBit, or kDeadCell when it takes no part
    unsigned long long key;     // order-preserving image of its height
    bool in_box;                // cell inside [-dim, dim): counts for the frame's cell box
    int row, col;
    bool index_error;           // passes the filters but lies below -dim (numpy raises)
    bool invalid;               // depth <= 0 after sentinel zeroing
};

// depth -> world point -> floored map coordinates (mapper.py:18-32, 305-311, 851-863); dk is the
// depth after sentinel zeroing, (u, yr) = (column, row - H/2).  An invalid depth (ok false) sits at
// the camera position.  FAST uses the verified reciprocal forms above and reports through `sure`
// whether every step was proven exact.
template <bool FAST>
__device__ inline void pixel_world(const View &v, const PoseBlock &pb, const ssmap_env_maps &mp, float dk, int u,
                                   double yr, bool &sure, bool &ok, double &wy, double &fr, double &fc) {
    ok = dk > 0.0f;
    double x = 0.0, y = 0.0, z = 0.0;
    if (ok) {
        z = (double)dk;
        const double ax = z * ((double)u - (double)v.W / 2.0), ay = z * yr;
        x = FAST ? div_known(ax, v.fx, v.inv_fx, v.inv_fx_wide, sure) : ax / v.fx;
        y = FAST ? div_known(ay, v.fy, v.inv_fy, v.inv_fy_wide, sure) : ay / v.fy;
    }
    const double wx = __builtin_fma(pb.rot[2], z, __builtin_fma(pb.rot[1], y, pb.rot[0] * x)) + pb.trans[0];
    wy = __builtin_fma(pb.rot[5], z, __builtin_fma(pb.rot[4], y, pb.rot[3] * x)) + pb.trans[1];
    const double wz = __builtin_fma(pb.rot[8], z, __builtin_fma(pb.rot[7], y, pb.rot[6] * x)) + pb.trans[2];
    fr = FAST ? floor_div_known(wz - mp.lower[2], v.inv_p2m, sure) : floor((wz - mp.lower[2]) / v.p2m);
    fc = FAST ? floor_div_known(wx - mp.lower[0], v.inv_p2m, sure) : floor((wx - mp.lower[0]) / v.p2m);
}

// The masks of mapper.py:326-338 on floored coordinates, in double so that huge or non-finite
// coordinates never reach a cast.  numpy casts the floored coordinate to int64 (mapper.py:863), and
// the x86 conversion turns a NaN, an infinity and any value beyond the int64 range into INT64_MIN:
// such a coordinate passes the upper-bound mask and is an index far below -dim.
__device__ inline void cell_masks(double fr, double fc, int Hm, int Wm, bool &below_upper, bool &above_lower) {
    const bool r_min = !(fr < 9223372036854775808.0), c_min = !(fc < 9223372036854775808.0);
    below_upper = (r_min || fr < (double)Hm) && (c_min || fc < (double)Wm);
    above_lower = !r_min && !c_min && fr >= -(double)Hm && fc >= -(double)Wm;
}

// mapper cell and height of one pixel (mapper.py:319-329) from pixel_world.
template <bool FAST>
__device__ inline PixelGeom pixel_geometry_impl(const View &v, const PoseBlock &pb, const ssmap_env_maps &mp,
                                                float dk, int u, double yr, bool &sure) {
    PixelGeom g;
    const int Hm = mp.map_h, Wm = mp.map_w;
    bool ok;
    double wy, fr, fc;
    pixel_world<FAST>(v, pb, mp, dk, u, yr, sure, ok, wy, fr, fc);
    g.id = kDeadCell;
    g.key = 0;
    g.row = g.col = 0;
    g.invalid = !ok;
    const bool pass = ok && (wy < pb.h_cut);
    if (FAST) {
        // a proven pixel has |coordinate / cell| < 2^24 and finite: integer comparisons
        const int row = (int)fr, col = (int)fc;
        const bool below_upper = row < Hm && col < Wm;
        const bool above_lower = row >= -Hm && col >= -Wm;
        g.in_box = below_upper && above_lower;
        g.index_error = below_upper && !above_lower && pass;
    } else {
    bool below_upper, above_lower;
    cell_masks(fr, fc, Hm, Wm, below_upper, above_lower);
    g.in_box = below_upper && above_lower && wy == wy;
    g.index_error = !g.in_box && below_upper && !above_lower && pass;   // index below -dim: numpy raises IndexError
    }
    if (g.in_box) {
        g.row = (int)fr;
        g.col = (int)fc;
        if (ok) {
            g.id = ((uint32_t)(g.row + kCoordBias) << kCoordBits) | (uint32_t)(g.col + kCoordBias);
            if (pass) g.id |= kUpdBit;
            g.key = order_f64(wy);
        }
    }
    return g;
}
__device__ inline PixelGeom pixel_geometry_exact(const View &v, const PoseBlock &pb, const ssmap_env_maps &mp,
                                                       float dk, int u, double yr) {
    bool unused = true;
    return pixel_geometry_impl<false>(v, pb, mp, dk, u, yr, unused);
}
__device__ inline PixelGeom pixel_geometry(const View &v, const PoseBlock &pb, const ssmap_env_maps &mp,
                                           float dk, int u, double yr) {
    bool sure = true;
    PixelGeom g = pixel_geometry_impl<true>(v, pb, mp, dk, u, yr, sure);
    if (!sure) g = pixel_geometry_exact(v, pb, mp, dk, u, yr);   // rare: a quotient too close to call
    return g;
}

// Binning: every live pixel of a frame leaves a 12-byte record {height image, slot << 19 | pixel}
// with the winner table of its cell, so that a winner-table block reads only its own records
// instead of every pixel of the frame.  Record order inside a table is arbitrary (the table
// reductions are order independent).
//
// Deal of cells to tables: runs of 8 cells of one map row ("chunks", numbered cc along the row
// in absolute biased coordinates) go to table (cc + 3 row) mod 16.  The table of a cell thus does
// not depend on any cell box; the skew by 3 per row spreads a wall seen edge-on over all tables.
// Inside a table a cell's slot is dense over the frame's predicted cell box (rmin.., from the
// host) padded to a multiple of 16 chunks per row:
//   slot = ((row - rmin) * per_row + (cc - cc0) / 16) * 8 + col % 8,   per_row = ceil(ncc / 16).
// A box whose padded cells exceed 16 tables of kSlots (about 131 thousand cells: far beyond what
// a 5 m depth range reaches at 3 cm) is not binned; the winner kernel scans the pixels then.
constexpr int kBinParts = kMaxParts;          // 16 tables per frame
constexpr int kBinShift = 4;
constexpr int kChunkShift = 3, kChunkMask = (1 << kChunkShift) - 1;
constexpr int kRowSkew = 3;
static_assert((1 << kBinShift) == kBinParts, "kBinShift must be log2 of kBinParts");

struct BinFrame {
    bool on;
    int rmin_b, cc0, per_row;
    unsigned per_row_magic;   // floor(2^20 / per_row) + 1: (k * magic) >> 20 == k / per_row for k < 2048, per_row < 200
};
static_assert(kChunkShift == 3 && kRowSkew == 3, "geom_cell_table spells these constants out");
__device__ inline int cell_table(uint32_t id) { return geom_cell_table(id); }
__device__ inline BinFrame bin_frame_box(const int32_t *box) {   // box = {rmin, rmax, cmin, cmax}
    BinFrame f;
    f.on = false;
    const int rmin = box[0], rmax = box[1], cmin = box[2], cmax = box[3];
    if (rmin > rmax || cmin > cmax) return f;
    f.rmin_b = rmin + kCoordBias;
    f.cc0 = (cmin + kCoordBias) >> kChunkShift;
    const int cc1 = (cmax + kCoordBias) >> kChunkShift;
    f.per_row = (cc1 - f.cc0 + kBinParts) >> kBinShift;
    f.on = ((long long)(rmax - rmin + 1) * f.per_row << kChunkShift) <= kSlots;
    // (a usable frame has at most kSlots / 8 chunks per table and, maps being at most 8192 cells
    // wide, at most 65 chunks of a row per table)
    f.per_row_magic = (1u << 20) / (unsigned)(f.per_row > 0 ? f.per_row : 1) + 1u;
    return f;
}
__device__ inline int cell_slot(const BinFrame &f, uint32_t id) {
    const int row_b = (int)((id >> kCoordBits) & kCoordMask), col_b = (int)(id & kCoordMask);
    const int j = (col_b >> kChunkShift) - f.cc0;
    return (((row_b - f.rmin_b) * f.per_row + (j >> kBinShift)) << kChunkShift) | (col_b & kChunkMask);
}
// inverse of (cell_table, cell_slot): unbiased cell of a slot of a table
__device__ inline void slot_cell(const BinFrame &f, int table, int slot, int &row, int &col) {
    const int kc = slot >> kChunkShift;
    static_assert((kSlots >> kChunkShift) <= 2048 && SSMAP_MAX_MAP_DIM / (8 * kBinParts) + 2 < 200, "range of the reciprocal");
    const int row_rel = (int)(((unsigned)kc * f.per_row_magic) >> 20), kk = kc - row_rel * f.per_row;
    const int row_b = f.rmin_b + row_rel;
    const int j = ((table - f.cc0 - row_b * kRowSkew) & (kBinParts - 1)) + (kk << kBinShift);
    row = row_b - kCoordBias;
    col = (((f.cc0 + j) << kChunkShift) | (slot & kChunkMask)) - kCoordBias;
}

// Place `place` of geometry block b's share of table `part`.  Shares are stored in rows of 64
// records, [place / 64][e][part][b][place % 64]: row 0 of all shares, which holds nearly all
// records, is one dense range of memory, row 1 the next, instead of one short run every
// kGeomPixPerBlock records.
__device__ inline int64_t rec_table_base(const View &v, int e, int part) {
    return ((int64_t)e * kBinParts + part) * v.geom_blocks * 64;
}
__device__ inline int64_t rec_offset(const View &v, int b, unsigned place) {
    const int64_t row_stride = (int64_t)v.E * kBinParts * v.geom_blocks * 64;
    return (int64_t)(place >> 6) * row_stride + (int64_t)(((unsigned)b << 6) | (place & 63u));
}

// One pass over the depth image.  The host predicts a cell box that contains every live pixel of
// the frame (the camera frustum up to depth_max, padded); cell slots are dense over that box, so
// nothing about the frame has to be known before the records are written, and no block waits
// for another:
//   * a block owns kGeomPixPerBlock record places in every winner table; a live pixel takes the
//     next free place of its table (LDS counter) and its 12-byte record is written at once; the
//     block's counts go to bin_cnt;
//   * the depth sentinels are zeroed in place and the owner flags cleared on the way;
//   * the block's lowest invalid pixel and error bits go to its report (GeomPart);
//   * a live pixel outside the predicted box (a depth above depth_max, a box too large to bin)
//     marks the report as spilled and widens its box: the winner kernel then ignores the tables of
//     that frame and scans the pixels instead.
// Block 0 of every env also writes the step's frame header and, when the poses travel as a kernel
// argument (POSE_ARG, at most kPosePack envs), stores the env's pose for the later kernels.
// PACKED: slot and pixel share one record word (frames of at most 2^kPackPixBits pixels).
// QUAD: the image width is a multiple of four, so a thread's four pixels lie in one image row and
// their depths are one aligned 16-byte load; other widths take the four pixels one by one.
template <bool POSE_ARG, bool PACKED, bool QUAD>
__global__ __launch_bounds__(256) void k_geometry(View v, float *__restrict__ depth_all, PosePack pack, int write_target) {
    __shared__ unsigned int s_cnt[kBinParts];
    __shared__ unsigned int s_inv, s_err, s_spill;
    __shared__ int s_box[4];
    __shared__ unsigned long long s_h[kBinParts][64];
    __shared__ uint32_t s_w[kBinParts][64];
    __shared__ uint32_t s_s[PACKED ? 1 : kBinParts][64];
    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    const int64_t blk_base = rec_table_base(v, e, 0) + ((int64_t)blockIdx.x << 6);
    const uint32_t part_stride = (uint32_t)v.geom_blocks * 64u;
    const uint32_t row_stride = (uint32_t)v.E * kBinParts * part_stride;
    const PoseBlock &pb = POSE_ARG ? pack.p[blockIdx.y] : v.poses[e];
    const ssmap_env_maps &mp = v.envs[e].maps;
    if (blockIdx.x == 0) {
        if (POSE_ARG && threadIdx.x < sizeof(PoseBlock) / 8)
            reinterpret_cast<double *>(v.poses + e)[threadIdx.x] = reinterpret_cast<const double *>(&pb)[threadIdx.x];
        if (threadIdx.x == 64) {
            FrameHeader h;
            h.rmin = pb.box[0]; h.rmax = pb.box[1]; h.cmin = pb.box[2]; h.cmax = pb.box[3];
            h.wrmin = h.wcmin = 0x7FFFFFFF;
            h.wrmax = h.wcmax = (int32_t)0x80000000;
            h.min_invalid_pixel = kNoPixel;
            h.n_winners = 0;
            h.status = 0;
            h.found = 0;
            h.any_goal = 0;
            h.any_eroded = 0;
            h.n_neg = 0;
            h.pad = 0;
            v.hdr[e] = h;
            if (!pb.skip) v.overlay_count[v.cur_overlay][e] = 0;   // (a skipped env's overlay list is carried over)
        }
    }
    if (pb.skip) return;   // no frame for this env: nothing is read, no record, no report (whole block)
    const int rmin = pb.box[0], rmax = pb.box[1], cmin = pb.box[2], cmax = pb.box[3];
    const BinFrame f = bin_frame_box(pb.box);
    if (threadIdx.x < kBinParts) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 64) {
        s_inv = kNoPixel; s_err = 0; s_spill = 0;
        s_box[0] = s_box[2] = 0x7FFFFFFF;
        s_box[1] = s_box[3] = (int)0x80000000;
    }
    __syncthreads();

    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned min_inv = kNoPixel, err = 0;
    int srmin = 0x7FFFFFFF, srmax = (int)0x80000000, scmin = 0x7FFFFFFF, scmax = (int)0x80000000;
    bool spill = false;
    if (i0 < v.N) {
        const int r = i0 / v.W;
        const int c0 = i0 - r * v.W;
        float d[4];
        if (QUAD) {
            const float4 d4 = *reinterpret_cast<const float4 *>(depth_all + fbase + i0);
            d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = i0 + k < v.N ? depth_all[fbase + i0 + k] : 0.0f;
        }
        const double yr = (double)r - (double)v.H / 2.0;
        bool dirty = false;
        uint32_t part_slot[4];          // table << 16 | slot, or kDeadCell
        unsigned long long hk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!QUAD && i0 + k >= v.N) { hk[k] = 0; part_slot[k] = kDeadCell; continue; }   // past the frame's end
            if (d[k] == v.depth_min || d[k] == v.depth_max) {   // mapper.py:279-283
                d[k] = 0.0f;
                dirty = true;
                if (!QUAD) depth_all[fbase + i0 + k] = 0.0f;
            }
            int ck = c0 + k;
            double yrk = yr;
            if (!QUAD) {
                const int rk = (i0 + k) / v.W;
                ck = i0 + k - rk * v.W;
                yrk = (double)rk - (double)v.H / 2.0;
            }
            const PixelGeom g = pixel_geometry(v, pb, mp, d[k], ck, yrk);
            hk[k] = g.key;
            part_slot[k] = kDeadCell;
            if (g.index_error) err |= SSMAP_ENV_INDEX_ERROR;
            if (g.invalid) min_inv = min(min_inv, (unsigned)(i0 + k));
            if (g.id == kDeadCell) continue;
            if (f.on && g.row >= rmin && g.row <= rmax && g.col >= cmin && g.col <= cmax) {
                part_slot[k] = ((uint32_t)cell_table(g.id) << 16) | (uint32_t)cell_slot(f, g.id);
            } else {
                spill = true;
                srmin = min(srmin, g.row); srmax = max(srmax, g.row);
                scmin = min(scmin, g.col); scmax = max(scmax, g.col);
            }
        }
        if (QUAD && dirty) *reinterpret_cast<float4 *>(depth_all + fbase + i0) = make_float4(d[0], d[1], d[2], d[3]);
        if (write_target == 2) {   // direct update: no pixel owns a cell yet
            if (QUAD) {
                *reinterpret_cast<int4 *>(v.ocell + fbase + i0) = make_int4(-1, -1, -1, -1);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k < v.N) v.ocell[fbase + i0 + k] = -1;
            }
        } else if (write_target) {   // staged rows: no pixel has a slot in the update list yet
            if (QUAD) {
                *reinterpret_cast<int4 *>(v.target + fbase + i0) = make_int4(-1, -1, -1, -1);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k < v.N) v.target[fbase + i0 + k] = -1;
            }
        }
        // at most kGeomPixPerBlock live pixels in a block: the place is inside the block's share.
        // The four places are asked for together, so that one wait covers the four LDS round trips.
        unsigned place4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            place4[k] = part_slot[k] != kDeadCell ? atomicAdd(&s_cnt[part_slot[k] >> 16], 1u) : 0u;
        // Row 0 of every share (its first 64 places, which hold nearly all records) is staged in LDS
        // and written out after the barrier below, one table row per store instruction: a lane
        // writing its own record is one memory request per record, sixty-four per instruction.
        // Records beyond place 63 go to their rows directly.
        // rec_table_base + rec_offset with the block's part in scalar registers and 32-bit lane
        // arithmetic: a row of places (E * 16 * geom_blocks * 64 of them) is far below 2^32, the
        // sixteen rows of 12-byte records would not fit the GPU's memory otherwise
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (part_slot[k] == kDeadCell) continue;
            const int part = (int)(part_slot[k] >> 16);
            const unsigned slot = part_slot[k] & 0xFFFFu;
            const unsigned place = place4[k];
            const uint32_t w = PACKED ? ((slot << kPackPixBits) | (unsigned)(i0 + k)) : (unsigned)(i0 + k);
            if (place < 64u) {
                s_h[part][place] = hk[k];
                s_w[part][place] = w;
                if (!PACKED) s_s[PACKED ? 0 : part][place] = slot;
            } else {
                const int64_t at = blk_base + (int64_t)((uint64_t)(place >> 6) * row_stride +
                                                        ((uint32_t)part * part_stride + (place & 63u)));
                v.rec_h[at] = hk[k];
                v.rec_w[at] = w;
                if (!PACKED) v.rec_s[at] = slot;
            }
        }
    }
    // the block's report
    const uint32_t w_inv = wave_min_u32(min_inv);
    if ((threadIdx.x & 63) == 0 && w_inv != kNoPixel) atomicMin(&s_inv, w_inv);
    if (__ballot(err != 0) && (threadIdx.x & 63) == 0) s_err = SSMAP_ENV_INDEX_ERROR;
    if (__ballot(spill)) {
        srmin = wave_min(srmin); srmax = wave_max(srmax);
        scmin = wave_min(scmin); scmax = wave_max(scmax);
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_box[0], srmin); atomicMax(&s_box[1], srmax);
            atomicMin(&s_box[2], scmin); atomicMax(&s_box[3], scmax);
            s_spill = 1u;
        }
    }
    __syncthreads();
    {
        // staged rows out: wave w writes the tables w, w + 4, ..., lane <-> place
        const unsigned lane = threadIdx.x & 63u;
        for (int part = (int)(threadIdx.x >> 6); part < kBinParts; part += 4) {
            if (lane < min(s_cnt[part], 64u)) {
                const int64_t at = blk_base + (int64_t)((uint32_t)part * part_stride + lane);
                // (nontemporal like the winner kernel's reads of them)
                __builtin_nontemporal_store(s_h[part][lane], &v.rec_h[at]);
                __builtin_nontemporal_store(s_w[part][lane], &v.rec_w[at]);
                if (!PACKED) __builtin_nontemporal_store(s_s[PACKED ? 0 : part][lane], &v.rec_s[at]);
            }
        }
    }
    if (threadIdx.x < kBinParts)
        v.bin_cnt[((int64_t)e * kBinParts + threadIdx.x) * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
    if (threadIdx.x == 64) {
        GeomPart part;
        part.min_invalid_pixel = s_inv; part.err = s_err; part.spill = s_spill; part.pad = 0;
        part.rmin = s_box[0]; part.rmax = s_box[1]; part.cmin = s_box[2]; part.cmax = s_box[3];
        v.geo_part[(int64_t)e * gridDim.x + blockIdx.x] = part;
    }
}

// One block = one winner table of kSlots cells:
//   phase a   table_h[slot] = max height image            (64-bit integer LDS max)
//   phase b   table_i[slot] = min pixel index among the pixels that reach that max
//   phase c   slots whose winner passes the filters are appended to the update list
// Integer max / min are order independent, so the result is deterministic.  On the binned path the
// block reads the records the geometry kernel left for its table; on the scanning path (spilled
// frames) table p holds cells p * kSlots .. of the frame's cell box numbered row-major, and the
// block walks all pixels of the frame, recomputing their geometry.
// A wave of the winner kernel takes every 16th geometry block, one lane per block for the
// bookkeeping: frames of more than 1024 blocks (a million pixels) take the scanning path.
constexpr int kMaxGeomBlocksBinned = (kWinnerThreads / 64) * 64;
struct WinnerShared {
    unsigned long long table_h[kSlots];
    unsigned int table_i[kSlots];
    unsigned int s_base, s_neg, s_total;
    unsigned short comp[kSlots];           // the table's surviving slots, compacted in slot order
    unsigned int s_cnt[(kSlots / kWinnerThreads) * (kWinnerThreads / 64)];
    int s_box[kWinnerThreads / 64][4];
    GeomPart frame;                        // the geometry blocks' reports, folded
    unsigned int s_tail[kWinnerThreads / 64][64];   // per wave: tail records in front of each of its blocks
};

// Folds the reports of the frame's geometry blocks (every winner block does it for itself).  The
// loads are a step of their own so that the kernel can issue them together with its other
// first loads, ahead of the barriers of the reduction.
struct ReportRegs {
    unsigned min_inv, flags;
    int rmin, rmax, cmin, cmax;
};
__device__ inline ReportRegs load_reports(const View &v, int e) {
    ReportRegs r;
    r.min_inv = kNoPixel; r.flags = 0;
    r.rmin = r.cmin = 0x7FFFFFFF;
    r.rmax = r.cmax = (int)0x80000000;
    for (int b = threadIdx.x; b < v.geom_blocks; b += kWinnerThreads) {
        const uint4 *q = reinterpret_cast<const uint4 *>(v.geo_part + (int64_t)e * v.geom_blocks + b);
        const uint4 a = q[0], c = q[1];
        r.min_inv = min(r.min_inv, a.x);
        r.flags |= a.y | (a.z ? 0x80000000u : 0u);
        r.rmin = min(r.rmin, (int)c.x); r.rmax = max(r.rmax, (int)c.y);
        r.cmin = min(r.cmin, (int)c.z); r.cmax = max(r.cmax, (int)c.w);
    }
    return r;
}
__device__ inline void fold_reports(ReportRegs r, WinnerShared &sh) {
    if (threadIdx.x == 0) {
        sh.frame.min_invalid_pixel = kNoPixel; sh.frame.err = 0; sh.frame.spill = 0; sh.frame.pad = 0;
        sh.frame.rmin = sh.frame.cmin = 0x7FFFFFFF;
        sh.frame.rmax = sh.frame.cmax = (int)0x80000000;
    }
    __syncthreads();
    const unsigned min_inv = wave_min_u32(r.min_inv);
    unsigned flags = r.flags;
    const bool any_flag = __ballot(flags != 0) != 0ull;
    if ((threadIdx.x & 63) == 0 && min_inv != kNoPixel) atomicMin(&sh.frame.min_invalid_pixel, min_inv);
    if (any_flag) {   // rare: index errors or spilled pixels
        for (int o = 32; o > 0; o >>= 1) flags |= (unsigned)__shfl_xor((int)flags, o);
        const int rmin = wave_min(r.rmin), rmax = wave_max(r.rmax), cmin = wave_min(r.cmin), cmax = wave_max(r.cmax);
        if ((threadIdx.x & 63) == 0) {
            if (flags & 0x7FFFFFFFu) atomicOr(&sh.frame.err, flags & 0x7FFFFFFFu);
            if (flags & 0x80000000u) sh.frame.spill = 1u;
            if (rmin <= rmax) {
                atomicMin(&sh.frame.rmin, rmin); atomicMax(&sh.frame.rmax, rmax);
                atomicMin(&sh.frame.cmin, cmin); atomicMax(&sh.frame.cmax, cmax);
            }
        }
    }
    __syncthreads();
}

// What a winner block reads about its frame before anything else, issued as one group of loads.
struct WinnerPre {
    int rmin, rmax, cmin, cmax;      // predicted box (frame header)
    int Hm, Wm;
    double lower_x, lower_z;
    double cam_x, cam_y, cam_z, h_cut, agent_y;
    unsigned cnt;                    // bin_cnt of geometry block wave + 16 lane for table blockIdx.x
    gptr<float> height_map;          // direct update: the winner kernel raises the height map and
    gptr<uint32_t> gt_map;           // writes the ground-truth map itself (mapper.py:338-341, 361-363)
};

// The word the streaming kernel finds for a pixel that owns a map cell: the cell, and what the
// occupancy override (mapper.py:556-569) will ask of the cell's new height.
__device__ inline int owner_word(int cell, float hnew) {
    return cell | (hnew > 0.1f ? (int)kCellHighBit : 0) | ((hnew != -INFINITY && hnew < 0.1f) ? (int)kCellLowBit : 0);
}

// One binned record in registers.  The not-updatable flag is not stored: a pixel in a table has
// valid depth and lies inside the upper bounds, so it is updatable exactly when its height is
// below the cut-off, which the table's winning height tells.
template <bool PACKED>
struct Rec;
template <>
struct Rec<true> {
    unsigned long long h;
    uint32_t w;
    __device__ uint32_t slot() const { return w >> kPackPixBits; }
    __device__ uint32_t word() const { return (w & ((1u << kPackPixBits) - 1u)) << 1; }   // pixel << 1
};
template <>
struct Rec<false> {
    unsigned long long h;
    uint32_t w, s;
    __device__ uint32_t slot() const { return s; }
    __device__ uint32_t word() const { return w << 1; }
};
template <bool PACKED>
__device__ inline Rec<PACKED> load_rec(const View &v, int64_t at);
template <>
__device__ inline Rec<true> load_rec<true>(const View &v, int64_t at) {
    Rec<true> r;
    r.h = v.rec_h[at];
    r.w = v.rec_w[at];
    return r;
}
template <>
__device__ inline Rec<false> load_rec<false>(const View &v, int64_t at) {
    Rec<false> r;
    r.h = v.rec_h[at];
    r.w = v.rec_w[at];
    r.s = v.rec_s[at];
    return r;
}

template <bool RUNS, bool PACKED>
__device__ __forceinline__ void winners_body(const View &v, const float *__restrict__ depth_all, int write_target,
                                             int write_list, WinnerShared &sh, const WinnerPre &pre,
                                             const int32_t *__restrict__ gt_all, bool commit_now, bool want_box) {
    unsigned long long (&table_h)[kSlots] = sh.table_h;
    unsigned int (&table_i)[kSlots] = sh.table_i;
    unsigned int &s_base = sh.s_base, &s_neg = sh.s_neg;
    unsigned int (&s_cnt)[(kSlots / kWinnerThreads) * (kWinnerThreads / 64)] = sh.s_cnt;
    int (&s_box)[kWinnerThreads / 64][4] = sh.s_box;

    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    FrameHeader *hdr = v.hdr + e;
    const GeomPart *acc = &sh.frame;
    int rmin = pre.rmin, rmax = pre.rmax, cmin = pre.cmin, cmax = pre.cmax;
    if (!RUNS) {
        // scanned frame: the cells lie in the box of the spilled pixels, joined with the predicted
        // box when that one was in use (its pixels were not accumulated)
        const int32_t pb4[4] = {pre.rmin, pre.rmax, pre.cmin, pre.cmax};
        const bool pred = bin_frame_box(pb4).on;
        if (!pred) { rmin = cmin = 0x7FFFFFFF; rmax = cmax = (int)0x80000000; }
        rmin = min(rmin, acc->rmin); rmax = max(rmax, acc->rmax);
        cmin = min(cmin, acc->cmin); cmax = max(cmax, acc->cmax);
    }
    if (rmin > rmax || cmin > cmax) return;
    const ssmap_env_maps &mp = v.envs[e].maps;
    const PoseBlock &pb = v.poses[e];
    const int Hm = pre.Hm, Wm = pre.Wm;
    const int bw = cmax - cmin + 1;
    const long long ncell = (long long)bw * (long long)(rmax - rmin + 1);
    const int nfull = (int)((ncell + kSlots - 1) >> kSlotShift);
    const int rmin_b = rmin + kCoordBias, cmin_b = cmin + kCoordBias;
    const int nparts = RUNS ? kBinParts : nfull;

    // the cell all invalid-depth pixels fall into (they sit at the camera position,
    // mapper.py:28-31) and their common height
    const unsigned min_inv = acc->min_invalid_pixel;
    long long inv_L = -1;
    unsigned long long inv_key = 0;
    if (min_inv != kNoPixel) {
        const double fr = floor(((0.0 + pre.cam_z) - pre.lower_z) / v.p2m);
        const double fc = floor(((0.0 + pre.cam_x) - pre.lower_x) / v.p2m);
        if (fabs(fr) < 1.0e9 && fabs(fc) < 1.0e9) {
            const int row = (int)fr, col = (int)fc;
            if (row >= rmin && row <= rmax && col >= cmin && col <= cmax) {
                inv_L = (long long)(row - rmin) * bw + (col - cmin);
                inv_key = order_f64(0.0 + pre.cam_y);
            }
        }
    }

    // the slot layout of the binned tables, from the header's box (already in registers)
    const int32_t pre_box[4] = {pre.rmin, pre.rmax, pre.cmin, pre.cmax};
    const BinFrame bf = bin_frame_box(pre_box);
    int inv_part = -1, inv_slot = 0;
    if (inv_L >= 0) {
        if (RUNS) {
            const uint32_t id = ((uint32_t)((int)(inv_L / bw) + rmin_b) << kCoordBits) | (uint32_t)((int)(inv_L % bw) + cmin_b);
            inv_part = cell_table(id);
            inv_slot = cell_slot(bf, id);
        } else {
            inv_part = (int)(inv_L >> kSlotShift);
            inv_slot = (int)(inv_L & (kSlots - 1));
        }
    }
    const float *depth_env = depth_all + fbase;   // sentinels already zeroed
    constexpr int kWaves = kWinnerThreads / 64;
    constexpr int kKeep = RUNS ? (PACKED ? 19 : 14) : 1;   // records per thread kept in registers between the phases
    Rec<PACKED> keep[kKeep], tail0;
    int lrmin = 0x7FFFFFFF, lrmax = (int)0x80000000, lcmin = 0x7FFFFFFF, lcmax = (int)0x80000000;
    if (threadIdx.x == 0) s_neg = 0;

    // binned frames: one table per block, a single trip (the launch has kBinParts blocks per env);
    // scanned frames may have more tables than blocks
    static_assert(kWinnerPartsPerEnv == kBinParts, "one block per table on the binned path");
    for (int p = blockIdx.x; p < nparts; p += RUNS ? kBinParts : (int)gridDim.x) {
        // Binned records of this table: geometry block b left bin_cnt[b] of them in its share, in
        // rows of 64.  Wave w takes the blocks w, w + 16, ...; lane l keeps the books of the l-th
        // of them.  Row 0 of the wave's first kKeep blocks is read one lane per record and stays in
        // registers between the phases; everything else (the few records beyond 64 of those
        // blocks, later blocks) forms the wave's tail, numbered through a prefix sum.
        const int nb = v.geom_blocks;
        const int64_t tbase = rec_table_base(v, e, p);
        const int wave_id = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
        unsigned cvec = 0, tail_total = 0;
        if (RUNS) {
            const unsigned int *cnt = v.bin_cnt + ((int64_t)e * kBinParts + p) * nb;
            const int b = wave_id + kWaves * lane_id;
            const unsigned raw = p == (int)blockIdx.x ? pre.cnt : (b < nb ? cnt[b] : 0u);   // first table: loaded up front
            cvec = min(raw, (unsigned)kGeomPixPerBlock);
            const unsigned tn = lane_id < kKeep ? (cvec > 64u ? cvec - 64u : 0u) : cvec;
            unsigned incl = tn;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o);
                if (lane_id >= o) incl += up;
            }
            sh.s_tail[wave_id][lane_id] = incl - tn;
            tail_total = __shfl(incl, 63);
        }
        // place of tail record i of this wave (i < tail_total); read after the barrier below
        auto tail_offset = [&](unsigned i) -> int64_t {
            int lo = 0, hi = 64;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (i >= sh.s_tail[wave_id][mid]) lo = mid; else hi = mid;
            }
            return rec_offset(v, wave_id + kWaves * lo, (lo < kKeep ? 64u : 0u) + (i - sh.s_tail[wave_id][lo]));
        };
        // (the tables are clear here: the kernel clears them first thing, and an iteration that
        // is followed by another clears them again at its end)
        __syncthreads();
        const bool own_inv = inv_L >= 0 && inv_part == p;
        if (own_inv && threadIdx.x == 0) table_h[inv_slot] = inv_key;
        __syncthreads();

        SSMAP_STAMP(3);
        // phase a
        if (RUNS) {
            {
                // row 0 of block wave_id + 16 u: one scalar base per u, one lane offset for all
                const unsigned lane_off = ((unsigned)wave_id << 6) | (unsigned)lane_id;
                const unsigned long long *hrow = v.rec_h + tbase;
                const uint32_t *wrow = v.rec_w + tbase;
                const uint32_t *srow = PACKED ? nullptr : v.rec_s + tbase;
#pragma unroll
                for (int u = 0; u < kKeep; ++u) {
                    const unsigned c = __builtin_amdgcn_readlane(cvec, u);
                    if ((unsigned)lane_id < c) {   // the block's first 64 records
                        // records are written once and read once: nontemporal on both sides keeps
                        // them from displacing what the streaming kernel will find in the caches
                        // (the winner kernel itself reads 1.6 us slower for it, the streaming kernel
                        // 3.5 us faster at 40 classes, 13 us at 150)
                        keep[u].h = __builtin_nontemporal_load(&hrow[lane_off + (unsigned)(u * kWaves * 64)]);
                        keep[u].w = __builtin_nontemporal_load(&wrow[lane_off + (unsigned)(u * kWaves * 64)]);
                        if constexpr (!PACKED) keep[u].s = __builtin_nontemporal_load(&srow[lane_off + (unsigned)(u * kWaves * 64)]);
                    }
                }
            }
            if ((unsigned)lane_id < tail_total) tail0 = load_rec<PACKED>(v, tbase + tail_offset(lane_id));
#pragma unroll
            for (int u = 0; u < kKeep; ++u)
                if ((unsigned)lane_id < (unsigned)__builtin_amdgcn_readlane(cvec, u))
                    atomicMax(&table_h[keep[u].slot()], keep[u].h);
            if ((unsigned)lane_id < tail_total) atomicMax(&table_h[tail0.slot()], tail0.h);
            for (unsigned i = 64u + lane_id; i < tail_total; i += 64u) {
                const Rec<PACKED> rec = load_rec<PACKED>(v, tbase + tail_offset(i));
                atomicMax(&table_h[rec.slot()], rec.h);
            }
        } else {
        // scanned frame: the geometry is computed again for every table pass (rare, and exact
        // by construction: the same function, the same inputs)
        for (int i = threadIdx.x; i < v.N; i += kWinnerThreads) {
            const int r = i / v.W, c = i - r * v.W;
            const PixelGeom g = pixel_geometry(v, pb, mp, depth_env[i], c, (double)r - (double)v.H / 2.0);
            if (g.id == kDeadCell) continue;
            const int L = (g.row - rmin) * bw + (g.col - cmin);
            if ((L >> kSlotShift) == p) atomicMax(&table_h[L & (kSlots - 1)], g.key);
        }
        }
        __syncthreads();
        SSMAP_STAMP(4);
        // phase b
        if (own_inv && threadIdx.x == 0) {
            const int s = inv_slot;
            if (table_h[s] == inv_key) table_i[s] = (min_inv << 1) | 1u;
        }
        __syncthreads();
        if (RUNS) {
#pragma unroll
            for (int u = 0; u < kKeep; ++u)
                if ((unsigned)lane_id < (unsigned)__builtin_amdgcn_readlane(cvec, u) &&
                    table_h[keep[u].slot()] == keep[u].h)
                    atomicMin(&table_i[keep[u].slot()], keep[u].word());
            if ((unsigned)lane_id < tail_total && table_h[tail0.slot()] == tail0.h)
                atomicMin(&table_i[tail0.slot()], tail0.word());
            for (unsigned i = 64u + lane_id; i < tail_total; i += 64u) {
                const Rec<PACKED> rec = load_rec<PACKED>(v, tbase + tail_offset(i));
                if (table_h[rec.slot()] == rec.h) atomicMin(&table_i[rec.slot()], rec.word());
            }
        } else {
        for (int i = threadIdx.x; i < v.N; i += kWinnerThreads) {
            const int r = i / v.W, c = i - r * v.W;
            const PixelGeom g = pixel_geometry(v, pb, mp, depth_env[i], c, (double)r - (double)v.H / 2.0);
            if (g.id == kDeadCell) continue;
            const int L = (g.row - rmin) * bw + (g.col - cmin);
            if ((L >> kSlotShift) == p) {
                const int s = L & (kSlots - 1);
                if (g.key == table_h[s]) atomicMin(&table_i[s], (unsigned)i << 1);
            }
        }
        }
        __syncthreads();
        SSMAP_STAMP(5);
        // phase c: emit the surviving slots in slot order (= row-major cell order), so that
        // neighbouring list entries are neighbouring map cells.  The surviving slots (an eighth of the
        // table on a typical frame) are first compacted in LDS, so that the emission proper runs with
        // every lane busy, one list entry per thread and trip.
        // Direct update: height map and ground-truth map of the emitted cells are written here too,
        // in that order (slots of a table are runs of eight neighbouring cells), and the owners' words
        // carry what the occupancy override will ask of the new height: the streaming kernel then
        // meets no per-cell scalar but the label.
        constexpr int kIters = kSlots / kWinnerThreads;
        const int lane = lane_id, wave = wave_id;
        const bool commit = write_target == 2 && commit_now;
        unsigned long long balls[kIters];
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const unsigned w = table_i[it * kWinnerThreads + threadIdx.x];
            // updatable: not the invalid-depth entry, and below the height cut-off
            balls[it] = __ballot(w != kNoPixel && !(w & 1u) &&
                                 unorder_f64(table_h[it * kWinnerThreads + threadIdx.x]) < pre.h_cut);
            if (lane == 0) s_cnt[it * kWaves + wave] = (unsigned)__popcll(balls[it]);
        }
        __syncthreads();
        // exclusive prefix sums of the kIters * kWaves ballot counts, by the first wave
        static_assert(kIters * kWaves == 128, "two counts per lane");
        if (wave == 0) {
            const unsigned c0 = s_cnt[2 * lane], c1 = s_cnt[2 * lane + 1];
            unsigned incl = c0 + c1;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            s_cnt[2 * lane] = incl - c0 - c1;
            s_cnt[2 * lane + 1] = incl - c1;
            if (lane == 63) {
                sh.s_total = incl;
                s_base = (incl && write_list) ? atomicAdd(&hdr->n_winners, incl) : 0u;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kIters; ++it)
            if ((balls[it] >> lane) & 1ull)
                sh.comp[s_cnt[it * kWaves + wave] + (unsigned)__popcll(balls[it] & ((1ull << lane) - 1ull))] =
                    (unsigned short)(it * kWinnerThreads + threadIdx.x);
        __syncthreads();
        SSMAP_STAMP(6);
        unsigned n_neg = 0;
        const unsigned total = write_list ? sh.s_total : 0u;
        // Two list entries per thread and trip (the tables of a typical frame hold a few more than
        // kWinnerThreads survivors each): the loads of both entries are issued before the first store,
        // so a trip costs one round trip to memory, not one per entry.  A lane without a second entry
        // repeats its first one's addresses for the loads and stores nothing for it.
        constexpr int kPer = 2;
        for (unsigned j0 = threadIdx.x; j0 < total; j0 += kPer * kWinnerThreads) {
            bool on[kPer];
            int cellv[kPer], pixv[kPer], gv[kPer];
            uint32_t wordv[kPer];
            unsigned long long hkv[kPer];
            float ho[kPer];
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const unsigned j = j0 + (unsigned)(k * kWinnerThreads);
                on[k] = j < total;
                const int s = (int)sh.comp[on[k] ? j : j0];
                pixv[k] = (int)(table_i[s] >> 1);
                hkv[k] = table_h[s];
                int row, col;
                if (RUNS) {
                    slot_cell(bf, p, s, row, col);
                } else {
                    // cell numbers fit 32 bits (a box holds at most 4 * 8192^2 = 2^28 cells)
                    const unsigned L = ((unsigned)p << kSlotShift) + (unsigned)s;
                    const unsigned lrow = L / (unsigned)bw;
                    row = (int)lrow + rmin;
                    col = (int)(L - lrow * (unsigned)bw) + cmin;
                }
                if (want_box && on[k]) {
                    lrmin = min(lrmin, row); lrmax = max(lrmax, row);
                    lcmin = min(lcmin, col); lcmax = max(lcmax, col);
                }
                // numpy index wrap for -dim <= idx < 0 (mapper.py:326-338 masks only the upper bound)
                const int wr = row < 0 ? row + Hm : row, wc = col < 0 ? col + Wm : col;
                cellv[k] = wr * Wm + wc;
                n_neg += (on[k] && (row < 0 || col < 0)) ? 1u : 0u;
                wordv[k] = (uint32_t)pixv[k] | (row < 0 ? kRowNegBit : 0u) | (col < 0 ? kColNegBit : 0u);
                ho[k] = 0.0f;
                gv[k] = 0;
                if (commit) {
                    ho[k] = pre.height_map[cellv[k]];
                    if (gt_all) gv[k] = gt_all[fbase + pixv[k]];
                }
            }
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                if (!on[k]) continue;
                const unsigned pos = s_base + j0 + (unsigned)(k * kWinnerThreads);
                const int cell = cellv[k], pix = pixv[k];
                int2 *out = reinterpret_cast<int2 *>(v.winners) + fbase + pos;
                if (write_target) {
                    // probability rules: the streaming kernel runs before anything reads the list
                    // again (nontemporal: 1 us per step); the label rules read it next (plain)
                    typedef int int2v __attribute__((ext_vector_type(2)));
                    int2v ev; ev.x = (int)wordv[k]; ev.y = cell;
                    __builtin_nontemporal_store(ev, reinterpret_cast<int2v *>(out));
                    __builtin_nontemporal_store(hkv[k], &v.wheight[fbase + pos]);
                } else {
                    *out = make_int2((int)wordv[k], cell);
                    v.wheight[fbase + pos] = hkv[k];
                }
                if (commit) {
                    const float hn = fmaxf(ho[k], (float)(unorder_f64(hkv[k]) - pre.agent_y));
                    // (neither map is read again in this step: nontemporal, 1.3 us per step)
                    __builtin_nontemporal_store(hn, &pre.height_map[cell]);
                    if (gt_all) __builtin_nontemporal_store((uint32_t)gv[k], &pre.gt_map[cell]);
                    v.ocell[fbase + pix] = owner_word(cell, hn);
                } else if (write_target == 2) {
                    // a frame whose list may still change (wrapped cells, an IndexError candidate): the
                    // frame's last block writes the maps and the height bits once the list stands
                    v.ocell[fbase + pix] = cell;
                } else if (write_target) {
                    v.target[fbase + pix] = (int)pos;
                }
            }
        }
        if (n_neg) atomicAdd(&s_neg, n_neg);
        __syncthreads();
        SSMAP_STAMP(7);
        if (!RUNS && p + (int)gridDim.x < nparts)
            for (int s = threadIdx.x; s < kSlots; s += kWinnerThreads) { table_h[s] = 0ull; table_i[s] = kNoPixel; }
    }

    __syncthreads();
    if (threadIdx.x == 0 && s_neg) atomicAdd(&hdr->n_neg, s_neg);
    if (!want_box) return;
    // bbox of the update cells (goal_decay works on it, mapper.py:1117-1118): one set of header
    // atomics per block (same-line atomics serialise at the memory side)
    lrmin = wave_min(lrmin); lrmax = wave_max(lrmax);
    lcmin = wave_min(lcmin); lcmax = wave_max(lcmax);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s_box[w][0] = lrmin; s_box[w][1] = lrmax; s_box[w][2] = lcmin; s_box[w][3] = lcmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWinnerThreads / 64; ++w) {
            lrmin = min(lrmin, s_box[w][0]); lrmax = max(lrmax, s_box[w][1]);
            lcmin = min(lcmin, s_box[w][2]); lcmax = max(lcmax, s_box[w][3]);
        }
        if (lrmin <= lrmax) {
            atomicMin(&hdr->wrmin, lrmin); atomicMax(&hdr->wrmax, lrmax);
            atomicMin(&hdr->wcmin, lcmin); atomicMax(&hdr->wcmax, lcmax);
        }
    }
}

// numpy keeps only the LAST write when one fancy-index assignment names a cell twice, which
// happens when an update cell with a negative index wraps onto another update cell
// (mapper.py:326-338 masks only the upper bounds).  At most four update pixels can reach one
// wrapped cell -- one per sign quadrant of (row, col) -- so frames with wrapped cells record
// each update pixel in its quadrant slot of the cell; the highest pixel index owns the cell
// and the others are flagged.  One block per env (the winner block of the env that finishes
// last): slots are written, then compared.
__device__ inline void resolve_wrapped(const View &v, int e, unsigned n, int write_target) {
    const int64_t fbase = (int64_t)e * v.N;
    int2 *list = reinterpret_cast<int2 *>(v.winners) + fbase;
    int32_t *quad = v.quad + e * v.quad_stride;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        const int2 pc = list[i];
        const uint32_t pw = (uint32_t)pc.x;
        const int q = ((pw & kRowNegBit) ? 2 : 0) | ((pw & kColNegBit) ? 1 : 0);
        quad[4ll * pc.y + q] = (int)(pw & kPixMask) + 1;
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        const int2 pc = list[i];
        const int pix = (int)((uint32_t)pc.x & kPixMask);
        const int4 qs = *reinterpret_cast<const int4 *>(quad + 4ll * pc.y);
        const int owner = max(max(qs.x, qs.y), max(qs.z, qs.w));
        if (owner != pix + 1) {
            list[i].x = (int)((uint32_t)pc.x | kLostBit);
            if (write_target == 2) v.ocell[fbase + pix] = -1;
            else if (write_target) v.target[fbase + pix] = -1;
        }
    }
}

// Direct update, frames whose update list was not final when it was emitted: height map,
// ground-truth map and the owners' words for every entry that kept its cell (one block).
__device__ inline void commit_heights(const View &v, int e, unsigned n, const int32_t *__restrict__ gt_all,
                                      gptr<float> height_map, gptr<uint32_t> gt_map, double agent_y) {
    const int64_t fbase = (int64_t)e * v.N;
    const int2 *list = reinterpret_cast<const int2 *>(v.winners) + fbase;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        const int2 pc = list[i];
        if ((uint32_t)pc.x & kLostBit) continue;
        const int pix = (int)((uint32_t)pc.x & kPixMask);
        const float hn = fmaxf(height_map[pc.y], (float)(unorder_f64(v.wheight[fbase + i]) - agent_y));
        height_map[pc.y] = hn;
        if (gt_all) gt_map[pc.y] = (uint32_t)gt_all[fbase + pix];
        v.ocell[fbase + pix] = owner_word(pc.y, hn);
    }
}

// numpy raises IndexError for a frame only if a SELECTED pixel has an index below -dim
// (mapper.py:331-338): the tallest pixel of its cell, over all pixels of the frame (ties to the
// lowest pixel), that also passes the height and depth tests.  The geometry kernel reports a
// candidate: some passing pixel lies in such a cell.  The candidate stands unless every such cell
// holds something taller that does not pass: a pixel at or above the height cut-off (it beats every
// passing pixel), or, in the cell of the camera, the invalid-depth pixels (they sit at the camera
// position, mapper.py:28-31) when none of the cell's passing pixels beats them.  One block (the
// frame's last winner block) decides, in two sweeps over the frame with the exact geometry:
//   A  the below-range cells of the pixels at or above the cut-off go into an LDS hash set
//      (coordinates as 32-bit codes, INT64_MIN as its own code);
//   B  a passing pixel in a below-range cell that is not in the set (and, in the camera's cell, beats
//      the invalid-depth pixels) confirms the error.
// Heights that are NaN never win a cell.  What cannot be decided exactly (a coordinate beyond 32
// bits that is not INT64_MIN, a set that does not fit) confirms the error.
constexpr uint32_t kCodeMin = 0x80000000u;   // numpy's cast of a non-finite / out-of-range coordinate
__device__ inline bool coord_code(double f, uint32_t &code) {
    if (!(f < 9223372036854775808.0) || f < -9223372036854775808.0) { code = kCodeMin; return true; }
    if (!(fabs(f) < 2147483647.0)) return false;
    code = (uint32_t)(int32_t)f;
    return true;
}
__device__ inline bool index_error_confirmed(const View &v, const float *__restrict__ depth_env, int e,
                                             WinnerShared &sh) {
    constexpr unsigned long long kEmpty = ~0ull;   // (-1, -1) is no below-range cell
    constexpr int kProbes = 128;
    __shared__ unsigned int s_verdict, s_overflow;
    // (everything is read again from memory here: nothing of this rare path is kept in registers
    // across the winner selection)
    const ssmap_env_maps &mp = v.envs[e].maps;
    const PoseBlock &pb = v.poses[e];
    const int Hm = mp.map_h, Wm = mp.map_w;
    for (int s = threadIdx.x; s < kSlots; s += kWinnerThreads) sh.table_h[s] = kEmpty;
    if (threadIdx.x == 0) { s_verdict = 0; s_overflow = 0; }
    __syncthreads();
    auto slot_of = [](unsigned long long key) {
        key *= 0x9E3779B97F4A7C15ull;
        return (unsigned)(key >> 40) & (unsigned)(kSlots - 1);
    };
    // sweep A
    for (int i = threadIdx.x; i < v.N; i += kWinnerThreads) {
        const int r = i / v.W, c = i - r * v.W;
        bool unused = true, ok, below_upper, above_lower;
        double wy, fr, fc;
        pixel_world<false>(v, pb, mp, depth_env[i], c, (double)r - (double)v.H / 2.0, unused, ok, wy, fr, fc);
        cell_masks(fr, fc, Hm, Wm, below_upper, above_lower);
        if (!ok || !below_upper || above_lower || !(wy == wy) || wy < pb.h_cut) continue;
        uint32_t cr, cc;
        if (!coord_code(fr, cr) || !coord_code(fc, cc)) continue;   // (its cell's candidates confirm the error in sweep B)
        const unsigned long long key = ((unsigned long long)cr << 32) | cc;
        unsigned s = slot_of(key);
        bool placed = false;
        for (int k = 0; k < kProbes && !placed; ++k) {
            const unsigned long long old = atomicCAS(&sh.table_h[s], kEmpty, key);
            placed = old == kEmpty || old == key;
            s = (s + 1) & (unsigned)(kSlots - 1);
        }
        if (!placed) s_overflow = 1u;
    }
    __syncthreads();
    if (s_overflow) return true;
    // the camera's cell, where the invalid-depth pixels sit
    bool has_cam = false;
    unsigned long long cam_key = 0;
    const unsigned min_inv = sh.frame.min_invalid_pixel;
    if (min_inv != kNoPixel) {
        const double fr = floor(((0.0 + pb.trans[2]) - mp.lower[2]) / v.p2m);
        const double fc = floor(((0.0 + pb.trans[0]) - mp.lower[0]) / v.p2m);
        bool below_upper, above_lower;
        cell_masks(fr, fc, Hm, Wm, below_upper, above_lower);
        uint32_t cr, cc;
        if (below_upper && !above_lower && coord_code(fr, cr) && coord_code(fc, cc)) {
            has_cam = true;
            cam_key = ((unsigned long long)cr << 32) | cc;
        }
    }
    const double cam_h = 0.0 + pb.trans[1];
    // sweep B
    for (int i = threadIdx.x; i < v.N; i += kWinnerThreads) {
        const int r = i / v.W, c = i - r * v.W;
        bool unused = true, ok, below_upper, above_lower;
        double wy, fr, fc;
        pixel_world<false>(v, pb, mp, depth_env[i], c, (double)r - (double)v.H / 2.0, unused, ok, wy, fr, fc);
        cell_masks(fr, fc, Hm, Wm, below_upper, above_lower);
        if (!ok || !below_upper || above_lower || !(wy < pb.h_cut)) continue;   // not a candidate
        uint32_t cr, cc;
        bool confirmed = true;
        if (coord_code(fr, cr) && coord_code(fc, cc)) {
            const unsigned long long key = ((unsigned long long)cr << 32) | cc;
            unsigned s = slot_of(key);
            bool in_set = false;
            for (int k = 0; k < kProbes; ++k) {
                const unsigned long long at = sh.table_h[s];
                if (at == key) { in_set = true; break; }
                if (at == kEmpty) break;
                s = (s + 1) & (unsigned)(kSlots - 1);
            }
            if (in_set) confirmed = false;
            else if (has_cam && key == cam_key) confirmed = wy > cam_h || (wy == cam_h && (unsigned)i < min_inv);
        }
        if (confirmed) s_verdict = 1u;
    }
    __syncthreads();
    return s_verdict != 0u;
}

// A frame that raises writes nothing (the reference raises at its first indexed statement,
// mapper.py:338, before any map is touched): the update list the winner blocks have emitted is
// withdrawn, so the rule, hit/view and found kernels leave the env as it is.
__device__ inline void withdraw_update_list(const View &v, int e, unsigned n, int write_target) {
    const int64_t fbase = (int64_t)e * v.N;
    const int2 *list = reinterpret_cast<const int2 *>(v.winners) + fbase;
    if (write_target)
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
            const int pix = (int)((uint32_t)list[i].x & kPixMask);
            if (write_target == 2) v.ocell[fbase + pix] = -1;
            else v.target[fbase + pix] = -1;
        }
    if (threadIdx.x == 0) {
        FrameHeader *hdr = v.hdr + e;
        hdr->n_winners = 0;
        hdr->n_neg = 0;
        hdr->wrmin = hdr->wcmin = 0x7FFFFFFF;
        hdr->wrmax = hdr->wcmax = (int32_t)0x80000000;
        atomicOr(&hdr->status, (uint32_t)SSMAP_ENV_INDEX_ERROR);
    }
}

// hv rules rebuild semantic_map from background_map every frame (mapper.py:662, 723); the engine keeps
// semantic_map == background_map except on the cells that showed the goal overlay last frame, and
// undoes that list here, ahead of this frame's rule kernel, instead of copying the whole map.  The
// list has one entry per slot of last frame's update list: the cell if it showed the overlay, -1 if
// not.  Entries first, first + stride, ... (a slice per winner block, or the whole list for one block).
__device__ inline void hv_revert_overlay(const View &v, int e, unsigned first, unsigned stride) {
    const int prev = v.cur_overlay ^ 1;
    const unsigned n = v.overlay_count[prev][e];
    const int32_t *list = v.overlay[prev] + (int64_t)e * v.N;
    const ssmap_env_maps &mp = v.envs[e].maps;
    const gptr<uint32_t> sem = as_global(mp.semantic_map);
    const gptr<const uint32_t> bg = as_global(mp.background_map);
    for (unsigned i = first; i < n; i += stride) {
        const int cell = list[i];
        if (cell >= 0) sem[cell] = bg[cell];
    }
}
// An env without a frame in this step, or whose frame raised IndexError, keeps last frame's overlay
// on its map: the list moves on to the buffer the next step will undo.
__device__ inline void hv_keep_overlay(const View &v, int e, unsigned first, unsigned stride, bool write_count) {
    const int prev = v.cur_overlay ^ 1;
    const unsigned n = v.overlay_count[prev][e];
    const int32_t *list = v.overlay[prev] + (int64_t)e * v.N;
    int32_t *keep = v.overlay[v.cur_overlay] + (int64_t)e * v.N;
    for (unsigned i = first; i < n; i += stride) keep[i] = list[i];
    if (write_count) v.overlay_count[v.cur_overlay][e] = n;
}

// Frames whose predicted cell box fits the 16 tables and holds every live pixel are read as
// per-table record ranges; the others (spilled) are scanned pixel by pixel, once per table.
// Only a frame whose cell box reaches below zero, or a spilled one, can hold wrapped cells: the
// winner blocks of such a frame publish their part of the update list (device-scope release), take
// a ticket, and the block that takes the last one resolves the wrapped cells of the whole list.
// Every block of a frame sees the same box and reports, so they agree on whether tickets are taken.
__global__ __launch_bounds__(kWinnerThreads) void k_winners(View v, const float *__restrict__ depth_all,
                                                            const int32_t *__restrict__ gt_all,
                                                            int write_target, int write_list, int hv_rule, int want_box) {
    __shared__ WinnerShared sh;
    __shared__ unsigned int s_last;
    const int e = blockIdx.y;
    FrameHeader *hdr = v.hdr + e;
    const unsigned hv_first = blockIdx.x * kWinnerThreads + threadIdx.x, hv_stride = gridDim.x * kWinnerThreads;
    SSMAP_STAMP(0);
    // every first load of the block in one group, the env's skipped flag among them: the latencies
    // overlap instead of adding up (a skipped env's reports are stale, and unused)
    const int skip = v.poses[e].skip;
    const ReportRegs rep = load_reports(v, e);
    WinnerPre pre;
    {
        const ssmap_env_maps &mp = v.envs[e].maps;
        const PoseBlock &pb = v.poses[e];
        pre.rmin = hdr->rmin; pre.rmax = hdr->rmax; pre.cmin = hdr->cmin; pre.cmax = hdr->cmax;
        pre.Hm = mp.map_h; pre.Wm = mp.map_w;
        pre.lower_x = mp.lower[0]; pre.lower_z = mp.lower[2];
        pre.cam_x = pb.trans[0]; pre.cam_y = pb.trans[1]; pre.cam_z = pb.trans[2];
        pre.h_cut = pb.h_cut;
        pre.agent_y = pb.agent[1];
        pre.height_map = as_global(mp.height_map);
        pre.gt_map = as_global(mp.gt_map);
        const int b = (int)(threadIdx.x >> 6) + (kWinnerThreads / 64) * (int)(threadIdx.x & 63);
        pre.cnt = (v.geom_blocks <= kMaxGeomBlocksBinned && b < v.geom_blocks)
                      ? v.bin_cnt[((int64_t)e * kMaxParts + blockIdx.x) * v.geom_blocks + b] : 0u;
    }
    if (skip) {   // no frame for this env (its header says zero update cells)
        if (hv_rule) hv_keep_overlay(v, e, hv_first, hv_stride, blockIdx.x == 0 && threadIdx.x == 0);
        return;
    }
    // clearing the tables overlaps with those loads
    for (int s = threadIdx.x; s < kSlots; s += kWinnerThreads) { sh.table_h[s] = 0ull; sh.table_i[s] = kNoPixel; }
    SSMAP_STAMP(1);
    fold_reports(rep, sh);
    SSMAP_STAMP(2);
    // an index-error report of the geometry kernel is a candidate: the frame's last block decides
    const bool err_candidate = (sh.frame.err & SSMAP_ENV_INDEX_ERROR) != 0u;
    // (a frame that may still raise leaves last frame's overlay to its last block, which knows)
    if (hv_rule && !err_candidate) hv_revert_overlay(v, e, hv_first, hv_stride);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr->min_invalid_pixel = sh.frame.min_invalid_pixel;
        if (sh.frame.err & ~(uint32_t)SSMAP_ENV_INDEX_ERROR) atomicOr(&hdr->status, sh.frame.err & ~(uint32_t)SSMAP_ENV_INDEX_ERROR);
    }
    const int32_t box[4] = {pre.rmin, pre.rmax, pre.cmin, pre.cmax};
    const bool may_wrap = write_list && (sh.frame.spill || pre.rmin < 0 || pre.cmin < 0);
    // the update list of such a frame may still change after it is emitted (every block of the
    // frame sees the same box and reports, so they agree on it)
    const bool deferred = may_wrap || err_candidate;
    if (sh.frame.spill || !bin_frame_box(box).on || v.geom_blocks > kMaxGeomBlocksBinned)
        winners_body<false, false>(v, depth_all, write_target, write_list, sh, pre, gt_all, !deferred, want_box != 0);
    else if (v.rec_packed) winners_body<true, true>(v, depth_all, write_target, write_list, sh, pre, gt_all, !deferred, want_box != 0);
    else winners_body<true, false>(v, depth_all, write_target, write_list, sh, pre, gt_all, !deferred, want_box != 0);
    SSMAP_STAMP(8);
    if (!deferred) return;
    __syncthreads();   // every thread's list entries and the block's header atomics are issued
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(&hdr->pad, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const unsigned n_list = __hip_atomic_load(&hdr->n_winners, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_expect(err_candidate, 0) && index_error_confirmed(v, depth_all + (int64_t)e * v.N, e, sh)) {
        withdraw_update_list(v, e, write_list ? n_list : 0u, write_target);
        if (hv_rule) hv_keep_overlay(v, e, threadIdx.x, blockDim.x, threadIdx.x == 0);
        return;
    }
    if (hv_rule && err_candidate) hv_revert_overlay(v, e, threadIdx.x, blockDim.x);
    if (may_wrap && __hip_atomic_load(&hdr->n_neg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        resolve_wrapped(v, e, n_list, write_target);
    if (write_target == 2 && write_list) {
        __syncthreads();
        commit_heights(v, e, n_list, gt_all, pre.height_map, pre.gt_map, pre.agent_y);
    }
}

void launch_geometry(const View &v, const ssmap_frames &f, const PoseBlock *host_poses, hipStream_t s) {
    const bool prob_rule = v.rule >= SSMAP_RULE_AVG_PROB && v.rule <= SSMAP_RULE_LOG_ODDS;
    const int wt = prob_rule ? (v.direct_update ? 2 : 1) : 0;
    dim3 grid((v.N + kGeomPixPerBlock - 1) / kGeomPixPerBlock, v.E);
    PosePack pack;
    if (host_poses) memcpy(pack.p, host_poses, sizeof(PoseBlock) * v.E);   // at most kPosePack envs
    const int variant = (host_poses ? 4 : 0) | (v.rec_packed ? 2 : 0) | ((v.W & 3) == 0 ? 1 : 0);
#define SSMAP_GEOMETRY(POSE, PACKED, QUAD) \
    hipLaunchKernelGGL((k_geometry<POSE, PACKED, QUAD>), grid, dim3(256), 0, s, v, f.depth, pack, wt)
    switch (variant) {
        case 7: SSMAP_GEOMETRY(true, true, true); break;
        case 6: SSMAP_GEOMETRY(true, true, false); break;
        case 5: SSMAP_GEOMETRY(true, false, true); break;
        case 4: SSMAP_GEOMETRY(true, false, false); break;
        case 3: SSMAP_GEOMETRY(false, true, true); break;
        case 2: SSMAP_GEOMETRY(false, true, false); break;
        case 1: SSMAP_GEOMETRY(false, false, true); break;
        default: SSMAP_GEOMETRY(false, false, false); break;
    }
#undef SSMAP_GEOMETRY
}

void launch_winners(const View &v, const ssmap_frames &f, hipStream_t s) {
    const bool prob_rule = v.rule >= SSMAP_RULE_AVG_PROB && v.rule <= SSMAP_RULE_LOG_ODDS;
    dim3 grid(kWinnerPartsPerEnv, v.E);
    const int hv_rule = (v.rule == SSMAP_RULE_HV || v.rule == SSMAP_RULE_PROB_HV) ? 1 : 0;
    hipLaunchKernelGGL(k_winners, grid, dim3(kWinnerThreads), 0, s, v, f.depth, f.semantic_gt, prob_rule ? (v.direct_update ? 2 : 1) : 0, 1, hv_rule,
                       v.rule == SSMAP_RULE_GOAL_DECAY ? 1 : 0);
}

}  // namespace ssmap
#ifdef SSMAP_TIMELINE
extern "C" int ssmap_debug_timeline(unsigned long long *out, int blocks) {
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssmap::g_timeline), (size_t)blocks * 12 * 8);
}
#endif
