// Stages 2 and 3 of the fusion path for gfx950:
//   k_geometry_count / k_geometry_scatter   depth -> world point -> map cell + height, run twice
//                 over the depth image (count per winner table, then scatter into the tables)
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

// Correctly rounded a / b for a divisor known in advance (rb = RN(1/b) from the host).  Two
// fused steps give a candidate q; the exact residual r = a - b*q then PROVES that q is the
// nearest double (|r| strictly below half an ulp of q times |b|, q not a power of two), in
// which case q equals the IEEE quotient.  `sure` is cleared when the proof fails; the caller
// then redoes the whole pixel with hardware divisions (one rare branch per pixel instead of one
// per division).
__device__ inline double div_known(double a, double b, double rb, bool &sure) {
    const double q0 = a * rb;
    const double q = __builtin_fma(__builtin_fma(-b, q0, a), rb, q0);
    const double r = __builtin_fma(-b, q, a);
    const unsigned long long bits = (unsigned long long)__builtin_bit_cast(long long, q);
    const unsigned ex = (unsigned)(bits >> 52) & 0x7FFu;
    // q must be a normal number away from both ends of the exponent range (half_ulp below is built
    // from its exponent) and not a power of two (the spacing below a power of two is halved)
    const bool plain = ex > 60u && ex < 0x7FEu && (bits & 0xFFFFFFFFFFFFFull) != 0ull;
    const double half_ulp = __builtin_bit_cast(double, (long long)((unsigned long long)(ex - 53u) << 52));
    // a == 0 gives the signed zero a / b gives
    // half_ulp is a power of two, so the product on the right is exact and so is the comparison
    sure &= (a == 0.0) | (plain & (fabs(r) < half_ulp * fabs(b)));
    return a == 0.0 ? q0 : q;
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

// depth -> world point -> map cell and height (mapper.py:18-32, 305-311, 851-863, 319-329);
// dk is the depth after sentinel zeroing, (u, yr) = (column, row - H/2).  FAST uses the verified
// reciprocal forms above and reports through `sure` whether every step was proven exact.
template <bool FAST>
__device__ inline PixelGeom pixel_geometry_impl(const View &v, const PoseBlock &pb, const ssmap_env_maps &mp,
                                                float dk, int u, double yr, bool &sure) {
    PixelGeom g;
    const int Hm = mp.map_h, Wm = mp.map_w;
    const bool ok = dk > 0.0f;
    double x = 0.0, y = 0.0, z = 0.0;
    if (ok) {
        z = (double)dk;
        const double ax = z * ((double)u - (double)v.W / 2.0), ay = z * yr;
        x = FAST ? div_known(ax, v.fx, v.inv_fx, sure) : ax / v.fx;
        y = FAST ? div_known(ay, v.fy, v.inv_fy, sure) : ay / v.fy;
    }
    const double wx = __builtin_fma(pb.rot[2], z, __builtin_fma(pb.rot[1], y, pb.rot[0] * x)) + pb.trans[0];
    const double wy = __builtin_fma(pb.rot[5], z, __builtin_fma(pb.rot[4], y, pb.rot[3] * x)) + pb.trans[1];
    const double wz = __builtin_fma(pb.rot[8], z, __builtin_fma(pb.rot[7], y, pb.rot[6] * x)) + pb.trans[2];
    const double fr = FAST ? floor_div_known(wz - mp.lower[2], v.inv_p2m, sure) : floor((wz - mp.lower[2]) / v.p2m);
    const double fc = FAST ? floor_div_known(wx - mp.lower[0], v.inv_p2m, sure) : floor((wx - mp.lower[0]) / v.p2m);
    g.id = kDeadCell;
    g.key = 0;
    g.row = g.col = 0;
    g.invalid = !ok;
    // comparisons in double so that huge or non-finite coordinates never reach a cast
    const bool below_upper = fr < (double)Hm && fc < (double)Wm;
    const bool above_lower = fr >= -(double)Hm && fc >= -(double)Wm;
    const bool pass = ok && (wy < pb.h_cut);
    g.in_box = below_upper && above_lower && wy == wy;
    g.index_error = !g.in_box && below_upper && !above_lower && pass;   // index below -dim: numpy raises IndexError
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

// The geometry runs twice over the depth image instead of parking 12 bytes per pixel in HBM
// between a counting and a scattering kernel:
//   k_geometry_count    cells only: per-block pixel counts per winner table, per-block partials of
//                       the cell box / lowest invalid pixel / error bits.  No per-pixel output.
//   (k_header_scan      frame header, prefix sums)
//   k_geometry_scatter  the same arithmetic again; zeroes the depth sentinels in place, clears the
//                       owner flags and writes each live pixel's 12-byte record straight into its
//                       table's range (or, for a frame too large to bin, cell ids and heights
//                       per pixel for the scanning winner kernel).
// PACKED: the poses arrive as a kernel argument (batches of at most kPosePack envs: no
// host-to-device copy in front of the step) and block 0 of every env stores its pose for the
// later kernels; otherwise the poses are already in v.poses.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_geometry_count(View v, const float *__restrict__ depth_all, PosePack pack) {
    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const PoseBlock &pb = PACKED ? pack.p[blockIdx.y] : v.poses[e];
    if (PACKED && blockIdx.x == 0 && threadIdx.x < sizeof(PoseBlock) / 8)
        reinterpret_cast<double *>(v.poses + e)[threadIdx.x] = reinterpret_cast<const double *>(&pb)[threadIdx.x];
    const ssmap_env_maps &mp = v.envs[e].maps;

    int rmin = 0x7FFFFFFF, rmax = (int)0x80000000, cmin = 0x7FFFFFFF, cmax = (int)0x80000000;
    unsigned min_inv = kNoPixel;
    unsigned err = 0;
    __shared__ unsigned int s_tab[kGeomBinParts];   // live pixels of this block per winner table
    if (threadIdx.x < kGeomBinParts) s_tab[threadIdx.x] = 0;
    __syncthreads();

    if (i0 < v.N) {
        const int r = i0 / v.W;
        const int c0 = i0 - r * v.W;
        const float4 d4 = *reinterpret_cast<const float4 *>(depth_all + fbase + i0);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w};
        const double yr = (double)r - (double)v.H / 2.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dk = (d[k] == v.depth_min || d[k] == v.depth_max) ? 0.0f : d[k];   // mapper.py:279-283
            const PixelGeom g = pixel_geometry(v, pb, mp, dk, c0 + k, yr);
            if (g.in_box) {
                rmin = min(rmin, g.row); rmax = max(rmax, g.row);
                cmin = min(cmin, g.col); cmax = max(cmax, g.col);
            }
            if (g.index_error) err |= SSMAP_ENV_INDEX_ERROR;
            if (g.invalid) min_inv = min(min_inv, (unsigned)(i0 + k));
            if (g.id != kDeadCell) atomicAdd(&s_tab[geom_cell_table(g.id)], 1u);
        }
    }

    // block reduction of the bbox, the lowest invalid pixel and the error bits
    __shared__ uint32_t s_red[4][4];
    const bool any_cell = rmin <= rmax;
    // biased coordinates fit 14 bits; a thread without cells contributes the neutral 0xFFFF
    const uint32_t lo = any_cell ? ((uint32_t)(rmin + kCoordBias) << 16) | (uint32_t)(cmin + kCoordBias) : 0xFFFFFFFFu;
    const uint32_t hi = any_cell ? ((uint32_t)(0xFFFF - (rmax + kCoordBias)) << 16) | (uint32_t)(0xFFFF - (cmax + kCoordBias)) : 0xFFFFFFFFu;
    const uint32_t w_lo = wave_pk_min16(lo), w_hi = wave_pk_min16(hi);
    const uint32_t w_inv = wave_min_u32(min_inv);
    const uint32_t w_err = __ballot(err != 0) ? SSMAP_ENV_INDEX_ERROR : 0u;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[wave][0] = w_lo; s_red[wave][1] = w_hi; s_red[wave][2] = w_inv; s_red[wave][3] = w_err;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t b_lo = s_red[0][0], b_hi = s_red[0][1];
        min_inv = s_red[0][2];
        err = s_red[0][3];
        for (int w = 1; w < 4; ++w) {
            b_lo = pk_min16(b_lo, s_red[w][0]);
            b_hi = pk_min16(b_hi, s_red[w][1]);
            min_inv = min(min_inv, s_red[w][2]);
            err |= s_red[w][3];
        }
        if (b_lo == 0xFFFFFFFFu) {   // no cell in this block
            rmin = cmin = 0x7FFFFFFF;
            rmax = cmax = (int)0x80000000;
        } else {
            rmin = (int)(b_lo >> 16) - kCoordBias;
            cmin = (int)(b_lo & 0xFFFFu) - kCoordBias;
            rmax = 0xFFFF - (int)(b_hi >> 16) - kCoordBias;
            cmax = 0xFFFF - (int)(b_hi & 0xFFFFu) - kCoordBias;
        }
        // per-block partial; k_header_scan folds them (hundreds of blocks hammering one header
        // line with atomics serialise at the memory side)
        int *part = v.geo_part + ((int64_t)e * gridDim.x + blockIdx.x) * 8;
        *reinterpret_cast<int4 *>(part) = make_int4(rmin, rmax, cmin, cmax);
        *reinterpret_cast<int4 *>(part + 4) = make_int4((int)min_inv, (int)err, 0, 0);
    }
    if (threadIdx.x < kGeomBinParts)   // layout [table][block]; the barrier above covers the LDS adds
        v.bin_cnt[((int64_t)e * kGeomBinParts + threadIdx.x) * gridDim.x + blockIdx.x] = s_tab[threadIdx.x];
}

// Binning: the live pixels of a frame are sorted by winner table into contiguous arrays of
// 12-byte records {height image, slot << 19 | pixel} so that a winner-table block reads only its
// own records instead of every pixel of the frame.  The geometry kernel already counts its
// pixels per table, so two small kernels remain: prefix sums (with the frame header) and scatter.
// Record order inside a table is arbitrary (the table reductions are order independent).
//
// Deal of cells to tables: runs of 8 cells of one map row ("chunks", numbered cc along the row
// in absolute biased coordinates) go to table (cc + 3 row) mod 16.  The table of a cell thus does
// not depend on the frame's cell box, which is why the geometry kernel can count before the box
// is known; the skew by 3 per row spreads a wall seen edge-on over all tables.  Inside a table a
// cell's slot is dense over the box padded to a multiple of 16 chunks per row:
//   slot = ((row - rmin) * per_row + (cc - cc0) / 16) * 8 + col % 8,   per_row = ceil(ncc / 16).
// A box whose padded cells exceed 16 tables of kSlots (about 131 thousand cells: far beyond what
// a 5 m depth range reaches at 3 cm) is not binned; the winner kernel scans all cell ids then.
constexpr int kBinParts = kMaxParts;          // 16 tables per frame
constexpr int kBinShift = 4;
constexpr int kChunkShift = 3, kChunkMask = (1 << kChunkShift) - 1;
constexpr int kRowSkew = 3;
static_assert((1 << kBinShift) == kBinParts, "kBinShift must be log2 of kBinParts");

struct BinFrame {
    bool on;
    int rmin_b, cc0, per_row;
};
static_assert(kChunkShift == 3 && kRowSkew == 3, "geom_cell_table spells these constants out");
__device__ inline int cell_table(uint32_t id) { return geom_cell_table(id); }
__device__ inline BinFrame bin_frame(const FrameHeader *hdr) {
    BinFrame f;
    f.on = false;
    const int rmin = hdr->rmin, rmax = hdr->rmax, cmin = hdr->cmin, cmax = hdr->cmax;
    if (rmin > rmax) return f;
    f.rmin_b = rmin + kCoordBias;
    f.cc0 = (cmin + kCoordBias) >> kChunkShift;
    const int cc1 = (cmax + kCoordBias) >> kChunkShift;
    f.per_row = (cc1 - f.cc0 + kBinParts) >> kBinShift;
    f.on = ((long long)(rmax - rmin + 1) * f.per_row << kChunkShift) <= kSlots;
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
    const int row_rel = kc / f.per_row, kk = kc - row_rel * f.per_row;
    const int row_b = f.rmin_b + row_rel;
    const int j = ((table - f.cc0 - row_b * kRowSkew) & (kBinParts - 1)) + (kk << kBinShift);
    row = row_b - kCoordBias;
    col = (((f.cc0 + j) << kChunkShift) | (slot & kChunkMask)) - kCoordBias;
}

// One block per frame: folds the geometry blocks' partials into the frame header, resets the
// per-step counters, and turns the per-block table counts bin_cnt[p][b] into the first record
// index of block b's pixels for table p; part_start[p] = first record of table p.
__global__ __launch_bounds__(1024) void k_header_scan(View v, int nbin) {
    __shared__ unsigned int total[kBinParts + 1];
    __shared__ int s_red[16][6];
    __shared__ FrameHeader s_hdr;
    const int e = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        int rmin = 0x7FFFFFFF, rmax = (int)0x80000000, cmin = 0x7FFFFFFF, cmax = (int)0x80000000;
        unsigned min_inv = kNoPixel, err = 0;
        for (int b = threadIdx.x; b < nbin; b += blockDim.x) {
            const int *part = v.geo_part + ((int64_t)e * nbin + b) * 8;
            const int4 a = *reinterpret_cast<const int4 *>(part);
            const int4 c = *reinterpret_cast<const int4 *>(part + 4);
            rmin = min(rmin, a.x); rmax = max(rmax, a.y); cmin = min(cmin, a.z); cmax = max(cmax, a.w);
            min_inv = min(min_inv, (unsigned)c.x);
            err |= (unsigned)c.y;
        }
        rmin = wave_min(rmin); rmax = wave_max(rmax);
        cmin = wave_min(cmin); cmax = wave_max(cmax);
        min_inv = wave_minu(min_inv);
        for (int o = 32; o > 0; o >>= 1) err |= (unsigned)__shfl_xor((int)err, o);
        if (lane == 0) {
            s_red[wave][0] = rmin; s_red[wave][1] = rmax; s_red[wave][2] = cmin; s_red[wave][3] = cmax;
            s_red[wave][4] = (int)min_inv; s_red[wave][5] = (int)err;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 16; ++w) {
                rmin = min(rmin, s_red[w][0]); rmax = max(rmax, s_red[w][1]);
                cmin = min(cmin, s_red[w][2]); cmax = max(cmax, s_red[w][3]);
                min_inv = min(min_inv, (unsigned)s_red[w][4]);
                err |= (unsigned)s_red[w][5];
            }
            FrameHeader h;
            h.rmin = rmin; h.rmax = rmax; h.cmin = cmin; h.cmax = cmax;
            h.wrmin = h.wcmin = 0x7FFFFFFF;
            h.wrmax = h.wcmax = (int32_t)0x80000000;
            h.min_invalid_pixel = min_inv;
            h.n_winners = 0;
            h.status = err;
            h.found = 0;
            h.any_goal = 0;
            h.n_overlay = 0;
            h.n_neg = 0;
            h.pad = 0;
            s_hdr = h;
            v.hdr[e] = h;
            v.overlay_count[v.cur_overlay][e] = 0;
        }
        __syncthreads();
    }
    if (!bin_frame(&s_hdr).on) return;
    unsigned int *cnt = v.bin_cnt + (int64_t)e * kBinParts * nbin;
    {
        const int p = wave;       // 16 waves, 16 tables
        unsigned sum = 0;
        for (int b = lane; b < nbin; b += 64) sum += cnt[p * nbin + b];
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) total[p] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int p = 0; p < kBinParts; ++p) { const unsigned c = total[p]; total[p] = run; run += c; }
        total[kBinParts] = run;
    }
    __syncthreads();
    if (threadIdx.x <= kBinParts) v.part_start[e * (kBinParts + 1) + threadIdx.x] = total[threadIdx.x];
    {
        const int p = wave;
        unsigned run = total[p];
        for (int b0 = 0; b0 < nbin; b0 += 64) {
            const int b = b0 + lane;
            const unsigned c = b < nbin ? cnt[p * nbin + b] : 0u;
            unsigned incl = c;   // inclusive wave scan
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            if (b < nbin) cnt[p * nbin + b] = run + incl - c;
            run += __shfl(incl, 63);
        }
    }
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_geometry_scatter(View v, float *__restrict__ depth_all, int write_target) {
    __shared__ unsigned int cnt[kBinParts];
    __shared__ unsigned int base[kBinParts];
    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    const PoseBlock &pb = v.poses[e];
    const ssmap_env_maps &mp = v.envs[e].maps;
    const BinFrame f = bin_frame(v.hdr + e);
    if (threadIdx.x < kBinParts) {
        cnt[threadIdx.x] = 0;
        base[threadIdx.x] = f.on ? v.bin_cnt[((int64_t)e * kBinParts + threadIdx.x) * gridDim.x + blockIdx.x] : 0u;
    }
    __syncthreads();
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= v.N) return;
    const int r = i0 / v.W;
    const int c0 = i0 - r * v.W;
    const float4 d4 = *reinterpret_cast<const float4 *>(depth_all + fbase + i0);
    float d[4] = {d4.x, d4.y, d4.z, d4.w};
    const double yr = (double)r - (double)v.H / 2.0;
    bool dirty = false;
    uint32_t ids[4];
    unsigned long long hk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (d[k] == v.depth_min || d[k] == v.depth_max) { d[k] = 0.0f; dirty = true; }
        const PixelGeom g = pixel_geometry(v, pb, mp, d[k], c0 + k, yr);
        ids[k] = g.id;
        hk[k] = g.key;
    }
    if (dirty) *reinterpret_cast<float4 *>(depth_all + fbase + i0) = make_float4(d[0], d[1], d[2], d[3]);
    if (write_target) *reinterpret_cast<uint32_t *>(v.wflag + fbase + i0) = 0u;   // four pixels' flags
    if (!f.on) {
        // not binned: the scanning winner kernel reads cell id and height per pixel
        *reinterpret_cast<uint4 *>(v.cellid + fbase + i0) = make_uint4(ids[0], ids[1], ids[2], ids[3]);
        unsigned long long *hp = v.horder + fbase + i0;
        *reinterpret_cast<ulonglong2 *>(hp) = make_ulonglong2(hk[0], hk[1]);
        *reinterpret_cast<ulonglong2 *>(hp + 2) = make_ulonglong2(hk[2], hk[3]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ids[k] == kDeadCell) continue;
        const int part = cell_table(ids[k]), slot = cell_slot(f, ids[k]);
        const unsigned pos = base[part] + atomicAdd(&cnt[part], 1u);
        v.rec_h[fbase + pos] = hk[k];
        if (PACKED) {
            v.rec_w[fbase + pos] = ((unsigned)slot << kPackPixBits) | (unsigned)(i0 + k);
        } else {
            v.rec_w[fbase + pos] = (unsigned)(i0 + k);
            v.rec_s[fbase + pos] = (unsigned)slot;
        }
    }
}

// One block = one table of kSlots consecutive cells of the frame's cell bounding box
// (cells are numbered row-major inside the box).  Every block scans all cell ids of
// its frame and keeps the pixels that fall into its table:
//   phase a   table_h[slot] = max height image            (64-bit integer LDS max)
//   phase b   table_i[slot] = min pixel index among the pixels that reach that max
//   phase c   slots whose winner passes the filters are appended to the update list
// Integer max / min are order independent, so the result is deterministic.
struct WinnerShared {
    unsigned long long table_h[kSlots];
    unsigned int table_i[kSlots];
    unsigned int s_base, s_neg;
    unsigned int s_cnt[(kSlots / kWinnerThreads) * (kWinnerThreads / 64)];
    int s_box[kWinnerThreads / 64][4];
};

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
__device__ inline Rec<PACKED> load_rec(const View &v, int64_t fbase, unsigned t);
template <>
__device__ inline Rec<true> load_rec<true>(const View &v, int64_t fbase, unsigned t) {
    Rec<true> r;
    r.h = v.rec_h[fbase + t];
    r.w = v.rec_w[fbase + t];
    return r;
}
template <>
__device__ inline Rec<false> load_rec<false>(const View &v, int64_t fbase, unsigned t) {
    Rec<false> r;
    r.h = v.rec_h[fbase + t];
    r.w = v.rec_w[fbase + t];
    r.s = v.rec_s[fbase + t];
    return r;
}

template <bool RUNS, bool PACKED>
__device__ __forceinline__ void winners_body(const View &v, int write_target, int write_list, WinnerShared &sh) {
    unsigned long long (&table_h)[kSlots] = sh.table_h;
    unsigned int (&table_i)[kSlots] = sh.table_i;
    unsigned int &s_base = sh.s_base, &s_neg = sh.s_neg;
    unsigned int (&s_cnt)[(kSlots / kWinnerThreads) * (kWinnerThreads / 64)] = sh.s_cnt;
    int (&s_box)[kWinnerThreads / 64][4] = sh.s_box;

    const int e = blockIdx.y;
    const int64_t fbase = (int64_t)e * v.N;
    FrameHeader *hdr = v.hdr + e;
    const int rmin = hdr->rmin, rmax = hdr->rmax, cmin = hdr->cmin, cmax = hdr->cmax;
    if (rmin > rmax) return;
    const ssmap_env_maps &mp = v.envs[e].maps;
    const PoseBlock &pb = v.poses[e];
    const int Hm = mp.map_h, Wm = mp.map_w;
    const int bw = cmax - cmin + 1;
    const long long ncell = (long long)bw * (long long)(rmax - rmin + 1);
    const int nfull = (int)((ncell + kSlots - 1) >> kSlotShift);
    const int rmin_b = rmin + kCoordBias, cmin_b = cmin + kCoordBias;
    const int nparts = RUNS ? kBinParts : nfull;

    // the cell all invalid-depth pixels fall into (they sit at the camera position,
    // mapper.py:28-31) and their common height
    const unsigned min_inv = hdr->min_invalid_pixel;
    long long inv_L = -1;
    unsigned long long inv_key = 0;
    if (min_inv != kNoPixel) {
        const double fr = floor(((0.0 + pb.trans[2]) - mp.lower[2]) / v.p2m);
        const double fc = floor(((0.0 + pb.trans[0]) - mp.lower[0]) / v.p2m);
        if (fabs(fr) < 1.0e9 && fabs(fc) < 1.0e9) {
            const int row = (int)fr, col = (int)fc;
            if (row >= rmin && row <= rmax && col >= cmin && col <= cmax) {
                inv_L = (long long)(row - rmin) * bw + (col - cmin);
                inv_key = order_f64(0.0 + pb.trans[1]);
            }
        }
    }

    int inv_part = -1, inv_slot = 0;
    if (inv_L >= 0) {
        if (RUNS) {
            const uint32_t id = ((uint32_t)((int)(inv_L / bw) + rmin_b) << kCoordBits) | (uint32_t)((int)(inv_L % bw) + cmin_b);
            inv_part = cell_table(id);
            inv_slot = cell_slot(bin_frame(hdr), id);
        } else {
            inv_part = (int)(inv_L >> kSlotShift);
            inv_slot = (int)(inv_L & (kSlots - 1));
        }
    }
    const uint4 *ids4 = reinterpret_cast<const uint4 *>(v.cellid + fbase);
    const unsigned long long *hk = v.horder + fbase;
    const int n4 = v.N >> 2;
    // binned records of this frame, contiguous per table
    const unsigned int *part_start = v.part_start + e * (kMaxParts + 1);
    constexpr int kRunBatch = 4;
    constexpr int kKeep = RUNS ? (PACKED ? 19 : 14) : 1;   // records per thread kept in registers between the phases
    Rec<PACKED> keep[kKeep];
    int lrmin = 0x7FFFFFFF, lrmax = (int)0x80000000, lcmin = 0x7FFFFFFF, lcmax = (int)0x80000000;
    if (threadIdx.x == 0) s_neg = 0;

    for (int p = blockIdx.x; p < nparts; p += gridDim.x) {
        const unsigned rec_beg = RUNS ? part_start[p] : 0u, rec_end = RUNS ? part_start[p + 1] : 0u;
        for (int s = threadIdx.x; s < kSlots; s += kWinnerThreads) { table_h[s] = 0ull; table_i[s] = kNoPixel; }
        __syncthreads();
        const bool own_inv = inv_L >= 0 && inv_part == p;
        if (own_inv && threadIdx.x == 0) table_h[inv_slot] = inv_key;
        __syncthreads();

        // phase a
        if (RUNS) {
            // the first kKeep records of every thread stay in registers for phase b
#pragma unroll
            for (int u = 0; u < kKeep; ++u) {
                const unsigned t = rec_beg + threadIdx.x + u * kWinnerThreads;
                if (t < rec_end) keep[u] = load_rec<PACKED>(v, fbase, t);
            }
#pragma unroll
            for (int u = 0; u < kKeep; ++u)
                if (rec_beg + threadIdx.x + u * kWinnerThreads < rec_end)
                    atomicMax(&table_h[keep[u].slot()], keep[u].h);
            for (unsigned t0 = rec_beg + kKeep * kWinnerThreads + threadIdx.x; t0 < rec_end; t0 += kWinnerThreads * kRunBatch) {
                Rec<PACKED> rec[kRunBatch];
#pragma unroll
                for (int u = 0; u < kRunBatch; ++u) {
                    const unsigned t = t0 + u * kWinnerThreads;
                    if (t < rec_end) rec[u] = load_rec<PACKED>(v, fbase, t);
                }
#pragma unroll
                for (int u = 0; u < kRunBatch; ++u)
                    if (t0 + u * kWinnerThreads < rec_end)
                        atomicMax(&table_h[rec[u].slot()], rec[u].h);
            }
        } else {
        for (int q = threadIdx.x; q < n4; q += kWinnerThreads) {
            const uint4 id4 = ids4[q];
            const uint32_t ids[4] = {id4.x, id4.y, id4.z, id4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t id = ids[k];
                if (id == kDeadCell) continue;
                const int rr = (int)((id >> kCoordBits) & kCoordMask) - rmin_b;
                const int cc = (int)(id & kCoordMask) - cmin_b;
                const int L = rr * bw + cc;
                if ((L >> kSlotShift) == p) atomicMax(&table_h[L & (kSlots - 1)], hk[4 * q + k]);
            }
        }
        }
        __syncthreads();
        // phase b
        if (own_inv && threadIdx.x == 0) {
            const int s = inv_slot;
            if (table_h[s] == inv_key) table_i[s] = (min_inv << 1) | 1u;
        }
        __syncthreads();
        if (RUNS) {
#pragma unroll
            for (int u = 0; u < kKeep; ++u)
                if (rec_beg + threadIdx.x + u * kWinnerThreads < rec_end &&
                    table_h[keep[u].slot()] == keep[u].h)
                    atomicMin(&table_i[keep[u].slot()], keep[u].word());
            for (unsigned t0 = rec_beg + kKeep * kWinnerThreads + threadIdx.x; t0 < rec_end; t0 += kWinnerThreads * kRunBatch) {
                Rec<PACKED> rec[kRunBatch];
#pragma unroll
                for (int u = 0; u < kRunBatch; ++u) {
                    const unsigned t = t0 + u * kWinnerThreads;
                    if (t < rec_end) rec[u] = load_rec<PACKED>(v, fbase, t);
                }
#pragma unroll
                for (int u = 0; u < kRunBatch; ++u)
                    if (t0 + u * kWinnerThreads < rec_end &&
                        table_h[rec[u].slot()] == rec[u].h)
                        atomicMin(&table_i[rec[u].slot()], rec[u].word());
            }
        } else {
        for (int q = threadIdx.x; q < n4; q += kWinnerThreads) {
            const uint4 id4 = ids4[q];
            const uint32_t ids[4] = {id4.x, id4.y, id4.z, id4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t id = ids[k];
                if (id == kDeadCell) continue;
                const int rr = (int)((id >> kCoordBits) & kCoordMask) - rmin_b;
                const int cc = (int)(id & kCoordMask) - cmin_b;
                const int L = rr * bw + cc;
                if ((L >> kSlotShift) == p) {
                    const int s = L & (kSlots - 1);
                    if (hk[4 * q + k] == table_h[s])
                        atomicMin(&table_i[s], (unsigned)(4 * q + k) << 1);
                }
            }
        }
        }
        __syncthreads();
        // phase c: emit the surviving slots in slot order (= row-major cell order), so that
        // neighbouring list entries are neighbouring map cells
        constexpr int kIters = kSlots / kWinnerThreads, kWaves = kWinnerThreads / 64;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned long long balls[kIters];
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const unsigned w = table_i[it * kWinnerThreads + threadIdx.x];
            // updatable: not the invalid-depth entry, and below the height cut-off
            balls[it] = __ballot(w != kNoPixel && !(w & 1u) &&
                                 unorder_f64(table_h[it * kWinnerThreads + threadIdx.x]) < pb.h_cut);
            if (lane == 0) s_cnt[it * kWaves + wave] = (unsigned)__popcll(balls[it]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned run = 0;
            for (int k = 0; k < kIters * kWaves; ++k) { const unsigned c = s_cnt[k]; s_cnt[k] = run; run += c; }
            s_base = run ? atomicAdd(&hdr->n_winners, run) : 0u;
        }
        __syncthreads();
        unsigned n_neg = 0;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int s = it * kWinnerThreads + threadIdx.x;
            if (!((balls[it] >> lane) & 1ull)) continue;
            const unsigned w = table_i[s];
            int row, col;
            if (RUNS) {
                slot_cell(bin_frame(hdr), p, s, row, col);   // recomputed here: keeps it out of the registers above
            } else {
                // cell numbers fit 32 bits (a box holds at most 4 * 8192^2 = 2^28 cells)
                const unsigned L = ((unsigned)p << kSlotShift) + (unsigned)s;
                const unsigned lrow = L / (unsigned)bw;
                row = (int)lrow + rmin;
                col = (int)(L - lrow * (unsigned)bw) + cmin;
            }
            lrmin = min(lrmin, row); lrmax = max(lrmax, row);
            lcmin = min(lcmin, col); lcmax = max(lcmax, col);
            // numpy index wrap for -dim <= idx < 0 (mapper.py:326-338 masks only the upper bound)
            const int wr = row < 0 ? row + Hm : row, wc = col < 0 ? col + Wm : col;
            const int cell = wr * Wm + wc;
            const int pix = (int)(w >> 1);
            n_neg += (row < 0 || col < 0) ? 1u : 0u;
            if (write_list) {
                const unsigned pos = s_base + s_cnt[it * kWaves + wave] +
                                     (unsigned)__popcll(balls[it] & ((1ull << lane) - 1ull));
                if (write_target) { v.target[fbase + pix] = (int)pos; v.wflag[fbase + pix] = 1; }
                int2 *out = reinterpret_cast<int2 *>(v.winners) + fbase + pos;
                const uint32_t word = (uint32_t)pix | (row < 0 ? kRowNegBit : 0u) | (col < 0 ? kColNegBit : 0u);
                *out = make_int2((int)word, cell);
                v.wheight[fbase + pos] = table_h[s];
            }
        }
        if (n_neg) atomicAdd(&s_neg, n_neg);
        __syncthreads();
    }

    // bbox of the update cells (goal_decay works on it, mapper.py:1117-1118)
    lrmin = wave_min(lrmin); lrmax = wave_max(lrmax);
    lcmin = wave_min(lcmin); lcmax = wave_max(lcmax);
    // one set of header atomics per block (same-line atomics serialise at the memory side)
    __syncthreads();
    if (threadIdx.x == 0 && s_neg) atomicAdd(&hdr->n_neg, s_neg);
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

// Frames whose padded cell box fits the 16 tables are binned (k_header_scan, k_geometry_scatter) and
// read as contiguous record ranges; larger boxes fall back to scanning every cell id once per table.
__global__ __launch_bounds__(kWinnerThreads) void k_winners(View v, int write_target, int write_list) {
    __shared__ WinnerShared sh;
    const FrameHeader *hdr = v.hdr + blockIdx.y;
    if (hdr->rmin > hdr->rmax) return;
    if (!bin_frame(hdr).on) winners_body<false, false>(v, write_target, write_list, sh);
    else if (v.rec_packed) winners_body<true, true>(v, write_target, write_list, sh);
    else winners_body<true, false>(v, write_target, write_list, sh);
}

// numpy keeps only the LAST write when one fancy-index assignment names a cell twice, which
// happens when an update cell with a negative index wraps onto another update cell
// (mapper.py:326-338 masks only the upper bounds).  At most four update pixels can reach one
// wrapped cell -- one per sign quadrant of (row, col) -- so frames with wrapped cells record
// each update pixel in its quadrant slot of the cell; the highest pixel index owns the cell
// and the others are flagged.  One block per env: slots are written, then compared.
__global__ __launch_bounds__(1024) void k_resolve_wrapped(View v, int write_target) {
    const int e = blockIdx.x;
    const FrameHeader *hdr = v.hdr + e;
    if (hdr->n_neg == 0) return;
    const unsigned n = hdr->n_winners;
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
            if (write_target) v.wflag[fbase + pix] = 0;
        }
    }
}

void launch_resolve_wrapped(const View &v, hipStream_t s) {
    const bool prob_rule = v.rule >= SSMAP_RULE_AVG_PROB && v.rule <= SSMAP_RULE_LOG_ODDS;
    hipLaunchKernelGGL(k_resolve_wrapped, dim3(v.E), dim3(1024), 0, s, v, prob_rule ? 1 : 0);
}


void launch_geometry(const View &v, const ssmap_frames &f, const PoseBlock *host_poses, hipStream_t s) {
    dim3 grid((v.N + kGeomPixPerBlock - 1) / kGeomPixPerBlock, v.E);
    PosePack pack;
    if (host_poses) {   // at most kPosePack envs
        memcpy(pack.p, host_poses, sizeof(PoseBlock) * v.E);
        hipLaunchKernelGGL(k_geometry_count<true>, grid, dim3(256), 0, s, v, f.depth, pack);
    } else {
        hipLaunchKernelGGL(k_geometry_count<false>, grid, dim3(256), 0, s, v, f.depth, pack);
    }
    hipLaunchKernelGGL(k_header_scan, dim3(v.E), dim3(1024), 0, s, v, (int)grid.x);
}

void launch_winners(const View &v, const ssmap_frames &f, hipStream_t s) {
    const bool prob_rule = v.rule >= SSMAP_RULE_AVG_PROB && v.rule <= SSMAP_RULE_LOG_ODDS;
    dim3 grid(kWinnerPartsPerEnv, v.E);
    dim3 grid_geo((v.N + kGeomPixPerBlock - 1) / kGeomPixPerBlock, v.E);
    if (v.rec_packed) hipLaunchKernelGGL(k_geometry_scatter<true>, grid_geo, dim3(256), 0, s, v, f.depth, prob_rule ? 1 : 0);
    else hipLaunchKernelGGL(k_geometry_scatter<false>, grid_geo, dim3(256), 0, s, v, f.depth, prob_rule ? 1 : 0);
    hipLaunchKernelGGL(k_winners, grid, dim3(kWinnerThreads), 0, s, v, prob_rule ? 1 : 0, 1);
}

}  // namespace ssmap
