// Internal definitions shared by the engine's translation units (host + device).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssmap.h"

namespace ssmap {

constexpr uint32_t kDeadCell = 0xFFFFFFFFu;   // pixel takes no part in the group-by
constexpr uint32_t kNoPixel = 0xFFFFFFFFu;
constexpr int kCoordBias = 8192;              // cell coordinates are stored biased by this
constexpr int kCoordBits = 14;
constexpr uint32_t kCoordMask = (1u << kCoordBits) - 1;
constexpr uint32_t kUpdBit = 1u << 28;        // pixel passes height cut-off, depth and upper bounds
constexpr uint32_t kRowNegBit = 1u << 30;     // update-list pixel word: the cell row was negative
constexpr uint32_t kColNegBit = 1u << 31;     // ... the cell column was negative (numpy wrap)
constexpr uint32_t kLostBit = 1u << 29;       // ... another update pixel owns the same wrapped cell
constexpr uint32_t kPixMask = (1u << 29) - 1;
// direct update: the word a pixel's owner cell travels in (View::ocell)
constexpr uint32_t kCellMask = (1u << 29) - 1;   // the map cell (maps have at most 8192 x 8192 cells)
constexpr uint32_t kCellLowBit = 1u << 29;       // the cell's new height is below 0.1 (and not -inf): "free"
constexpr uint32_t kCellHighBit = 1u << 30;      // the cell's new height is above 0.1: "occupied"
constexpr int kClassWords = 8;                // 32-bit words of a bit per model class (room for 256 classes)
static_assert(SSMAP_MAX_CLASSES <= 32 * kClassWords, "class masks");
constexpr int kSlots = 8192;                  // cells per LDS winner table
constexpr int kSlotShift = 13;
constexpr int kPackPixBits = 32 - kSlotShift;   // slot and pixel share one record word when N <= 2^19
constexpr int kGeomPixPerBlock = 1024;        // geometry kernel: 256 threads x 4 pixels
constexpr int kWinnerThreads = 1024;
constexpr int kMaxParts = 16;                // winner tables per frame on the binned path
constexpr int kWinnerPartsPerEnv = 16;        // blocks launched per env for the winner kernel

// Per-env, per-step geometry constants: the host derives them from the two pose vectors
// exactly as mapper.py:290-301 does (quaternion products, rotation matrix).
struct PoseBlock {
    double rot[9];        // R(q_cam * rotvec(pi,0,0)), row-major
    double trans[3];      // camera position
    double h_cut;         // camera y + height_cutoff_m
    double agent[3];      // agent position
    int32_t robot_row, robot_col;   // world2map(agent position)
    int32_t goal_id;      // schema id of the goal
    int32_t skip;         // the env has no frame in this step: every kernel leaves it alone
    // cell box {rmin, rmax, cmin, cmax} predicted to hold every live pixel of the frame: the camera
    // frustum up to depth_max, padded, cut to the indexable range [-dim, dim).  rmin > rmax: none.
    int32_t box[4];
};

// Engines of at most kPosePack envs hand the poses to the geometry kernel as a kernel argument
// (no host-to-device copy in front of every step); the kernel stores them for the later kernels
// of the step.  Larger engines copy them through a pinned staging ring.
constexpr int kPosePack = 16;
struct PosePack {
    PoseBlock p[kPosePack];
};

// Per-env scratch header, rewritten every step.
struct FrameHeader {
    int32_t rmin, rmax, cmin, cmax;       // predicted box of the live pixels' cells (unwrapped)
    int32_t wrmin, wrmax, wcmin, wcmax;   // bbox of the update cells (unwrapped)
    uint32_t min_invalid_pixel;           // lowest flat index with depth <= 0
    uint32_t n_winners;
    uint32_t status;
    int32_t found;
    uint32_t any_goal;                    // goal_decay: some update pixel carries the goal id
    uint32_t any_eroded;                  // goal_decay: the eroded goal mask of this frame has a set cell
    uint32_t n_neg;                       // update cells reached through a negative (wrapped) index
    uint32_t pad;                         // tickets of the winner blocks of a frame that may hold wrapped cells
};

// What a geometry block reports about its pixels besides the records; the winner kernel folds
// the blocks' reports of a frame.
struct GeomPart {
    uint32_t min_invalid_pixel;           // lowest flat index with depth <= 0, or kNoPixel
    uint32_t err;                         // SSMAP_ENV_* bits
    uint32_t spill;                       // some live pixel of the block is not in the tables
    uint32_t pad;
    int32_t rmin, rmax, cmin, cmax;       // box of the cells of the spilled pixels
};

// Gaussian naive Bayes over four features, as sklearn evaluates it: for class k
//   jll_k = log(prior_k) + (konst_k - 0.5 * sum_j (x_j - theta_kj)^2 / var_kj),
//   konst_k = -0.5 * sum_j log(2 pi var_kj); prediction = label of the first largest jll.
struct NbModel {
    double theta[2][4], var[2][4];
    double log_prior[2], konst[2];
    int32_t classes[2];
    int32_t n_classes;       // 0 = no model set
    int32_t pad;
};

// Pointers typed as global memory.  A pointer that a kernel reads out of a struct in memory is a
// generic (flat) pointer to the compiler; flat loads and stores count on both memory counters of a
// wave, so every wait for an LDS access also waits for them, and their addresses are 64-bit register
// pairs.  The maps are device allocations: their pointers are cast once, where they are taken.
#define SSMAP_GLOBAL __attribute__((address_space(1)))
template <typename T>
using gptr = SSMAP_GLOBAL T *;
template <typename T>
__host__ __device__ inline gptr<T> as_global(T *p) { return (gptr<T>)p; }

// ssmap_env_maps as the kernels use it: by value, pointers typed as global memory.
struct GMaps {
    int32_t map_h, map_w;
    double lower[3];
    gptr<uint32_t> semantic_map;
    gptr<uint8_t> semantic_map_color;
    gptr<float> height_map;
    gptr<uint32_t> gt_map;
    gptr<float> unc_map, prob_map, weight_map, goal_map;
    gptr<uint32_t> background_map;
    gptr<float> hits, views;
    gptr<uint32_t> total_view;
    gptr<float> cumulative_conf, max_conf, max_other_conf;
    __host__ __device__ explicit GMaps(const ssmap_env_maps &m)
        : map_h(m.map_h), map_w(m.map_w), lower{m.lower[0], m.lower[1], m.lower[2]},
          semantic_map(as_global(m.semantic_map)), semantic_map_color(as_global(m.semantic_map_color)),
          height_map(as_global(m.height_map)), gt_map(as_global(m.gt_map)), unc_map(as_global(m.unc_map)),
          prob_map(as_global(m.prob_map)), weight_map(as_global(m.weight_map)), goal_map(as_global(m.goal_map)),
          background_map(as_global(m.background_map)), hits(as_global(m.hits)), views(as_global(m.views)),
          total_view(as_global(m.total_view)), cumulative_conf(as_global(m.cumulative_conf)),
          max_conf(as_global(m.max_conf)), max_other_conf(as_global(m.max_other_conf)) {}
};

struct EnvState {
    ssmap_env_maps maps;
    uint32_t n_overlay_prev;   // unused on host; kept for alignment
    uint32_t bound;
};

// Class counts whose probability rules are updated by the streaming kernel itself (no staged rows, no
// update kernel): all of them.  The counts of the reference's segmentation heads (8, 33, 37, 40, 53,
// 81, 150) have a register kernel compiled for exactly that many classes, any other count runs the
// kernel of the next size up with its spare class registers idle.  At 150 classes the kernel runs at
// two waves per SIMD and its tail is not hidden (200 us of 700 at 16 envs), but staging the rows
// costs more: 100 us for their scattered stores beside the streaming reads and 127 us for the update
// kernel that reads them back from cold caches.
constexpr bool direct_class_count(int c) { return c >= 1 && c <= SSMAP_MAX_CLASSES; }
constexpr bool exact_class_count(int c) { return c == 8 || c == 33 || c == 37 || c == 40 || c == 53 || c == 81 || c == 150; }

// Everything a kernel needs, passed by value.
struct View {
    int32_t E, H, W, N, C, n_schema, rule, use_height;
    float depth_min, depth_max;
    double fx, fy, p2m;
    double inv_fx, inv_fy, inv_p2m;   // correctly rounded reciprocals (host division)
    double inv_fx_wide, inv_fy_wide;  // the same times (1 + 2^-20): the exactness test of div_known
    // scratch (device)
    const EnvState *envs;
    PoseBlock *poses;          // [E] written by the geometry kernel, read by the later kernels
    FrameHeader *hdr;
    // Binned pixel records, one array per field.  Geometry block b owns kGeomPixPerBlock record
    // places in every winner table (all its pixels may fall into one table) and reports how many
    // it used: no block waits for another, and nothing has to be reset between steps.
    int32_t geom_blocks;       // geometry blocks per frame = ceil(N / kGeomPixPerBlock)
    GeomPart *geo_part;        // [E][geom_blocks]
    unsigned int *bin_cnt;     // [E][kMaxParts][geom_blocks] records of block b in table t
    unsigned long long *rec_h; // [E][kMaxParts][geom_blocks][kGeomPixPerBlock] height image
    uint32_t *rec_w;           // same shape: slot << kPackPixBits | pixel when the frame has at most
                               //        2^kPackPixBits pixels (rec_packed), else the pixel
    uint32_t *rec_s;           // same shape: slot (only when !rec_packed)
    int32_t rec_packed;
    int32_t *target;           // [E][N] staged rows: the pixel's slot in the frame's update list, -1 = it owns no cell (cleared by the geometry kernel)
    int32_t *ocell;            // [E][N] direct update (see direct_update): map cell a pixel owns this step (kCellMask)
                               //        with what the occupancy override asks of the cell's new height (kCellHighBit,
                               //        kCellLowBit), -1 = the pixel owns no cell
    int32_t direct_update;     // the streaming kernel updates the owners' map cells itself (no staged rows, no update kernel)
    float *wprob;              // [E][N][wprob_stride] class probabilities of the update pixels, row = list slot
    int32_t wprob_stride;      // floats per row (>= C)
    int32_t *winners;          // [E][N][2] update list: (pixel word, cell) pairs in cell order per table
    unsigned long long *wheight;  // [E][N] height image of each list slot's pixel
    int32_t *quad;             // [E][maxHm*maxWm][4] 1 + update pixel of each index-sign quadrant that
                               // lands on the wrapped cell (0 = none); non-zero only inside a step
    int64_t quad_stride;       // ints per env
    uint8_t *local_ids;        // goal_decay: [E][4*maxHm*maxWm] id+1 on the update bbox
    uint8_t *local_eroded;     // goal_decay: same shape
    int32_t *overlay[2];       // hv: [E][N] cells showing the goal overlay (prev / current)
    uint32_t *overlay_count[2];// [E]
    int32_t cur_overlay;       // which of the two lists is written this step
    int64_t local_stride;      // bytes per env in local_ids
    int32_t local_clean;       // goal_decay: local_ids is zero between steps (the found kernel wipes what a step wrote)
    const int32_t *lut;        // [C] model class -> schema id
    const int32_t *pool_ids;   // [6][SSMAP_MAX_POOL]
    const int32_t *pool_count; // [6]
    int32_t pool_trivial;      // no goal id pools more than one class
    // for the register streaming kernel, by value: pool_mask bit c = class c is in some list; its
    // staging slot = number of such classes below c; pool_pack byte g * SSMAP_MAX_POOL + k = slot of the
    // k-th class of goal g's list, word 12 = six 4-bit list lengths; pool_slots = classes in lists
    uint32_t pool_mask[kClassWords];
    uint32_t pool_pack[13];
    int32_t pool_slots;
    int32_t *path_cells;       // [E][path_cap] cells of the agent's trail (path overlay)
    uint32_t *path_count;      // [E]
    int64_t path_cap;          // max_map_h * max_map_w
    const NbModel *nb;         // [6] by goal id - 3 (stubborn rule)
    const float *nb_probs;     // [E][C][N] class probabilities of this step for the goal statistics
    const uint8_t *palette;    // [palette_rows][3]
    int32_t palette_rows;
    const int32_t *disk_hw;    // [radius+1] half-widths of the found disk
    int32_t disk_radius;
    ssmap_params prm;
    // measurement knobs (SSMAP_EXP, SSMAP_EXP_LDS in the environment; wrong results, safe addresses): parts
    // of the direct update switched off, to see what each costs.  1 nothing after the row requests,
    // 2 nothing after the owners have parked their rows, 4 all arithmetic but no stores, 8 every row of
    // cell 0, 16 no label / colour / weight stores, 32 no map-row stores
    int32_t exp, exp_lds;
    int32_t force_generic;     // SSMAP_GENERIC=1 at create: the generic (LDS column) streaming kernel whatever the class count (tests)
};

__host__ __device__ inline unsigned long long order_f64(double h) {
    h = h + 0.0;   // -0.0 -> +0.0 so that equal values have equal images
    unsigned long long b = (unsigned long long)__builtin_bit_cast(long long, h);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double unorder_f64(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __builtin_bit_cast(double, (long long)b);
}

// ---- egocentric observation maps (mapper.py:876-998)
constexpr int kEgoMaxOut = 256;        // output side at most
constexpr int kEgoTapPool = 16384;     // area-table entries of both crops and both axes
constexpr int kEgoMaxRadius = 63;      // agent disk radius at most
constexpr int kEgoTapLimit = 32;       // area-table taps per output pixel and axis at most (shrink factors up to 30)
constexpr int kEgoPack = 16;           // envs per launch (their poses travel as a kernel argument)
constexpr int kEgoArea = 0, kEgoLinear = 1, kEgoBox = 2;   // kEgoBox: integer shrink factors (plain box mean)

// Per env and call: inverse rotation (OpenCV's inverted 2x3 matrix), crop origin, goal.
struct EgoPose {
    double m[6];
    int32_t row0, col0;    // global cell of pixel (0, 0) of the crop (may lie outside the map)
    int32_t goal_id;
    int32_t has_aux;
    uint8_t aux_color[3 * SSMAP_MAX_AUX];   // colour of each auxiliary-angle square
};
struct EgoPack {
    EgoPose p[kEgoPack];
};
// One axis of one resize: either an area table (shrinking) or bilinear area-mode weights.
struct EgoAxis {
    int32_t ssize;
    int32_t pad;
    double scale;                       // 1 / (dsize / ssize), as OpenCV derives it
    int32_t first[kEgoMaxOut], count[kEgoMaxOut];                        // area: slice of the tap pool
    int32_t lin_src[kEgoMaxOut], lin_w0[kEgoMaxOut], lin_w1[kEgoMaxOut]; // enlarging
};
struct EgoTables {
    int32_t side, origin, out_w, out_h;
    int32_t crop_off[2], mode[2];       // [0] small crop, [1] large crop
    float box_scale[2];                 // kEgoBox: 1 / (factor_x * factor_y) in float32
    int32_t box_area[2];                // kEgoBox: factor_x * factor_y
    int32_t agent_radius, aux_half, n_aux;
    int32_t aux_reach;                  // no auxiliary square reaches further from the centre than this
    int32_t max_taps;                   // largest tap count of any output pixel and axis
    int32_t agent_hw[kEgoMaxRadius + 1];
    int32_t aux_points[2 * SSMAP_MAX_AUX];
    uint8_t active_goal_color[4], arrow_color[4];
    EgoAxis axis[2][2];                 // [crop][0 = x, 1 = y]
    int32_t tap_src[kEgoTapPool];
    float tap_w[kEgoTapPool];
};

// launchers (defined in the .hip files)
void launch_geometry(const View &v, const ssmap_frames &f, const PoseBlock *host_poses, hipStream_t s);
void launch_winners(const View &v, const ssmap_frames &f, hipStream_t s);
void launch_stream(const View &v, const ssmap_frames &f, bool fuse, float *probs_out, hipStream_t s);
void launch_update_prob(const View &v, const ssmap_frames &f, hipStream_t s);
void launch_label_rules(const View &v, const ssmap_frames &f, hipStream_t s);
void launch_found(const View &v, hipStream_t s);
void launch_gt_goal(const View &v, int what, int32_t *out, hipStream_t s);
void launch_path_paint(const View &v, uint32_t max_count, hipStream_t s);
void launch_ego_maps(const View &v, const EgoPose *host_poses, const EgoTables *d_tables, const EgoTables &h_tables,
                     uint8_t *ego_small, uint8_t *ego_large, float *unc_small, float *unc_large, hipStream_t s);

}  // namespace ssmap
