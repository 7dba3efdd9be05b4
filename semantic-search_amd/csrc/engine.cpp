// Host side of the C ABI declared in include/ssmap.h: engine life cycle, per-step pose
// preparation (mapper.py:290-301) and kernel sequencing.  No torch types, no exceptions
// across the boundary.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "ssmap_internal.h"

using namespace ssmap;

namespace {

constexpr int kPoseRing = 8;
thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t err__ = (expr);                                                         \
        if (err__ != hipSuccess)                                                           \
            return fail(SSMAP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__)); \
    } while (0)


struct Quat { double w, x, y, z; };

// Hamilton product, evaluated left to right like numpy-quaternion's C kernel.
Quat qmul(const Quat &a, const Quat &b) {
    return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}

// quaternion.from_rotation_vector: exp of the pure quaternion r/2.
Quat qexp_half(double rx, double ry, double rz) {
    const double x = rx / 2, y = ry / 2, z = rz / 2;
    const double n = std::sqrt(x * x + y * y + z * z);
    if (n > 1e-14) {
        const double s = std::sin(n) / n, e = std::exp(0.0);
        return Quat{e * std::cos(n), e * s * x, e * s * y, e * s * z};
    }
    return Quat{std::exp(0.0), 0, 0, 0};
}

// quaternion.as_rotation_matrix (array branch: entries divided by the Cayley norm).
void qmatrix(const Quat &q, double *m) {
    const double n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    m[0] = 1.0 - 2 * (q.y * q.y + q.z * q.z) / n;
    m[1] = 2 * (q.x * q.y - q.z * q.w) / n;
    m[2] = 2 * (q.x * q.z + q.y * q.w) / n;
    m[3] = 2 * (q.x * q.y + q.z * q.w) / n;
    m[4] = 1.0 - 2 * (q.x * q.x + q.z * q.z) / n;
    m[5] = 2 * (q.y * q.z - q.x * q.w) / n;
    m[6] = 2 * (q.x * q.z - q.y * q.w) / n;
    m[7] = 2 * (q.y * q.z + q.x * q.w) / n;
    m[8] = 1.0 - 2 * (q.x * q.x + q.y * q.y) / n;
}

int clamp_cell(double f) {
    if (!(f > -1.0e9)) return -1000000000;
    if (!(f < 1.0e9)) return 1000000000;
    return (int)f;
}

// half widths of OpenCV's filled circle (drawing.cpp Circle(), fill branch)
std::vector<int32_t> disk_halfwidths(int radius) {
    std::vector<int32_t> hw(radius + 1, -1);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hw[dy] = std::max(hw[dy], dx);
        hw[dx] = std::max(hw[dx], dy);
        ++dy;
        err += plus;
        plus += 2;
        if (err > 0) { err -= minus; --dx; minus -= 2; }
    }
    return hw;
}

template <typename T>
int dev_alloc(T **p, size_t count) {
    *p = nullptr;
    if (count == 0) return SSMAP_OK;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return SSMAP_OK;
}

int dev_zero(void *p, size_t bytes) {
    if (bytes == 0) return SSMAP_OK;
    HIP_TRY(hipMemset(p, 0, bytes));
    return SSMAP_OK;
}

// Per-step scratch: everything a step writes and reads again within the same step (pixel records,
// per-block counts and reports, owner flags, update list, staged probability rows, goal_decay's
// local maps).  One layout for both homes of it: the block ssmap_create allocates, and the
// file-scope array below that ssmap_step_noalloc uses instead.  Returns the bytes needed; with
// `base` the pointers of `v` are set.
size_t carve_step_scratch(const ssmap_config &c, View &v, unsigned char *base) {
    size_t at = 0;
    auto take = [&](auto *&ptr, size_t count) {
        using T = std::remove_reference_t<decltype(*ptr)>;
        at = (at + 255) & ~size_t(255);
        if (base) ptr = count ? reinterpret_cast<T *>(base + at) : nullptr;
        at += count * sizeof(T);
    };
    const bool prob_rule = c.rule >= SSMAP_RULE_AVG_PROB && c.rule <= SSMAP_RULE_LOG_ODDS;
    const size_t E = (size_t)c.n_envs, N = (size_t)c.img_h * c.img_w, EN = E * N;
    const size_t geom_blocks = (N + kGeomPixPerBlock - 1) / kGeomPixPerBlock;
    const size_t n_rec = E * kMaxParts * geom_blocks * kGeomPixPerBlock;
    const bool packed = N <= (size_t(1) << kPackPixBits);
    take(v.geo_part, E * geom_blocks);
    take(v.bin_cnt, E * kMaxParts * geom_blocks);
    take(v.rec_h, n_rec);
    take(v.rec_w, n_rec);
    take(v.rec_s, packed ? 0 : n_rec);
    const bool direct = prob_rule && v.direct_update;
    take(v.target, prob_rule && !direct ? EN : 0);
    take(v.wprob, prob_rule && !direct ? EN * (size_t)(v.wprob_stride > c.n_classes ? v.wprob_stride : c.n_classes) : 0);
    take(v.ocell, direct ? EN : 0);
    take(v.winners, EN * 2);
    take(v.wheight, EN);
    const size_t local = c.rule == SSMAP_RULE_GOAL_DECAY ? E * 4 * (size_t)c.max_map_h * c.max_map_w : 0;
    take(v.local_ids, local);
    take(v.local_eroded, local);
    return (at + 255) & ~size_t(255);
}

// Home of the per-step scratch for ssmap_step_noalloc: a file-scope device array, one copy per
// GPU that loads this library, held until the process ends.  In host code the name is only a
// handle for hipGetSymbolAddress; the device address it returns (per device) is what kernels and
// copies get.
constexpr size_t kStaticScratchBytes = SSMAP_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

}  // namespace

struct ssmap_engine {
    ssmap_config cfg{};
    View view{};
    std::vector<EnvState> envs;        // host mirror
    EnvState *d_envs = nullptr;
    PoseBlock *d_poses = nullptr;
    std::vector<PoseBlock> h_poses;       // this step's poses when they travel as kernel arguments
    std::vector<PoseBlock> last_poses;    // pose of every env's last frame (a skipped env keeps it)
    std::vector<uint8_t> has_frame;       // a frame has arrived for the env since its maps were bound
    PoseBlock *h_ring[kPoseRing] = {};    // pinned staging otherwise (more than kPosePack envs)
    hipEvent_t pose_done[kPoseRing] = {};
    int ring = 0;
    FrameHeader *d_hdr = nullptr;
    std::vector<FrameHeader> h_hdr;
    int32_t *d_lut = nullptr, *d_pool_ids = nullptr, *d_pool_count = nullptr, *d_disk = nullptr;
    int32_t *d_query = nullptr;            // [E][2] answers of ssmap_gt_goal_query
    std::vector<int32_t> h_query;
    uint8_t *d_palette = nullptr;
    NbModel *d_nb = nullptr;               // [6] naive-Bayes models of the stubborn rule, by goal id - 3
    bool nb_set[6] = {};
    std::vector<uint32_t> path_n;          // host mirror of view.path_count
    float *d_nb_probs = nullptr;           // [E][C][N] stage-1 probabilities when that rule is fed logits
    int palette_rows = 0;
    bool have_params = false, have_schema = false, have_palette = false;
    Quat flip_cam{}, flip_agent{};
    int device = 0;
    unsigned char *d_step = nullptr;       // per-step scratch of ssmap_step (carve_step_scratch)
    unsigned char *d_static = nullptr;     // this device's copy of the file-scope array, for ssmap_step_noalloc
    size_t step_bytes = 0;
    const int32_t *last_list = nullptr;    // update list of the last step (either home)
    bool stepped = false;                  // some step has run (its poses and goal ids are on the device)
    // egocentric observation maps (ssmap_set_egomap / ssmap_ego_maps)
    EgoTables *d_ego = nullptr;
    std::vector<EgoTables> h_ego;          // one element once configured
    std::vector<EgoPose> h_ego_poses;
    // optional per-stage timing (ssmap_profile_begin / ssmap_profile_read)
    std::vector<hipEvent_t> prof_events;   // kProfMarks per recorded step
    int prof_capacity = 0, prof_steps = 0;
};

constexpr int kProfMarks = 6;   // 5 stage intervals

// Cell box that holds every pixel of a frame whose depth is at most depth_max: the camera position
// and the four far corners of the view pyramid span it (world = R p + t is linear in the camera
// point), padded by two cells and cut to the range numpy can index, [-dim, dim).  A container only:
// a pixel outside it makes the kernels take their scanning path, never a wrong result.
static void predict_box(const ssmap_config &c, const ssmap_env_maps &mp, PoseBlock &b) {
    const double xs[2] = {(0.0 - c.img_w / 2.0) / c.fx, ((c.img_w - 1) - c.img_w / 2.0) / c.fx};
    const double ys[2] = {(0.0 - c.img_h / 2.0) / c.fy, ((c.img_h - 1) - c.img_h / 2.0) / c.fy};
    const double z = c.depth_max;
    double lo_x = b.trans[0], hi_x = b.trans[0], lo_z = b.trans[2], hi_z = b.trans[2];
    for (int i = 0; i < 4; ++i) {
        const double x = z * xs[i & 1], y = z * ys[i >> 1];
        const double wx = b.rot[0] * x + b.rot[1] * y + b.rot[2] * z + b.trans[0];
        const double wz = b.rot[6] * x + b.rot[7] * y + b.rot[8] * z + b.trans[2];
        lo_x = std::min(lo_x, wx); hi_x = std::max(hi_x, wx);
        lo_z = std::min(lo_z, wz); hi_z = std::max(hi_z, wz);
    }
    const auto cell = [&](double w, double lb) { return clamp_cell(std::floor((w - lb) / c.pixel2meter)); };
    b.box[0] = std::max(cell(lo_z, mp.lower[2]) - 2, -mp.map_h);
    b.box[1] = std::min(cell(hi_z, mp.lower[2]) + 2, mp.map_h - 1);
    b.box[2] = std::max(cell(lo_x, mp.lower[0]) - 2, -mp.map_w);
    b.box[3] = std::min(cell(hi_x, mp.lower[0]) + 2, mp.map_w - 1);
    if (b.box[0] > b.box[1] || b.box[2] > b.box[3]) { b.box[0] = b.box[2] = 0; b.box[1] = b.box[3] = -1; }
}

extern "C" {

const char *ssmap_last_error(void) { return g_error.c_str(); }
const char *ssmap_version(void) { return "ssmap 0.1 (gfx950)"; }

void ssmap_destroy(ssmap_engine *eng) {
    if (!eng) return;
    View &v = eng->view;
    hipFree(eng->d_envs); hipFree(eng->d_poses); hipFree(eng->d_hdr);
    hipFree(eng->d_step); hipFree(v.quad);
    for (int i = 0; i < 2; ++i) { hipFree(v.overlay[i]); hipFree(v.overlay_count[i]); }
    hipFree(eng->d_lut); hipFree(eng->d_pool_ids); hipFree(eng->d_pool_count); hipFree(eng->d_query);
    hipFree(eng->d_disk); hipFree(eng->d_palette); hipFree(eng->d_ego);
    hipFree(eng->d_nb); hipFree(eng->d_nb_probs);
    hipFree(v.path_cells); hipFree(v.path_count);
    for (hipEvent_t ev : eng->prof_events) hipEventDestroy(ev);
    for (int i = 0; i < kPoseRing; ++i) {
        if (eng->h_ring[i]) hipHostFree(eng->h_ring[i]);
        if (eng->pose_done[i]) hipEventDestroy(eng->pose_done[i]);
    }
    delete eng;
}

int ssmap_create(const ssmap_config *cfg, ssmap_engine **out) {
    if (!cfg || !out) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    const ssmap_config &c = *cfg;
    if (c.n_envs < 1 || c.img_h < 1 || c.img_w < 1)
        return fail(SSMAP_ERR_INVALID_ARG, "n_envs >= 1, img_h >= 1 and img_w >= 1 are required");
    if ((int64_t)c.img_h * c.img_w >= (1 << 29))
        return fail(SSMAP_ERR_INVALID_ARG, "frame too large");
    if (c.rule < SSMAP_RULE_DIRECT || c.rule > SSMAP_RULE_STUBBORN)
        return fail(SSMAP_ERR_INVALID_ARG, "Invalid prob_aggregate");
    const bool prob_rule = c.rule >= SSMAP_RULE_AVG_PROB && c.rule <= SSMAP_RULE_LOG_ODDS;
    // the class tensor lives in LDS as [C][257] floats next to 2 KiB of bookkeeping
    static_assert(SSMAP_MAX_CLASSES == (160 * 1024 - 4096) / (257 * 4), "the header's limit is the LDS limit");
    if (c.n_classes < 0 || c.n_classes > SSMAP_MAX_CLASSES || (prob_rule && c.n_classes < 1))
        return fail(SSMAP_ERR_UNSUPPORTED, "n_classes must be in [0, 155] (at least 1 for the probability rules)");
    if (c.max_map_h < 1 || c.max_map_w < 1 || c.max_map_h > SSMAP_MAX_MAP_DIM || c.max_map_w > SSMAP_MAX_MAP_DIM)
        return fail(SSMAP_ERR_INVALID_ARG, "map dimensions must be in [1, 8192]");
    if ((c.rule == SSMAP_RULE_HV || c.rule == SSMAP_RULE_PROB_HV) && !c.use_height_map)
        return fail(SSMAP_ERR_INVALID_ARG, "Invalid prob_aggregate");
    if (c.rule == SSMAP_RULE_LOG_ODDS && c.use_height_map && c.n_schema != 9)
        return fail(SSMAP_ERR_INVALID_ARG, "Invalid num_semantic_classes for height_map_log_odds aggregation");
    if (!(c.pixel2meter > 0) || !(c.fx > 0) || !(c.fy > 0))
        return fail(SSMAP_ERR_INVALID_ARG, "pixel2meter, fx, fy must be positive");
    if (c.fx < 1e-3 || c.fx > 1e9 || c.fy < 1e-3 || c.fy > 1e9 || c.pixel2meter < 1e-6 || c.pixel2meter > 1e6)
        return fail(SSMAP_ERR_INVALID_ARG, "fx, fy must lie in [1e-3, 1e9] and pixel2meter in [1e-6, 1e6]");

    ssmap_engine *eng = new (std::nothrow) ssmap_engine();
    if (!eng) return fail(SSMAP_ERR_INVALID_ARG, "out of host memory");
    eng->cfg = c;
    if (hipError_t err = hipGetDevice(&eng->device); err != hipSuccess) {
        delete eng;
        return fail(SSMAP_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(err));
    }
    eng->flip_cam = qexp_half(M_PI, 0.0, 0.0);
    eng->flip_agent = qexp_half(0.0, M_PI, 0.0);
    eng->envs.assign(c.n_envs, EnvState{});
    eng->h_hdr.assign(c.n_envs, FrameHeader{});

    View &v = eng->view;
    v.E = c.n_envs; v.H = c.img_h; v.W = c.img_w; v.N = c.img_h * c.img_w; v.C = c.n_classes;
    v.n_schema = c.n_schema; v.rule = c.rule; v.use_height = c.use_height_map;
    v.depth_min = c.depth_min; v.depth_max = c.depth_max;
    v.fx = c.fx; v.fy = c.fy; v.p2m = c.pixel2meter;
    v.inv_fx = 1.0 / c.fx; v.inv_fy = 1.0 / c.fy; v.inv_p2m = 1.0 / c.pixel2meter;
    v.inv_fx_wide = v.inv_fx * (1.0 + 0x1p-20); v.inv_fy_wide = v.inv_fy * (1.0 + 0x1p-20);
    const size_t EN = (size_t)v.E * v.N;
    int rc = SSMAP_OK;
#define TRY_ALLOC(call) do { rc = (call); if (rc != SSMAP_OK) { ssmap_destroy(eng); return rc; } } while (0)
    TRY_ALLOC(dev_alloc(&eng->d_envs, (size_t)v.E));
    TRY_ALLOC(dev_alloc(&eng->d_poses, (size_t)v.E));
    TRY_ALLOC(dev_alloc(&eng->d_hdr, (size_t)v.E));
    v.geom_blocks = (v.N + kGeomPixPerBlock - 1) / kGeomPixPerBlock;
    v.rec_packed = v.N <= (1 << kPackPixBits) ? 1 : 0;
    v.wprob_stride = v.C;   // padding the rows (to 16 bytes, to whole cache lines) measured slower
    if (const char *x = getenv("SSMAP_WPROB_STRIDE")) v.wprob_stride = atoi(x) > v.C ? atoi(x) : v.C;
    // Class counts with a direct variant of the register streaming kernel: it updates the owners'
    // map cells itself; the others go through staged rows and the update kernel.
    v.direct_update = (prob_rule && direct_class_count(v.C) && (int64_t)v.C * v.N * 4 < (1ll << 31) &&
                       c.n_schema + 3 <= 64) ? 1 : 0;
    if (const char *x = getenv("SSMAP_DIRECT")) v.direct_update = v.direct_update && atoi(x) != 0;
    // (the generic kernel has no direct variant: forcing it also selects staged rows + the update kernel)
    if (const char *x = getenv("SSMAP_GENERIC")) { v.force_generic = atoi(x) != 0; if (v.force_generic) v.direct_update = 0; }
    if (const char *x = getenv("SSMAP_EXP")) v.exp = atoi(x);
    if (const char *x = getenv("SSMAP_EXP_LDS")) v.exp_lds = atoi(x);
    v.local_stride = 4ll * c.max_map_h * c.max_map_w;
    eng->step_bytes = carve_step_scratch(c, v, nullptr);
    TRY_ALLOC(dev_alloc(&eng->d_step, eng->step_bytes));
    carve_step_scratch(c, v, eng->d_step);
    // goal_decay's local maps start out zero and every step leaves them zero (k_found)
    if (v.local_ids) TRY_ALLOC(dev_zero(v.local_ids, (size_t)v.E * v.local_stride));
    // this device's address of the file-scope scratch array (ssmap_step_noalloc)
    {
        void *sym = nullptr;
        hipError_t err = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
        if (err != hipSuccess || !sym) {
            ssmap_destroy(eng);
            return fail(SSMAP_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(err));
        }
        eng->d_static = static_cast<unsigned char *>(sym);
    }
    v.quad_stride = 4ll * c.max_map_h * c.max_map_w;
    TRY_ALLOC(dev_alloc(&v.quad, (size_t)v.E * v.quad_stride));
    TRY_ALLOC(dev_zero(v.quad, sizeof(int32_t) * v.E * v.quad_stride));
    for (int i = 0; i < 2; ++i) {
        if (c.rule == SSMAP_RULE_HV || c.rule == SSMAP_RULE_PROB_HV) TRY_ALLOC(dev_alloc(&v.overlay[i], EN));
        TRY_ALLOC(dev_alloc(&v.overlay_count[i], (size_t)v.E));
        TRY_ALLOC(dev_zero(v.overlay_count[i], sizeof(uint32_t) * v.E));
    }
    TRY_ALLOC(dev_alloc(&eng->d_lut, (size_t)32 * kClassWords));
    TRY_ALLOC(dev_alloc(&eng->d_pool_ids, (size_t)6 * SSMAP_MAX_POOL));
    TRY_ALLOC(dev_alloc(&eng->d_pool_count, (size_t)6));
    TRY_ALLOC(dev_alloc(&eng->d_query, (size_t)2 * v.E));
    eng->h_query.assign((size_t)2 * v.E, 0);
    v.path_cap = (int64_t)c.max_map_h * c.max_map_w;
    TRY_ALLOC(dev_alloc(&v.path_cells, (size_t)v.E * v.path_cap));
    TRY_ALLOC(dev_alloc(&v.path_count, (size_t)v.E));
    TRY_ALLOC(dev_zero(v.path_count, sizeof(uint32_t) * v.E));
    eng->path_n.assign(c.n_envs, 0u);
    TRY_ALLOC(dev_alloc(&eng->d_nb, (size_t)6));
    TRY_ALLOC(dev_zero(eng->d_nb, 6 * sizeof(NbModel)));
    v.nb = eng->d_nb;
    TRY_ALLOC(dev_zero(eng->d_pool_count, 6 * sizeof(int32_t)));
    TRY_ALLOC(dev_zero(eng->d_lut, 32 * kClassWords * sizeof(int32_t)));
#undef TRY_ALLOC
    eng->h_poses.assign(c.n_envs, PoseBlock{});
    eng->last_poses.assign(c.n_envs, PoseBlock{});
    eng->has_frame.assign(c.n_envs, 0);
    for (int i = 0; i < kPoseRing && c.n_envs > kPosePack; ++i) {
        if (hipHostMalloc(reinterpret_cast<void **>(&eng->h_ring[i]), sizeof(PoseBlock) * v.E) != hipSuccess ||
            hipEventCreateWithFlags(&eng->pose_done[i], hipEventDisableTiming) != hipSuccess) {
            ssmap_destroy(eng);
            return fail(SSMAP_ERR_HIP, "pinned pose staging allocation failed");
        }
    }
    v.envs = eng->d_envs; v.poses = eng->d_poses; v.hdr = eng->d_hdr;
    v.lut = eng->d_lut; v.pool_ids = eng->d_pool_ids; v.pool_count = eng->d_pool_count;
    v.cur_overlay = 0;
    *out = eng;
    return SSMAP_OK;
}

int ssmap_set_schema(ssmap_engine *eng, const int32_t *lut, int32_t n, const int32_t *pool_ids,
                     const int32_t *pool_count) {
    if (!eng || !lut || n < 1 || n > SSMAP_MAX_CLASSES) return fail(SSMAP_ERR_INVALID_ARG, "bad schema table");
    if (eng->cfg.n_classes && n != eng->cfg.n_classes) return fail(SSMAP_ERR_INVALID_ARG, "schema table length != n_classes");
    for (int i = 0; i < n; ++i)
        if (lut[i] < 0 || lut[i] >= eng->cfg.n_schema + 3) return fail(SSMAP_ERR_INVALID_ARG, "schema id out of range");
    HIP_TRY(hipMemcpy(eng->d_lut, lut, n * sizeof(int32_t), hipMemcpyHostToDevice));
    int32_t ids[6 * SSMAP_MAX_POOL] = {};
    int32_t cnt[6] = {};
    if (pool_ids && pool_count) {
        for (int g = 0; g < 6; ++g) {
            if (pool_count[g] < 0 || pool_count[g] > SSMAP_MAX_POOL) return fail(SSMAP_ERR_INVALID_ARG, "pool too long");
            for (int k = 0; k < pool_count[g]; ++k)
                if (pool_ids[g * SSMAP_MAX_POOL + k] < 0 || pool_ids[g * SSMAP_MAX_POOL + k] >= n)
                    return fail(SSMAP_ERR_INVALID_ARG, "pool class out of range");
            cnt[g] = pool_count[g];
            memcpy(ids + g * SSMAP_MAX_POOL, pool_ids + g * SSMAP_MAX_POOL, sizeof(int32_t) * pool_count[g]);
        }
    } else {
        // no lists: every class the table maps to the goal id, ascending
        for (int c = 0; c < n; ++c) {
            const int g = lut[c] - 3;
            if (g < 0 || g >= 6) continue;
            if (cnt[g] >= SSMAP_MAX_POOL) return fail(SSMAP_ERR_INVALID_ARG, "more than 8 classes map to one goal id: pass pool lists");
            ids[g * SSMAP_MAX_POOL + cnt[g]++] = c;
        }
    }
    HIP_TRY(hipMemcpy(eng->d_pool_ids, ids, sizeof(ids), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(eng->d_pool_count, cnt, sizeof(cnt), hipMemcpyHostToDevice));
    {
        View &v = eng->view;
        memset(v.pool_pack, 0, sizeof(v.pool_pack));
        memset(v.pool_mask, 0, sizeof(v.pool_mask));
        for (int g = 0; g < 6; ++g)
            for (int k = 0; k < cnt[g]; ++k) {
                const int c = ids[g * SSMAP_MAX_POOL + k];
                v.pool_mask[c >> 5] |= 1u << (c & 31);
            }
        auto below = [&](int c) {   // pooled classes below c = c's staging slot (the kernel walks upwards)
            int r = 0;
            for (int d = 0; d < c; ++d) r += (v.pool_mask[d >> 5] >> (d & 31)) & 1u;
            return r;
        };
        v.pool_slots = below(n);
        for (int g = 0; g < 6; ++g) {
            v.pool_pack[12] |= (uint32_t)cnt[g] << (4 * g);
            for (int k = 0; k < cnt[g]; ++k) {
                const int at = g * SSMAP_MAX_POOL + k;
                v.pool_pack[at >> 2] |= (uint32_t)below(ids[at]) << (8 * (at & 3));
            }
        }
    }
    // The pooled confidence of a pixel is its largest probability itself when every goal id is
    // reached through exactly one class and that class is the goal's whole list.
    eng->view.pool_trivial = 1;
    for (int g = 0; g < 6; ++g) {
        int mapped = 0, last = -1;
        for (int c = 0; c < n; ++c)
            if (lut[c] == g + 3) { ++mapped; last = c; }
        if (mapped == 0) continue;   // no pixel can carry this id
        if (mapped != 1 || cnt[g] != 1 || ids[g * SSMAP_MAX_POOL] != last) eng->view.pool_trivial = 0;
    }
    eng->have_schema = true;
    return SSMAP_OK;
}

int ssmap_set_palette(ssmap_engine *eng, const uint8_t *rgb, int32_t n) {
    if (!eng || !rgb || n < eng->cfg.n_schema + 2) return fail(SSMAP_ERR_INVALID_ARG, "palette too short");
    if (eng->d_palette) { hipFree(eng->d_palette); eng->d_palette = nullptr; }
    int rc = dev_alloc(&eng->d_palette, (size_t)3 * n);
    if (rc != SSMAP_OK) return rc;
    HIP_TRY(hipMemcpy(eng->d_palette, rgb, 3 * n, hipMemcpyHostToDevice));
    eng->view.palette = eng->d_palette;
    eng->view.palette_rows = n;
    eng->palette_rows = n;
    eng->have_palette = true;
    return SSMAP_OK;
}

int ssmap_set_nb_model(ssmap_engine *eng, int32_t goal_sem_id, int32_t n_classes, const int32_t *classes,
                       const double *class_log_prior, const double *norm_const, const double *theta,
                       const double *var) {
    if (!eng || !classes || !class_log_prior || !norm_const || !theta || !var)
        return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    if (goal_sem_id < 3 || goal_sem_id > 8) return fail(SSMAP_ERR_INVALID_ARG, "goal_sem_id must be in [3, 8]");
    if (n_classes < 1 || n_classes > 2) return fail(SSMAP_ERR_UNSUPPORTED, "naive-Bayes models of one or two classes");
    NbModel m{};
    m.n_classes = n_classes;
    for (int k = 0; k < n_classes; ++k) {
        m.classes[k] = classes[k];
        m.log_prior[k] = class_log_prior[k];
        m.konst[k] = norm_const[k];
        for (int j = 0; j < 4; ++j) {
            if (!(var[4 * k + j] > 0)) return fail(SSMAP_ERR_INVALID_ARG, "variances must be positive");
            m.theta[k][j] = theta[4 * k + j];
            m.var[k][j] = var[4 * k + j];
        }
    }
    HIP_TRY(hipMemcpy(eng->d_nb + (goal_sem_id - 3), &m, sizeof(NbModel), hipMemcpyHostToDevice));
    eng->nb_set[goal_sem_id - 3] = true;
    return SSMAP_OK;
}

int ssmap_set_params(ssmap_engine *eng, const ssmap_params *p) {
    if (!eng || !p) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    const int radius = (int)(p->found_dist_m / eng->cfg.pixel2meter);
    if (radius < 0 || radius > 4096) return fail(SSMAP_ERR_INVALID_ARG, "found_dist_m out of range");
    if (!eng->have_params || radius != eng->view.disk_radius) {
        std::vector<int32_t> hw = disk_halfwidths(radius);
        if (eng->d_disk) { hipFree(eng->d_disk); eng->d_disk = nullptr; }
        int rc = dev_alloc(&eng->d_disk, hw.size());
        if (rc != SSMAP_OK) return rc;
        HIP_TRY(hipMemcpy(eng->d_disk, hw.data(), hw.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        eng->view.disk_hw = eng->d_disk;
        eng->view.disk_radius = radius;
    }
    eng->view.prm = *p;
    eng->have_params = true;
    return SSMAP_OK;
}

int ssmap_bind_env(ssmap_engine *eng, int32_t e, const ssmap_env_maps *maps) {
    if (!eng || !maps || e < 0 || e >= eng->cfg.n_envs) return fail(SSMAP_ERR_INVALID_ARG, "bad env index");
    const ssmap_config &c = eng->cfg;
    if (maps->map_h < 1 || maps->map_w < 1 || maps->map_h > c.max_map_h || maps->map_w > c.max_map_w)
        return fail(SSMAP_ERR_INVALID_ARG, "map larger than max_map_h x max_map_w");
    if (!maps->semantic_map || !maps->semantic_map_color || !maps->height_map)
        return fail(SSMAP_ERR_INVALID_ARG, "semantic_map, semantic_map_color and height_map are required");
    const int r = c.rule;
    if (r != SSMAP_RULE_DIRECT && (!maps->gt_map || !maps->unc_map))
        return fail(SSMAP_ERR_INVALID_ARG, "gt_map and unc_map are required in probabilistic mode");
    if (r >= SSMAP_RULE_AVG_PROB && r <= SSMAP_RULE_LOG_ODDS && !maps->prob_map)
        return fail(SSMAP_ERR_INVALID_ARG, "prob_map is required for this rule");
    if (maps->prob_map && reinterpret_cast<uintptr_t>(maps->prob_map) % 16)
        return fail(SSMAP_ERR_INVALID_ARG, "prob_map must be 16-byte aligned (cell rows move as vectors)");
    if (r >= SSMAP_RULE_AVG_PROB && r <= SSMAP_RULE_W2_AVG_PROB && !maps->weight_map)
        return fail(SSMAP_ERR_INVALID_ARG, "weight_map is required for this rule");
    if (r == SSMAP_RULE_GOAL_DECAY && !maps->goal_map)
        return fail(SSMAP_ERR_INVALID_ARG, "goal_map is required for this rule");
    if ((r == SSMAP_RULE_HV || r == SSMAP_RULE_PROB_HV) && (!maps->background_map || !maps->hits || !maps->views))
        return fail(SSMAP_ERR_INVALID_ARG, "background_map, hits and views are required for this rule");
    if (r == SSMAP_RULE_STUBBORN &&
        (!maps->total_view || !maps->cumulative_conf || !maps->max_conf || !maps->max_other_conf))
        return fail(SSMAP_ERR_INVALID_ARG, "total_view, cumulative_conf, max_conf and max_other_conf are required for this rule");
    EnvState st{};
    st.maps = *maps;
    st.bound = 1;
    eng->envs[e] = st;
    HIP_TRY(hipMemcpy(eng->d_envs + e, &st, sizeof(EnvState), hipMemcpyHostToDevice));
    const uint32_t zero = 0;
    for (int i = 0; i < 2; ++i)
        HIP_TRY(hipMemcpy(eng->view.overlay_count[i] + e, &zero, sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(eng->view.path_count + e, &zero, sizeof(uint32_t), hipMemcpyHostToDevice));
    eng->path_n[e] = 0;
    // a new episode: the env has no robot cell and no goal until its first frame
    eng->last_poses[e] = PoseBlock{};
    eng->has_frame[e] = 0;
    return SSMAP_OK;
}

int ssmap_path_append(ssmap_engine *eng, int32_t e, const int32_t *cells, int32_t n, void *stream_v) {
    if (!eng || e < 0 || e >= eng->cfg.n_envs || n < 0 || (n > 0 && !cells))
        return fail(SSMAP_ERR_INVALID_ARG, "bad env index or cell list");
    if (!eng->envs[e].bound) return fail(SSMAP_ERR_NOT_READY, "the env has no maps bound (ssmap_bind_env)");
    if (n == 0) return SSMAP_OK;
    const ssmap_env_maps &mp = eng->envs[e].maps;
    const int64_t n_cells = (int64_t)mp.map_h * mp.map_w;
    for (int i = 0; i < n; ++i)
        if (cells[i] < 0 || cells[i] >= n_cells) return fail(SSMAP_ERR_INVALID_ARG, "path cell outside the map");
    if ((int64_t)eng->path_n[e] + n > eng->view.path_cap)
        return fail(SSMAP_ERR_INVALID_ARG, "more path cells than the map has cells (append only new cells)");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    HIP_TRY(hipMemcpyAsync(eng->view.path_cells + (int64_t)e * eng->view.path_cap + eng->path_n[e], cells,
                           sizeof(int32_t) * n, hipMemcpyHostToDevice, stream));
    eng->path_n[e] += (uint32_t)n;
    HIP_TRY(hipMemcpyAsync(eng->view.path_count + e, &eng->path_n[e], sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    // both sources are pageable host memory the caller or a later append may change: finish the copies
    HIP_TRY(hipStreamSynchronize(stream));
    return SSMAP_OK;
}

int ssmap_path_paint(ssmap_engine *eng, void *stream_v) {
    if (!eng) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    if (!eng->have_palette) return fail(SSMAP_ERR_NOT_READY, "ssmap_set_palette was not called");
    uint32_t most = 0;
    for (uint32_t c : eng->path_n) most = std::max(most, c);
    launch_path_paint(eng->view, most, reinterpret_cast<hipStream_t>(stream_v));
    HIP_TRY(hipGetLastError());
    return SSMAP_OK;
}

// One axis of one of the two resizes, the way OpenCV's resize() sets it up (4.x resize.cpp):
// inv = dst / src, scale = 1 / inv; area table (computeResizeAreaTab) when shrinking, bilinear
// area-mode weights otherwise.
static bool build_ego_axis(EgoTables &t, EgoAxis &ax, int ssize, int dsize, int mode, bool horizontal, int &pool) {
    const double inv = (double)dsize / ssize, scale = 1.0 / inv;
    ax.ssize = ssize;
    ax.scale = scale;
    for (int d = 0; d < dsize; ++d) {
        ax.first[d] = pool;
        ax.count[d] = 0;
        if (mode == kEgoBox) {
            // integer factor: the destination pixel is the mean of `factor` whole source pixels
            const int factor = ssize / dsize;
            for (int k = 0; k < factor; ++k) {
                if (pool >= kEgoTapPool || ax.count[d] >= kEgoTapLimit) return false;
                t.tap_src[pool] = d * factor + k;
                t.tap_w[pool] = 1.0f;
                ++pool;
                ++ax.count[d];
            }
        } else if (mode == kEgoArea) {
            const double fsx1 = d * scale, fsx2 = fsx1 + scale;
            const double cell = std::min(scale, ssize - fsx1);
            int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
            sx2 = std::min(sx2, ssize - 1);
            sx1 = std::min(sx1, sx2);
            auto push = [&](int si, double w) {
                if (pool >= kEgoTapPool) return false;
                if (ax.count[d] >= kEgoTapLimit) return false;
                t.tap_src[pool] = si;
                t.tap_w[pool] = (float)w;
                ++pool;
                ++ax.count[d];
                return true;
            };
            if (sx1 - fsx1 > 1e-3 && !push(sx1 - 1, (sx1 - fsx1) / cell)) return false;
            for (int sx = sx1; sx < sx2; ++sx)
                if (!push(sx, 1.0 / cell)) return false;
            if (fsx2 - sx2 > 1e-3 && !push(sx2, std::min(std::min(fsx2 - sx2, 1.0), cell) / cell)) return false;
        } else {
            int sx = (int)std::floor(d * scale);
            float fx = (float)((d + 1) - (sx + 1) * inv);
            fx = fx <= 0 ? 0.f : fx - std::floor(fx);
            if (horizontal && sx < 0) { fx = 0.f; sx = 0; }
            if (horizontal && sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
            ax.lin_src[d] = sx;
            auto sat_short = [](float v) { const long r = std::lrintf(v); return (int)std::max(-32768l, std::min(32767l, r)); };
            ax.lin_w0[d] = sat_short((1.f - fx) * 2048.f);
            ax.lin_w1[d] = sat_short(fx * 2048.f);
        }
    }
    return true;
}

int ssmap_set_egomap(ssmap_engine *eng, const ssmap_egomap_config *cfg) {
    if (!eng || !cfg) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    const ssmap_egomap_config &c = *cfg;
    if (c.small_crop_px < 1 || c.large_crop_px < c.small_crop_px || c.offset_for_cut < 0 ||
        c.large_crop_px + c.offset_for_cut > 32767)
        return fail(SSMAP_ERR_INVALID_ARG, "ego-map crop sizes out of range");
    if (c.out_w < 1 || c.out_h < 1 || c.out_w > kEgoMaxOut || c.out_h > kEgoMaxOut)
        return fail(SSMAP_ERR_INVALID_ARG, "egomap_size must be within 256 x 256");
    if (c.agent_radius_px < 0 || c.agent_radius_px > kEgoMaxRadius || c.n_aux < 0 || c.n_aux > SSMAP_MAX_AUX)
        return fail(SSMAP_ERR_INVALID_ARG, "agent radius or auxiliary bins out of range");
    eng->h_ego.assign(1, EgoTables{});
    EgoTables &t = eng->h_ego[0];
    t.side = c.large_crop_px + c.offset_for_cut;
    t.origin = (int)(float)(t.side / 2.0);
    t.out_w = c.out_w; t.out_h = c.out_h;
    t.agent_radius = c.agent_radius_px;
    t.aux_half = c.aux_bin_px / 2;
    t.n_aux = c.n_aux;
    const std::vector<int32_t> hw = disk_halfwidths(c.agent_radius_px);
    for (int i = 0; i <= c.agent_radius_px; ++i) t.agent_hw[i] = hw[i];
    memcpy(t.aux_points, c.aux_points, sizeof(int32_t) * 2 * c.n_aux);
    t.aux_reach = 0;
    for (int i = 0; i < 2 * c.n_aux; ++i) t.aux_reach = std::max(t.aux_reach, std::abs(c.aux_points[i]) + t.aux_half + 1);
    memcpy(t.active_goal_color, c.active_goal_color, 3);
    memcpy(t.arrow_color, c.arrow_color, 3);
    int pool = 0;
    const int crop[2] = {c.small_crop_px, c.large_crop_px};
    for (int w = 0; w < 2; ++w) {
        t.crop_off[w] = (int)(t.side / 2.0 - crop[w] / 2.0);
        const double sx = 1.0 / ((double)c.out_w / crop[w]), sy = 1.0 / ((double)c.out_h / crop[w]);
        if (sx >= 1 && sy >= 1) {
            const double eps = 2.220446049250313e-16;
            t.mode[w] = kEgoArea;
            t.box_scale[w] = 1.0f;
            t.box_area[w] = 1;
            if (std::fabs(sx - (int)sx) < eps && std::fabs(sy - (int)sy) < eps) {
                // OpenCV's integer-factor area path: sum of the block times 1 / area in float32
                t.mode[w] = kEgoBox;
                t.box_area[w] = (int)sx * (int)sy;
                t.box_scale[w] = 1.0f / (float)t.box_area[w];
            }
        } else {
            t.mode[w] = kEgoLinear;
        }
        if (!build_ego_axis(t, t.axis[w][0], crop[w], c.out_w, t.mode[w], true, pool) ||
            !build_ego_axis(t, t.axis[w][1], crop[w], c.out_h, t.mode[w], false, pool)) {
            eng->h_ego.clear();
            return fail(SSMAP_ERR_UNSUPPORTED, "ego-map crops too large for the resize tables");
        }
    }
    t.max_taps = 0;
    for (int w = 0; w < 2; ++w)
        for (int a = 0; a < 2; ++a)
            if (t.mode[w] != kEgoLinear)
                for (int d = 0; d < (a ? c.out_h : c.out_w); ++d) t.max_taps = std::max(t.max_taps, t.axis[w][a].count[d]);
    if (!eng->d_ego) {
        int rc = dev_alloc(&eng->d_ego, (size_t)1);
        if (rc != SSMAP_OK) return rc;
    }
    HIP_TRY(hipMemcpy(eng->d_ego, &t, sizeof(EgoTables), hipMemcpyHostToDevice));
    return SSMAP_OK;
}

int ssmap_ego_maps(ssmap_engine *eng, const double *agent_pose, const int32_t *goal_sem_id,
                   const uint8_t *aux_color, uint8_t *ego_small, uint8_t *ego_large,
                   float *unc_small, float *unc_large, void *stream_v) {
    if (!eng || !agent_pose || !goal_sem_id || !ego_small || !ego_large)
        return fail(SSMAP_ERR_INVALID_ARG, "agent_pose, goal_sem_id, ego_small and ego_large are required");
    if (eng->h_ego.empty()) return fail(SSMAP_ERR_NOT_READY, "ssmap_set_egomap was not called");
    if (!eng->have_palette) return fail(SSMAP_ERR_NOT_READY, "ssmap_set_palette was not called");
    for (const EnvState &s : eng->envs)
        if (!s.bound) return fail(SSMAP_ERR_NOT_READY, "an env has no maps bound (ssmap_bind_env)");
    const ssmap_config &c = eng->cfg;
    const EgoTables &t = eng->h_ego[0];
    eng->h_ego_poses.assign(c.n_envs, EgoPose{});
    for (int e = 0; e < c.n_envs; ++e) {
        const double *ag = agent_pose + 7 * e;
        const ssmap_env_maps &mp = eng->envs[e].maps;
        EgoPose &p = eng->h_ego_poses[e];
        // heading = R(agent rotation) (0,0,1); phi = atan2(heading z, heading x) (mapper.py:885-890),
        // where the agent rotation is the pose quaternion times a half turn about y, as update_map
        // passes it on (mapper.py:291-293, 840-847)
        double R[9];
        qmatrix(qmul(Quat{ag[3], ag[4], ag[5], ag[6]}, eng->flip_agent), R);
        const double phi = std::atan2(R[8], R[2]);
        // getRotationMatrix2D((side/2, side/2), deg(phi) + 90, 1) and warpAffine's inversion
        double angle = phi * (180.0 / M_PI) + 90;
        angle *= M_PI / 180;
        const double alpha = std::cos(angle), beta = std::sin(angle);
        const double cx = (double)(float)(t.side / 2.0), cy = cx;
        double m[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
        double D = m[0] * m[4] - m[1] * m[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = m[4] * D, A22 = m[0] * D;
        m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22;
        const double b1 = -m[0] * m[2] - m[1] * m[5], b2 = -m[3] * m[2] - m[4] * m[5];
        m[2] = b1; m[5] = b2;
        memcpy(p.m, m, sizeof(m));
        // crop window of side t.side centred on the agent's cell (mapper.py:891, 904-911, 1000-1004)
        const double pr = std::floor((ag[2] - mp.lower[2]) / c.pixel2meter);
        const double pc = std::floor((ag[0] - mp.lower[0]) / c.pixel2meter);
        p.row0 = clamp_cell(pr - t.side / 2.0);
        p.col0 = clamp_cell(pc - t.side / 2.0);
        p.goal_id = goal_sem_id[e];
        p.has_aux = aux_color ? 1 : 0;
        if (aux_color) memcpy(p.aux_color, aux_color + (size_t)e * t.n_aux * 3, (size_t)t.n_aux * 3);
    }
    launch_ego_maps(eng->view, eng->h_ego_poses.data(), eng->d_ego, t, ego_small, ego_large, unc_small, unc_large,
                    reinterpret_cast<hipStream_t>(stream_v));
    HIP_TRY(hipGetLastError());
    return SSMAP_OK;
}

static int check_ready(ssmap_engine *eng) {
    if (!eng->have_params) return fail(SSMAP_ERR_NOT_READY, "ssmap_set_params was not called");
    if (!eng->have_palette) return fail(SSMAP_ERR_NOT_READY, "ssmap_set_palette was not called");
    for (const EnvState &s : eng->envs)
        if (!s.bound) return fail(SSMAP_ERR_NOT_READY, "an env has no maps bound (ssmap_bind_env)");
    return SSMAP_OK;
}

// One step; `static_home` says where the per-step scratch lives: the engine's own block
// (ssmap_step) or this device's copy of the file-scope array (ssmap_step_noalloc).
static int step_impl(ssmap_engine *eng, const ssmap_frames *f, void *stream_v, bool static_home) {
    if (!eng || !f) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    int rc = check_ready(eng);
    if (rc != SSMAP_OK) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const ssmap_config &c = eng->cfg;
    View &v = eng->view;
    const int r = c.rule;
    const bool prob_rule = r >= SSMAP_RULE_AVG_PROB && r <= SSMAP_RULE_LOG_ODDS;
    if (!f->depth || !f->agent_pose || !f->camera_pose || !f->goal_sem_id)
        return fail(SSMAP_ERR_INVALID_ARG, "depth, agent_pose, camera_pose and goal_sem_id are required");
    if (prob_rule && !f->logits && !f->probs)
        return fail(SSMAP_ERR_INVALID_ARG, "this rule needs logits or semantic_prob");
    if ((prob_rule || f->logits) && !eng->have_schema)
        return fail(SSMAP_ERR_NOT_READY, "ssmap_set_schema was not called");
    if (f->logits && !(f->temperature > 0)) return fail(SSMAP_ERR_INVALID_ARG, "temperature must be positive");
    if (f->logits && c.n_classes < 1) return fail(SSMAP_ERR_INVALID_ARG, "engine was created without classes");
    if (!prob_rule) {
        if (!f->semantic) return fail(SSMAP_ERR_INVALID_ARG, "semantic is required for this rule");
        const bool needs_unc = (r == SSMAP_RULE_PROB_HV) || ((r == SSMAP_RULE_ARGMAX || r == SSMAP_RULE_STUBBORN || r == SSMAP_RULE_GOAL_DECAY) && v.prm.use_filter);
        if (needs_unc && !f->semantic_unc) return fail(SSMAP_ERR_INVALID_ARG, "semantic_unc is required for this rule");
        const bool needs_maxp = (r == SSMAP_RULE_PROB_HV) || (r == SSMAP_RULE_HV && v.prm.use_filter);
        if (needs_maxp && !f->semantic_max_prob) return fail(SSMAP_ERR_INVALID_ARG, "semantic_max_prob is required for this rule");
    }
    if (r != SSMAP_RULE_DIRECT && !f->semantic_gt) return fail(SSMAP_ERR_INVALID_ARG, "semantic_gt is required in probabilistic mode");
    const bool goal_stats = r == SSMAP_RULE_STUBBORN || (r == SSMAP_RULE_ARGMAX && v.prm.nb_stats);
    v.nb_probs = nullptr;
    if (goal_stats) {
        if (c.n_classes < 1 || !eng->have_schema)
            return fail(SSMAP_ERR_NOT_READY, "the goal statistics need the class count and ssmap_set_schema (goal class lists)");
        if (!f->probs && !f->logits) return fail(SSMAP_ERR_INVALID_ARG, "semantic_prob (or logits) is required for the goal statistics");
        for (int e = 0; e < c.n_envs; ++e) {
            if (f->skip && f->skip[e]) continue;   // no frame for this env: its slices are not read
            const int g = f->goal_sem_id[e];
            if (g < 3 || g > 8) return fail(SSMAP_ERR_INVALID_ARG, "goal id outside [3, 8]");
            if (r == SSMAP_RULE_STUBBORN && !v.prm.nb_stats && !eng->nb_set[g - 3])
                return fail(SSMAP_ERR_NOT_READY, "no naive-Bayes model for this goal (ssmap_set_nb_model)");
            const ssmap_env_maps &mp = eng->envs[e].maps;
            if (!mp.total_view || !mp.cumulative_conf || !mp.max_conf || !mp.max_other_conf)
                return fail(SSMAP_ERR_INVALID_ARG, "the goal statistics maps are not bound");
        }
        if (f->probs) {
            v.nb_probs = f->probs;
        } else {
            if (!eng->d_nb_probs) {   // first step fed with logits: room for one batch of probabilities
                if (static_home)
                    return fail(SSMAP_ERR_UNSUPPORTED, "ssmap_step_noalloc: goal statistics fed by logits need a "
                                "probability buffer that an earlier ssmap_step would have allocated; pass semantic_prob");
                rc = dev_alloc(&eng->d_nb_probs, (size_t)v.E * v.C * v.N);
                if (rc != SSMAP_OK) return rc;
            }
            v.nb_probs = eng->d_nb_probs;
        }
    }

    // a refused call changes nothing: the size check of the static home comes before any pose work
    if (static_home && eng->step_bytes > kStaticScratchBytes)
        return fail(SSMAP_ERR_UNSUPPORTED, "ssmap_step_noalloc: this engine's per-step scratch (" +
                    std::to_string(eng->step_bytes) + " bytes) exceeds the static array (" +
                    std::to_string(kStaticScratchBytes) + " bytes); use ssmap_step");

    // ---- pose preparation on the host (a few hundred flops per env)
    const bool packed = c.n_envs <= kPosePack;
    const int slot = eng->ring;
    PoseBlock *pb = eng->h_poses.data();
    if (!packed) {
        eng->ring = (eng->ring + 1) % kPoseRing;
        HIP_TRY(hipEventSynchronize(eng->pose_done[slot]));
        pb = eng->h_ring[slot];
    }
    for (int e = 0; e < c.n_envs; ++e) {
        const double *ag = f->agent_pose + 7 * e, *cam = f->camera_pose + 7 * e;
        const ssmap_env_maps &mp = eng->envs[e].maps;
        PoseBlock &b = pb[e];
        if (f->skip && f->skip[e]) {   // no frame for this env: its last pose stays on the device
            b = eng->last_poses[e];
            b.skip = 1;
            b.box[0] = 1; b.box[1] = 0; b.box[2] = 1; b.box[3] = 0;
            continue;
        }
        qmatrix(qmul(Quat{cam[3], cam[4], cam[5], cam[6]}, eng->flip_cam), b.rot);
        b.trans[0] = cam[0]; b.trans[1] = cam[1]; b.trans[2] = cam[2];
        b.h_cut = cam[1] + c.height_cutoff_m;
        b.agent[0] = ag[0]; b.agent[1] = ag[1]; b.agent[2] = ag[2];
        b.robot_row = clamp_cell(std::floor((ag[2] - mp.lower[2]) / c.pixel2meter));
        b.robot_col = clamp_cell(std::floor((ag[0] - mp.lower[0]) / c.pixel2meter));
        b.goal_id = f->goal_sem_id[e];
        b.skip = 0;
        predict_box(c, mp, b);
        eng->last_poses[e] = b;
        eng->has_frame[e] = 1;
    }

    if (!packed) {
        HIP_TRY(hipMemcpyAsync(eng->d_poses, pb, sizeof(PoseBlock) * c.n_envs, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(eng->pose_done[slot], stream));
    }

    // the view the kernels get: the engine's, with the per-step scratch at its home for this call
    View lv = v;
    if (static_home) carve_step_scratch(c, lv, eng->d_static);
    lv.local_clean = static_home ? 0 : 1;   // the shared static array holds whatever its last user left
    eng->last_list = lv.winners;
    hipEvent_t *marks = nullptr;
    if (eng->prof_capacity && eng->prof_steps < eng->prof_capacity)
        marks = eng->prof_events.data() + (size_t)kProfMarks * eng->prof_steps++;
#define MARK(i) do { if (marks) HIP_TRY(hipEventRecord(marks[i], stream)); } while (0)
    MARK(0);
    launch_geometry(lv, *f, packed ? pb : nullptr, stream);
    MARK(1);
    launch_winners(lv, *f, stream);
    MARK(2);
    if (prob_rule) {
        launch_stream(lv, *f, true, nullptr, stream);
        MARK(3);
        if (!lv.direct_update) launch_update_prob(lv, *f, stream);
    } else {
        if (f->logits) launch_stream(lv, *f, false, goal_stats ? eng->d_nb_probs : nullptr, stream);
        MARK(3);
        launch_label_rules(lv, *f, stream);
    }
    MARK(4);
    launch_found(lv, stream);
    MARK(5);
#undef MARK
    HIP_TRY(hipGetLastError());
    v.cur_overlay ^= 1;
    eng->stepped = true;
    return SSMAP_OK;
}

int ssmap_step(ssmap_engine *eng, const ssmap_frames *f, void *stream_v) {
    return step_impl(eng, f, stream_v, false);
}

int ssmap_step_noalloc(ssmap_engine *eng, const ssmap_frames *f, void *stream_v) {
    return step_impl(eng, f, stream_v, true);
}

int ssmap_stage1(ssmap_engine *eng, const float *logits, float temperature, int32_t *semantic,
                 float *semantic_unc, float *semantic_max_prob, float *probs_out, void *stream_v) {
    if (!eng || !logits) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    if (!eng->have_schema) return fail(SSMAP_ERR_NOT_READY, "ssmap_set_schema was not called");
    if (eng->cfg.n_classes < 1 || !(temperature > 0)) return fail(SSMAP_ERR_INVALID_ARG, "bad class count or temperature");
    ssmap_frames f{};
    f.logits = logits;
    f.temperature = temperature;
    f.semantic = semantic;
    f.semantic_unc = semantic_unc;
    f.semantic_max_prob = semantic_max_prob;
    launch_stream(eng->view, f, false, probs_out, reinterpret_cast<hipStream_t>(stream_v));
    HIP_TRY(hipGetLastError());
    return SSMAP_OK;
}

int ssmap_read_results(ssmap_engine *eng, int32_t *found, uint32_t *status, int32_t *n_updates, void *stream_v) {
    if (!eng) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    HIP_TRY(hipMemcpyAsync(eng->h_hdr.data(), eng->d_hdr, sizeof(FrameHeader) * eng->cfg.n_envs,
                           hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int e = 0; e < eng->cfg.n_envs; ++e) {
        if (found) found[e] = eng->h_hdr[e].found;
        if (status) status[e] = eng->h_hdr[e].status;
        if (n_updates) n_updates[e] = (int32_t)eng->h_hdr[e].n_winners;
    }
    return SSMAP_OK;
}

int ssmap_gt_goal_query(ssmap_engine *eng, int32_t *in_disk, int32_t *anywhere, void *stream_v) {
    if (!eng) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    if (!eng->stepped || !eng->d_disk) return fail(SSMAP_ERR_NOT_READY, "ssmap_gt_goal_query before any ssmap_step");
    int rc = check_ready(eng);
    if (rc != SSMAP_OK) return rc;
    const int what = (in_disk ? 1 : 0) | (anywhere ? 2 : 0);
    if (!what) return SSMAP_OK;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    launch_gt_goal(eng->view, what, eng->d_query, stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(eng->h_query.data(), eng->d_query, sizeof(int32_t) * 2 * eng->cfg.n_envs,
                           hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int e = 0; e < eng->cfg.n_envs; ++e) {
        // an env without a frame since its maps were bound has no robot cell and no goal yet
        const bool live = eng->has_frame[e] != 0;
        if (in_disk) in_disk[e] = live ? eng->h_query[2 * e] : 0;
        if (anywhere) anywhere[e] = live ? eng->h_query[2 * e + 1] : 0;
    }
    return SSMAP_OK;
}

int ssmap_profile_begin(ssmap_engine *eng, int32_t max_steps) {
    if (!eng || max_steps < 0 || max_steps > 4096) return fail(SSMAP_ERR_INVALID_ARG, "bad step count");
    for (hipEvent_t ev : eng->prof_events) hipEventDestroy(ev);
    eng->prof_events.clear();
    eng->prof_capacity = eng->prof_steps = 0;
    for (int i = 0; i < max_steps * kProfMarks; ++i) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        eng->prof_events.push_back(ev);
    }
    eng->prof_capacity = max_steps;
    return SSMAP_OK;
}

int ssmap_profile_read(ssmap_engine *eng, double *stage_ms, int32_t *n_steps, void *stream_v) {
    if (!eng || !stage_ms || !n_steps) return fail(SSMAP_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream_v)));
    for (int k = 0; k < kProfMarks - 1; ++k) stage_ms[k] = 0.0;
    for (int s = 0; s < eng->prof_steps; ++s) {
        const hipEvent_t *m = eng->prof_events.data() + (size_t)kProfMarks * s;
        for (int k = 0; k < kProfMarks - 1; ++k) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, m[k], m[k + 1]));
            stage_ms[k] += ms;
        }
    }
    *n_steps = eng->prof_steps;
    eng->prof_steps = 0;
    return SSMAP_OK;
}

int ssmap_update_list(ssmap_engine *eng, int32_t e, const int32_t **pairs_dev) {
    if (!eng || !pairs_dev || e < 0 || e >= eng->cfg.n_envs) return fail(SSMAP_ERR_INVALID_ARG, "bad env index");
    *pairs_dev = (eng->last_list ? eng->last_list : eng->view.winners) + 2ll * e * eng->view.N;
    return SSMAP_OK;
}

}  // extern "C"
