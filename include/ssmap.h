/*
 * ssmap -- MI355X (gfx950) semantic-map fusion engine, C ABI.
 *
 * Drop-in boundary for the per-step map fusion of Semantic-Search's
 * `SemanticMapper` (reference: sem_objnav/obj_nav/mapper.py).  The reference has no FFI
 * layer: its boundary is the Python class, so each entry point below cites the Python
 * method (file:line) whose work it takes over.  A Python host (ctypes) owns every map
 * array as a device tensor and passes plain pointers; the engine allocates only its own
 * per-frame scratch in ssmap_create().  No entry point throws; all return an
 * ssmap_status code.  Of the per-step calls only ssmap_read_results(), ssmap_profile_read() and
 * ssmap_path_append() wait for the stream; the set-up calls (ssmap_set_*, ssmap_bind_env) copy
 * on the null stream and expect the engine's streams to be idle.
 *
 * Layouts (all row-major, little-endian, device memory unless marked HOST):
 *   frames   depth [E][H][W] f32 (mutated in place), logits/probs [E][C][H][W] f32,
 *            semantic / semantic_gt [E][H][W] i32, semantic_unc / semantic_max_prob [E][H][W] f32
 *   maps     semantic_map [Hm][Wm] u32, semantic_map_color [Hm][Wm][3] u8,
 *            height_map [Hm][Wm] f32, gt_map [Hm][Wm] u32, unc_map [Hm][Wm][1] f32,
 *            prob_map [Hm][Wm][C] f32 (avg_prob_seg_map or log_odds_seg_map),
 *            weight_map [Hm][Wm][1] f32, goal_map / hits / views [Hm][Wm] f32,
 *            background_map [Hm][Wm] u32            -- exactly the reference's arrays
 *                                                     (mapper.py:205-255)
 */
#ifndef SSMAP_H
#define SSMAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssmap_engine ssmap_engine;

typedef enum ssmap_status {
    SSMAP_OK = 0,
    SSMAP_ERR_INVALID_ARG = 1,
    SSMAP_ERR_HIP = 2,           /* a HIP runtime call failed; see ssmap_last_error() */
    SSMAP_ERR_UNSUPPORTED = 3,
    SSMAP_ERR_NOT_READY = 4      /* ssmap_step before every env was bound */
} ssmap_status;

/* Aggregation rule = the reference's `prob_aggregate` with the "height_map_" prefix
 * stripped (mapper.py:145-166, 211-254); the prefix is `use_height_map` below. */
typedef enum ssmap_rule {
    SSMAP_RULE_DIRECT = 0,       /* probablistic_mapping=False          mapper.py:343-359 */
    SSMAP_RULE_ARGMAX = 1,       /* argmax (+ filter_goal_prob)         mapper.py:412-447 */
    SSMAP_RULE_GOAL_DECAY = 2,   /* goal_decay (+ filter_goal_prob)     mapper.py:364-411 */
    SSMAP_RULE_AVG_PROB = 3,     /* avg_prob                            mapper.py:517-569 */
    SSMAP_RULE_W_AVG_PROB = 4,   /* w_avg_prob   (weight 1 - entropy)   mapper.py:528-532 */
    SSMAP_RULE_W2_AVG_PROB = 5,  /* w2_avg_prob  (weight 1 / entropy)   mapper.py:533-537 */
    SSMAP_RULE_LOG_ODDS = 6,     /* log_odds                            mapper.py:570-603 */
    SSMAP_RULE_HV = 7,           /* height_map_hv                       mapper.py:604-666 */
    SSMAP_RULE_PROB_HV = 8,      /* height_map_prob_hv                  mapper.py:667-727 */
    SSMAP_RULE_STUBBORN = 9      /* stubborn: argmax + per-cell goal statistics + a Gaussian
                                    naive-Bayes veto of goal labels      mapper.py:412-515 */
} ssmap_rule;

/* Per-env status bits reported by ssmap_read_results(). */
#define SSMAP_ENV_INDEX_ERROR 1u   /* a cell index below -dim was selected (the tallest pixel of
                                      that cell passes the height and depth tests): the reference
                                      raises IndexError at mapper.py:338, before its first map
                                      write; the step leaves this env's maps as they were */

#define SSMAP_MAX_CLASSES 155   /* largest n_classes ssmap_create accepts: class counts without a compiled
                                   streaming kernel keep a frame tile's classes in one CU's LDS */
#define SSMAP_MAX_MAP_DIM 8192
#define SSMAP_MAX_POOL 8
#define SSMAP_MAX_AUX 32

/* Fixed for the life of an engine: SemanticMapper.__init__ arguments (mapper.py:49-72)
 * plus the frame geometry. */
typedef struct ssmap_config {
    int32_t n_envs;              /* E: environments stepped per call */
    int32_t img_h, img_w;        /* H, W of every frame (any size; widths that are a multiple of 4 read the depth 16 bytes at a time) */
    int32_t n_classes;           /* C = seg_model_n_classes; 0 when no class tensor is used */
    int32_t n_schema;            /* num_semantic_classes (9 for the ObjectNav schema) */
    int32_t rule;                /* ssmap_rule */
    int32_t use_height_map;      /* 1 for the "height_map_*" variants */
    int32_t max_map_h, max_map_w;/* largest map any env will bind (sizes scratch) */
    float   depth_min, depth_max;/* depth_range sentinels           mapper.py:282-283 */
    double  fx, fy;              /* focal lengths of pointcloud()   mapper.py:21-25 */
    double  pixel2meter;
    double  height_cutoff_m;
} ssmap_config;

/* One environment's maps, bound after SemanticMapper.reset(map_bounds) allocated them
 * (mapper.py:193-269).  Unused maps for the engine's rule may be NULL. */
typedef struct ssmap_env_maps {
    int32_t map_h, map_w;
    double  lower[3];            /* map_bounds[:3] */
    uint32_t *semantic_map;
    uint8_t  *semantic_map_color;
    float    *height_map;
    uint32_t *gt_map;            /* == semantic_map when probablistic_mapping is False */
    float    *unc_map;
    float    *prob_map;
    float    *weight_map;
    float    *goal_map;
    uint32_t *background_map;
    float    *hits;
    float    *views;
    /* stubborn rule / collect_nb_data (mapper.py:244-250, 260-264) */
    uint32_t *total_view;
    float    *cumulative_conf;
    float    *max_conf;
    float    *max_other_conf;
} ssmap_env_maps;

/* Attributes the reference's callers overwrite after construction
 * (eval_objnav.py:118-137) -- re-sent whenever they change. */
typedef struct ssmap_params {
    float   entropy_threshold;
    float   filter_goal_prob;    /* used only when use_filter != 0 */
    int32_t use_filter;
    float   goal_decay, goal_mark;
    float   hv_ratio, hv_min_views;
    double  hv_limit_obj_view_dist;
    double  found_dist_m;        /* radius of the found disk = int(found_dist_m / pixel2meter) */
    int32_t nb_stats;            /* collect_nb_data with the argmax rule: keep the four goal
                                    statistics of mapper.py:448-469 (always kept by stubborn) */
    int32_t pad;
} ssmap_params;

/* One batch of frames (one per env) = the `obs` dicts of E update_map calls
 * (mapper.py:279-313).  Device pointers; NULL where a key is absent. */
typedef struct ssmap_frames {
    float         *depth;              /* mutated in place like obs["depth"] */
    const float   *logits;             /* raw network output; stage 1 runs when non-NULL */
    const float   *probs;              /* obs["semantic_prob"], used when logits is NULL */
    float          temperature;        /* divides logits (infer.py:97, 309) */
    int32_t       *semantic;           /* obs["semantic"]: input, or stage-1 output */
    float         *semantic_unc;       /* likewise */
    float         *semantic_max_prob;  /* likewise */
    const int32_t *semantic_gt;
    const double  *agent_pose;         /* HOST [E][7]: x y z qw qx qy qz (task.py:190-193) */
    const double  *camera_pose;        /* HOST [E][7] */
    const int32_t *goal_sem_id;        /* HOST [E]: goal2sem_id[objectgoal][0] */
    const uint8_t *skip;               /* HOST [E] or NULL: non-zero = env e has no frame in this batch
                                          (its update_map was not called): its maps, pose, path and
                                          hv overlay stay as they are, its results read found 0,
                                          status 0, n_updates 0; its slices of the arrays above are
                                          neither read nor validated (its stage-1 outputs are
                                          unspecified) */
} ssmap_frames;

/* SemanticMapper.__init__ (mapper.py:49-191): validates the config, allocates scratch. */
int ssmap_create(const ssmap_config *cfg, ssmap_engine **out);
void ssmap_destroy(ssmap_engine *eng);

/* map_seg_to_schema as a table (infer.py:133-138, 336-341): lut[c] for c in [0, C).
 * pool_ids/pool_count: for schema ids 3..8 the model classes whose probabilities are
 * summed into semantic_max_prob (infer.py:109-118, 318-327); pool_ids is [6][SSMAP_MAX_POOL]. */
int ssmap_set_schema(ssmap_engine *eng, const int32_t *lut, int32_t n,
                     const int32_t *pool_ids, const int32_t *pool_count);
/* color_pallete (mapper.py:102-105): n rows of RGB. */
int ssmap_set_palette(ssmap_engine *eng, const uint8_t *rgb, int32_t n);
int ssmap_set_params(ssmap_engine *eng, const ssmap_params *p);

/* One Gaussian naive-Bayes model of the stubborn rule (mapper.py:184-191 loads them from a
 * pickle; here the caller passes the fitted arrays): the model for schema goal id `goal_sem_id`
 * (3..8) over the features {total_view, cumulative_conf, max_conf, max_other_conf}
 * (mapper.py:502-510).  n_classes is 1 or 2; classes[k] is the label of row k (sklearn's
 * classes_), class_log_prior[k] = log(class_prior_[k]), norm_const[k] =
 * -0.5 * sum_j log(2 pi var_[k][j]), theta / var are [n_classes][4] row-major (theta_, var_).
 * The engine evaluates log prior + (norm_const - 0.5 * sum_j (x_j - theta)^2 / var) in float64
 * and takes the first largest.  A goal cell predicted as label 0 is stored as id 2
 * (mapper.py:512-515). */
int ssmap_set_nb_model(ssmap_engine *eng, int32_t goal_sem_id, int32_t n_classes,
                       const int32_t *classes, const double *class_log_prior,
                       const double *norm_const, const double *theta, const double *var);

/* SemanticMapper.reset (mapper.py:193-269) for env `e`: binds freshly initialised maps.
 * Like ssmap_set_schema / ssmap_set_palette / ssmap_set_nb_model / ssmap_set_egomap it updates
 * engine state with blocking copies on the null stream: the caller must have drained every
 * stream that still runs a step of this engine (hipStreamSynchronize) before calling it. */
int ssmap_bind_env(ssmap_engine *eng, int32_t e, const ssmap_env_maps *maps);

/* SemanticMapper.update_map for all E envs, without the ego-map rendering
 * (mapper.py:276-821): stages 1-5 enqueued on `stream` (a hipStream_t passed as void*). */
int ssmap_step(ssmap_engine *eng, const ssmap_frames *frames, void *stream);

/* ssmap_step for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc, code that runs while another thread captures a HIP graph).  No
 * counterpart in the reference.  Every step must be enqueued by a live call: the call prepares the
 * poses on the host (they travel as kernel arguments or through a pinned ring it waits on) and
 * flips host-side state, so a captured graph of it would replay stale poses -- it allocates
 * nothing, it is not replayable.  Same arguments, same kernels and bit-identical results; the difference is where the per-step
 * scratch lives (pixel records, per-block counts, owner flags, update list, staged probability
 * rows, goal_decay's local maps): not in the block ssmap_create allocated but in a file-scope
 * device array of SSMAP_STATIC_SCRATCH_BYTES that every GPU loading the library holds until the
 * process ends.  The call allocates and frees nothing.  It returns SSMAP_ERR_UNSUPPORTED, with
 * the two sizes in ssmap_last_error(), when the engine's per-step scratch is larger than the
 * array (it grows with n_envs * img_h * img_w and, for the probability rules, n_classes), and
 * when the stubborn rule / collect_nb_data is fed logits before any ssmap_step allocated its
 * probability buffer.  All engines on one GPU share the array: nothing in it is carried from one
 * step to the next (every step rewrites what it reads), but calls that overlap in time on one
 * GPU (two streams, two threads) race; ssmap_update_list after such a step points into the array
 * and is valid until the next ssmap_step_noalloc on that GPU. */
#define SSMAP_STATIC_SCRATCH_BYTES (64u << 20)
int ssmap_step_noalloc(ssmap_engine *eng, const ssmap_frames *frames, void *stream);

/* Stage 1 alone: the softmax/entropy/confidence tail of predict()
 * (infer.py:97-118, 302-327).  probs_out may be NULL. */
int ssmap_stage1(ssmap_engine *eng, const float *logits, float temperature,
                 int32_t *semantic, float *semantic_unc, float *semantic_max_prob,
                 float *probs_out, void *stream);

/* The two map questions of the shortest-path policy's found override (env.py:564-605), for every
 * env, answered from the maps where they lie, after the last ssmap_step: in_disk[e] = some gt_map
 * cell inside the found disk around the robot cell of that step carries the goal's schema id
 * (env.py:581-593; the disk is the one the found decision uses), anywhere[e] = some gt_map cell at
 * all does (env.py:600, 604).  HOST arrays [E]; either may be NULL (its scan is then skipped).
 * An env that has had no frame since ssmap_bind_env (every step so far skipped it) has no robot
 * cell and no goal yet: both answers are 0.  Waits for `stream`. */
int ssmap_gt_goal_query(ssmap_engine *eng, int32_t *in_disk, int32_t *anywhere, void *stream);

/* Waits for `stream`, then copies per-env results of the last step to HOST arrays [E]:
 * found (1 = is_found True, 0 = None), status (SSMAP_ENV_* bits), n_updates (K). */
int ssmap_read_results(ssmap_engine *eng, int32_t *found, uint32_t *status,
                       int32_t *n_updates, void *stream);

/* Device pointer to the last step's update list of env `e` (the reference's
 * valid_flat_idx / update_cell_idx, mapper.py:333-337): n_updates int32 pairs, unordered.
 * Word 0: flat pixel index in bits 0-28; bit 29 = dropped because a later pixel writes the
 * same wrapped cell; bits 30 / 31 = the cell's row / column index was negative before numpy's
 * wrap.  Word 1: flat index of the (wrapped) map cell. */
int ssmap_update_list(ssmap_engine *eng, int32_t e, const int32_t **pairs_dev);

/* Measurement hooks (no counterpart in the reference).  After ssmap_profile_begin the next
 * `max_steps` calls of ssmap_step record HIP events on their stream around five stage
 * groups: [0] geometry pass (pixel records + frame header), [1] tallest-per-cell + wrap
 * resolution,
 * [2] the streaming stage-1 kernel (one launch), [3] rule update, [4] found decision.
 * ssmap_profile_read waits for `stream`, returns the summed milliseconds per group
 * (stage_ms[5]) and the number of recorded steps, and rearms. */
int ssmap_profile_begin(ssmap_engine *eng, int32_t max_steps);
int ssmap_profile_read(ssmap_engine *eng, double *stage_ms, int32_t *n_steps, void *stream);

/* Egocentric observation maps, `SemanticMapper.ego_map` (mapper.py:876-998): crop around the
 * agent (large crop + offset_for_cut), goal cells recoloured, nearest-neighbour rotation by the
 * heading, agent disk and auxiliary-angle squares drawn on top, centre crops of the small and
 * the large side, INTER_AREA resize to out_w x out_h (uncertainty: border -0.5, nearest resize).
 * The constants are the ones the reference derives in its constructor (mapper.py:77-109). */
typedef struct ssmap_egomap_config {
    int32_t small_crop_px, large_crop_px;   /* int(egomap_*_side_m / pixel2meter), mapper.py:91-95 */
    int32_t offset_for_cut;                 /* 150, mapper.py:96 */
    int32_t out_w, out_h;                   /* egomap_size, mapper.py:98 */
    int32_t agent_radius_px;                /* mapper.py:97 */
    int32_t aux_bin_px;                     /* mapper.py:100 */
    int32_t n_aux;                          /* aux_num_bins, <= SSMAP_MAX_AUX */
    int32_t aux_points[2 * SSMAP_MAX_AUX];  /* (row, col) offsets from the image centre, mapper.py:82-89 */
    uint8_t active_goal_color[3];           /* mapper.py:103 */
    uint8_t arrow_color[3];                 /* mapper.py:106-108 */
    uint8_t pad[2];
} ssmap_egomap_config;
int ssmap_set_egomap(ssmap_engine *eng, const ssmap_egomap_config *cfg);

/* One set of ego maps per env from the maps as they stand (call after ssmap_step).
 * agent_pose: host [E][7] = x,y,z,qw,qx,qy,qz; goal_sem_id: host [E]; aux_color: host
 * [E][n_aux][3] = (aux_pred_color * onehot[j]).astype(uint8) (mapper.py:1038) or NULL for no
 * auxiliary squares.  Outputs are device arrays: ego_small / ego_large [E][out_h][out_w][3] u8,
 * unc_small / unc_large [E][out_h][out_w] f32 (may be NULL; they need unc_map bound). */
int ssmap_ego_maps(ssmap_engine *eng, const double *agent_pose, const int32_t *goal_sem_id,
                   const uint8_t *aux_color, uint8_t *ego_small, uint8_t *ego_large,
                   float *unc_small, float *unc_large, void *stream);

/* Path overlay (mapper.py:822-833): the cells of the agent's trail are shown as id n_schema + 1
 * in semantic_map / semantic_map_color at the end of every update_map from the second frame on.
 * The line raster itself (cv2.line, mapper.py:823-830) stays with the caller, who appends the
 * cells it newly covers: cells is HOST [n], flat indices into env e's map.  ssmap_bind_env
 * empties the list.  ssmap_path_paint stamps every listed cell of every env; call it after
 * ssmap_step (the found decision reads the map before the trail is drawn, mapper.py:735-833). */
int ssmap_path_append(ssmap_engine *eng, int32_t e, const int32_t *cells, int32_t n, void *stream);
int ssmap_path_paint(ssmap_engine *eng, void *stream);

const char *ssmap_last_error(void);
const char *ssmap_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SSMAP_H */
