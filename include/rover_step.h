/*
 * rover_step.h — C ABI of librover_step.so: the MI355X (gfx950) implementation of the rover task's
 * vectorised env.step() hot path of abmoRobotics/isaac_rover_2.0.
 *
 * The reference is pure Python/PyTorch and has no FFI; each entry point below names the reference
 * function(s) it replaces (paths relative to omniisaacgymenvs/ in the reference repository).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add to call them from
 * tasks/rover.py.
 *
 * Conventions
 *  - Plain C, no torch types.  All `*_d` / step pointers are DEVICE pointers the caller owns (e.g.
 *    tensor.data_ptr()); the library borrows them for the duration of the call and never frees them.
 *  - `set_*` calls take HOST or DEVICE pointers (hipMemcpyDefault), copy into library-owned device memory,
 *    and synchronise; they are init-time calls.
 *  - Step calls only enqueue work on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *    do not synchronise.  A ctx belongs to one host thread at a time; distinct ctxs are independent.
 *  - Every function returns 0 on success or a negative ROVER_E_* code; rover_last_error() gives the text.
 *    No C++ exception crosses the boundary.
 *  - float = IEEE binary32, env-major row-major arrays, quaternions (w,x,y,z), int64 flags like the
 *    reference's torch.long buffers (tasks/base/rl_task.py:98-107).
 */
#ifndef ROVER_STEP_H
#define ROVER_STEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROVER_API __attribute__((visibility("default")))

#define ROVER_OK            0
#define ROVER_E_INVALID    -1   /* bad argument / shape                         */
#define ROVER_E_STATE      -2   /* required set_* call missing                  */
#define ROVER_E_HIP        -3   /* HIP runtime error (text has the HIP message) */
#define ROVER_E_NOMEM      -4

#define ROVER_MAP_TERRAIN   0   /* tasks/utils/terrain/knn_terrain/{map_indices,triangles,vertices}.pt (camera.py:154-161)      */
#define ROVER_MAP_ROCKS     1   /* tasks/utils/terrain/knn_rocks/{...}.pt (rock_detect.py:151-158)  */

/* rover_step() flags */
#define ROVER_STEP_INCREMENT_PROGRESS 1u  /* progress_buf += 1 first (rl_task.py:250)                      */
#define ROVER_STEP_COMPACT            2u  /* also emit reset_ids / n_reset (rover.py:356)                   */

typedef struct rover_ctx rover_ctx;

/* Constants the reference keeps in cfg/task/Rover.yaml:11,37-46 and hard-codes in rover.py:119,101,353 */
typedef struct {
    int32_t num_envs;              /* envs handled by this ctx (this GPU's shard)                         */
    int32_t num_envs_global;       /* self.num_envs of rover.py:517 (collision_penalty); 0 = num_envs     */
    int32_t env_offset;            /* global id of local env 0 (compaction emits global ids)              */
    int32_t device;                /* HIP device ordinal                                                   */
    int32_t curriculum_level;      /* rover.py:101,353: rock collision active when >= 2                    */
    int32_t max_episode_length;    /* rover.py:119 (3000)                                                  */
    float pos_reward;              /* Rover.yaml:39 */
    float heading_contraint_reward;/* Rover.yaml:43 */
    float motion_contraint_reward; /* Rover.yaml:44 */
    float goal_angle_reward;       /* Rover.yaml:45 */
    float boogie_contraint_reward; /* Rover.yaml:46 */
} rover_cfg;

/* Sim state in (rover.py:274-275,291,343,470-476; Memory rover.py:60-77).  Device pointers. */
typedef struct {
    const float *pos;        /* [E,3]  RoverView.get_world_poses()[0]                                     */
    const float *quat;       /* [E,4]  RoverView.get_world_poses()[1], (w,x,y,z)                          */
    const float *joints;     /* [E,13] RoverView.get_joint_positions()  (legend rock_detect.py:175-187)   */
    const float *target;     /* [E,3]  self.target_positions                                              */
    const float *lin_hist;   /* [E,3]  linear_velocity.tracker, newest first                              */
    const float *ang_hist;   /* [E,3]  angular_velocity.tracker                                           */
    const float *euler_pre;  /* [E,3]  self.rover_rot captured in pre_physics_step (rover.py:343)         */
    int64_t *progress;       /* [E]    progress_buf (in/out)                                              */
} rover_step_in;

/* Outputs (rl_task.py:98-107 buffers, rover.py:524-531 extras).  Device pointers; NULL = not wanted
 * for the optional ones. */
typedef struct {
    float *obs;                    /* [E, obs_stride] obs_buf; row = [4 proprio | Ns sparse | Nd dense]   */
    int64_t obs_stride;            /* elements per obs row; 0 = 4+Ns+Nd                                   */
    float *rew;                    /* [E] rew_buf                                                         */
    int64_t *reset;                /* [E] reset_buf                                                       */
    int64_t *rock_collision;       /* [E] self.rock_collison (rover.py:667-668)                           */
    float *ex_pos_reward;          /* [E] extras, rover.py:524-531 — all optional                         */
    int64_t *ex_collision_penalty;
    float *ex_uprightness_penalty;
    float *ex_heading_contraint_penalty;
    float *ex_motion_contraint_penalty;
    float *ex_goal_angle_penalty;
    float *ex_torque_penalty_driving;
    float *ex_torque_penalty_steering;
    int64_t *reset_ids;            /* [E] ascending global env ids with reset != 0 (ROVER_STEP_COMPACT)   */
    int32_t *n_reset;              /* [1]                                                                 */
    float *euler;                  /* optional [E,3] self.rover_rotation (rover.py:275)                   */
    float *heading_diff;           /* optional [E]   self.heading_diff  (rover.py:283)                    */
    float *ray_dist;               /* optional [E,P] Camera.get_depths distances (camera.py:145)          */
    float *wheel_dist;             /* optional [E,24] rock_detect.py:146                                  */
    float *body_dist;              /* optional [E,2]  rock_detect.py:147                                  */
    /* ADDITIONAL output, not part of the reference's step (its per-step collision term is the ray-based
     * rock_collision above): the stone_info occupancy mask BASELINE.json configs[2] names.  1 where
     * nearest_rock(pos_xy) = min_s(|pos_xy - stone_s| - r_s) <= stone_margin, the clearance test of
     * check_goal_collision / avoid_pos_rock_collision (rover.py:536-539,655-658) applied to the rover itself.
     * Written by the collision stage (rover_get_observations / rover_step); never feeds reward or done. */
    int64_t *stone_collision;      /* optional [E]; needs rover_set_stones                                */
    float stone_margin;            /* metres, <= 1.4 (the reach of the occupancy grid); 0 = centre inside a disc */
    /* ADDITIONAL output: reset != 0 as one byte per env — the form the multi-GPU gather ships (SURVEY.md 8e:
     * "done [E/8] u8 or i64"); written by the is_done stage next to the int64 reset_buf. */
    uint8_t *done_u8;              /* optional [E]                                                         */
    /* The other two return values of Camera.get_depths (camera.py:118-120,145: `return output_distances, output_pt, sources`;
     * rover.py:286 binds and drops them): per terrain ray its origin (camera.py:212) and the point the reference calls the
     * intersection, sources - d * k with d = -normalize(direction) and k the ray's distance (ray_casting.py:63; for a miss
     * k = 11.0).  In the as-shipped fp16 mode both are fp16 values (widened to f32), each operation rounded to fp16. */
    float *ray_src;                /* optional [E,P,3]                                                     */
    float *hit_pt;                 /* optional [E,P,3]                                                     */
} rover_step_out;

/* ---- lifetime ------------------------------------------------------------------------------------ */
ROVER_API int rover_create(const rover_cfg *cfg, rover_ctx **out);        /* RoverTask.__init__ rover.py:81-185 */
ROVER_API void rover_destroy(rover_ctx *ctx);
ROVER_API const char *rover_last_error(const rover_ctx *ctx);             /* ctx may be NULL: last create error */
ROVER_API const char *rover_version(void);

/* ---- init-time tables ------------------------------------------------------------------------------ */
/* Camera._load_triangles_with_indices camera.py:154-161 / Rock_Detection rock_detect.py:151-158.
 * map_idx [X][Y][K] int32 (the layout after the two swapaxes), tris [T][3] int32, verts [V][3] IEEE half bits.
 * The library re-packs the three tables into one per-cell contiguous fp16 block [X*Y][9][K8]. */
ROVER_API int rover_set_knn_map(rover_ctx *ctx, int which, const int32_t *map_idx, int32_t X, int32_t Y, int32_t K,
                                const int32_t *tris, int32_t T, const uint16_t *verts_f16, int32_t V,
                                float cell_size, float shift_x, float shift_y);
/* Heightmap (heightmap_distribution.py:11-134): points [P][3] float64 in the post-swap frame, index lists */
ROVER_API int rover_set_distribution(rover_ctx *ctx, const double *points, int32_t P, const int64_t *sparse_idx,
                                     int32_t Ns, const int64_t *dense_idx, int32_t Nd);
/* heightmap_tensor.pt, rover.py:210-213 */
ROVER_API int rover_set_heightfield(rover_ctx *ctx, const float *hm, int32_t N0, int32_t N1, float horizontal_scale,
                                    float vertical_scale, float shift_x, float shift_y);
/* read_stone_info output [S][7] float32 (utils/terrain_utils/terrain_utils.py:416-424) */
ROVER_API int rover_set_stones(rover_ctx *ctx, const float *info7, int32_t S);
ROVER_API int rover_set_curriculum_level(rover_ctx *ctx, int32_t level);  /* rover.py:353 */

/* ---- per-step hot path ----------------------------------------------------------------------------- */
/* One fused RLTask.post_physics_step (rl_task.py:239-259): get_observations + calculate_metrics + is_done
 * (+ progress increment, + done compaction), same dataflow and order as the reference. */
ROVER_API int rover_step(rover_ctx *ctx, const rover_step_in *in, const rover_step_out *out, uint32_t flags, void *stream);
/* The same three stages as separate calls, for a task that keeps the reference's method split:
 * RoverTask.get_observations rover.py:272-336 (writes obs, rock_collision, optional intermediates) */
ROVER_API int rover_get_observations(rover_ctx *ctx, const rover_step_in *in, const rover_step_out *out, void *stream);
/* RoverTask.calculate_metrics rover.py:460-531 (reads out->rock_collision and the heading/position state
 * left by the last rover_get_observations on this ctx; writes rew + extras) */
ROVER_API int rover_calculate_metrics(rover_ctx *ctx, const rover_step_in *in, const rover_step_out *out, void *stream);
/* RoverTask.is_done rover.py:610-647 (writes reset) */
ROVER_API int rover_is_done(rover_ctx *ctx, const rover_step_in *in, const rover_step_out *out, void *stream);
/* Camera.get_depths(positions, rotations) camera.py:60-145 as its own call: positions [E,3], rotations [E,3] EULER angles (what the
 * reference passes: self.rover_rotation, rover.py:286) -> distances [E,P], points [E,P,3], sources [E,P,3] (each optional).  Runs the
 * step's ray pipeline on the given poses (ray precision / cell index mode / ray-cast variant as set); the observation state of the
 * ctx (heading, euler of the last rover_get_observations) is left untouched. */
ROVER_API int rover_get_depths(rover_ctx *ctx, const float *positions, const float *rotations_euler, float *distances,
                               float *points, float *sources, void *stream);
/* Rock_Detection.get_collisions(positions, rotations, joint_states) rock_detect.py:52-149 as its own call (the task holds
 * self.Rock_detector and calls it at rover.py:291): positions [E,3], rotations [E,3] EULER angles (self.rover_rotation), joints [E,13]
 * (RoverView.get_joint_positions; NULL = all zero) -> wheel_dist [E,24], body_dist [E,2] (each optional), the reference's
 * (output_distances[:, 0:24], output_distances[:, 24:]).  Same ray pipeline and options as the step.
 * rover_get_depths and rover_get_collisions overwrite the ctx's ray workspace (ray records, sorted list, distances, cull counters:
 * what rover_replay_raycast / rover_get_cull_info / the profile describe afterwards is THIS call's rays); they leave the observation
 * state (euler, heading) alone and do not stand in for rover_get_observations: rover_calculate_metrics still requires that one. */
ROVER_API int rover_get_collisions(rover_ctx *ctx, const float *positions, const float *rotations_euler, const float *joints,
                                   float *wheel_dist, float *body_dist, void *stream);
/* The ray phase on its own, for parity tests that must not depend on the pose trigonometry (ray_casting.py:3-66 + the cell lookup
 * camera.py:233-264 + min over K, camera.py:116-117):
 *  rover_export_rays: the rays of the last cast on this ctx in slot order — per env 24 wheel, 2 body, P heightmap rays: origins
 *    src [E,26+P,3], the ray records' directions dir [E,26+P,3] (= -normalize(direction), ray_casting.py:31, as the kernels use it),
 *    cell ids cell [E,26+P] int32, distances dist [E,26+P] (each optional).
 *  rover_cast_rays: casts caller-supplied rays in the same layout (origins + record directions; slots 0..25 against the rocks map,
 *    the rest against the terrain map) through the step's sort + ray-cast kernels (variant / precision / cell index mode as set) and
 *    returns their distances dist [E,26+P].  Overwrites the ray workspace like rover_get_depths.  The directions are used as they are:
 *    with the culled / staged ray cast (variants 3, 4), whose rejection proofs assume what -normalize() produces, a finite direction
 *    whose squared length is not within 1e-5 of 1 (4e-3 with ray_precision 2) is ROVER_E_INVALID — the call synchronises the stream to
 *    find out; variants 1 and 2 evaluate every triangle and take any direction. */
ROVER_API int rover_export_rays(rover_ctx *ctx, float *src, float *dir, int32_t *cell, float *dist, void *stream);
ROVER_API int rover_cast_rays(rover_ctx *ctx, const float *src, const float *dir, float *dist, void *stream);
/* reset_buf.nonzero() rover.py:356 without the host sync: ids ascending (+env_offset), count to n_reset[0] */
ROVER_API int rover_compact_resets(rover_ctx *ctx, const int64_t *reset, int64_t *reset_ids, int32_t *n_reset, void *stream);
/* tensor_quat_to_eul tasks/utils/math/tensor_quat_to_euler.py:6-31 */
ROVER_API int rover_quat_to_euler(rover_ctx *ctx, const float *quat, float *euler, int32_t n, void *stream);

/* ---- reset / spawn / goal validation (rover.py:533-564, 588-608, 649-661) ---------------------------- */
/* nearest_rock = min_s(|p - stone_s| - r_s)  (rover.py:536-538,655-658); xy [n][2] -> out [n] */
ROVER_API int rover_clearance(rover_ctx *ctx, const float *xy, int32_t n, float *out, void *stream);
/* avoid_pos_rock_collision rover.py:649-661: per env, x += 0.05 while clearance <= 1.4; pos [n][3] in place */
ROVER_API int rover_shift_spawns(rover_ctx *ctx, float *pos3, int32_t n, int32_t max_iter, void *stream);
/* get_pos_height rover.py:588-608: xy [n][2] -> out [n] */
ROVER_API int rover_sample_height(rover_ctx *ctx, const float *xy, int32_t n, float *out, void *stream);
/* generate_goals + random_goals + check_goal_collision rover.py:533-564, radius 8 (rover.py:578), then the
 * goal z lookup of set_targets rover.py:582-583.  env_ids [n] int64 (local ids into target3/initial_pos3);
 * draws = [max_draws][n] uniforms in [0,1) replacing torch.rand (NULL = library Philox stream seeded by
 * `seed`); reproduces the env_ids = mask*env_ids aliasing of rover.py:540.  n_draws_used[0] (optional)
 * receives the number of draws consumed, or -1 if max_draws ran out before every goal was clear. */
ROVER_API int rover_generate_goals(rover_ctx *ctx, const int64_t *env_ids, int32_t n, const float *initial_pos3,
                                   float *target3, float radius, const float *draws, int32_t max_draws, uint64_t seed,
                                   int32_t *n_draws_used, void *stream);

/* ---- device-side reset orchestration ("next" row f-2): reset_idx + set_targets rover.py:416-453,566-584 ----- */
/* Consumes the compacted reset ids of the last step WITHOUT the host sync of rover.py:357: the count is read from
 * device memory (n_reset_dev) by the kernels.  For every listed env: pose = initial_pos, orientation = the
 * reference's (w,x,y,z) <- scipy (x,y,z,w) yaw quirk with d = yaw_deg[i] or a Philox draw in [0,360], joint
 * positions / velocities zeroed, reset = 0, progress = 0; then (if target3 != NULL) goals are re-drawn and validated
 * exactly like rover_generate_goals, including the goal z lookup.  All pointers are device pointers. */
typedef struct {
    const int64_t *reset_ids;     /* [E] ascending GLOBAL env ids (rover_step's reset_ids)                         */
    const int32_t *n_reset_dev;   /* [1] device count; NULL = use n_reset_host                                     */
    int32_t n_reset_host;
    const float *initial_pos3;    /* [E,3] self.initial_pos                                                        */
    float *pos3, *quat4;          /* [E,3], [E,4] RoverView poses (in place)                                       */
    float *joint_pos13, *joint_vel13; /* [E,13] optional                                                           */
    float *base_pos3;             /* [E,3] optional self.base_pos                                                  */
    int64_t *reset, *progress;    /* [E] reset_buf, progress_buf                                                   */
    const int32_t *yaw_deg;       /* [yaw_deg_len] optional: replaces random.randint(0, 360) (rover.py:429); entry i
                                   * belongs to reset_ids[i]                                                        */
    float *target3;               /* [E,3] optional self.target_positions: re-draw + validate goals                */
    float radius;                 /* rover.py:578 (8)                                                              */
    const float *draws;           /* optional [max_draws][n] uniforms (needs n_reset_host)                         */
    int32_t max_draws;
    uint64_t seed;
    int32_t *n_draws_used;        /* optional [1]                                                                  */
    int32_t yaw_deg_len;          /* entries in yaw_deg: >= n_reset_host, or >= num_envs when n_reset_dev is used   */
    const uint64_t *seed_dev;     /* optional [1] device word ADDED to `seed` (mod 2^64) when the kernels run: a caller that replays the
                                   * call from a captured hipGraph keeps its step counter there, so that every replay draws anew   */
} rover_reset_io;
ROVER_API int rover_reset_envs(rover_ctx *ctx, const rover_reset_io *io, void *stream);

/* ---- action side ("next" row f-1) --------------------------------------------------------------------------- */
/* pre_physics_step rover.py:338-414 minus the reset branch, one kernel: euler_pre = tensor_quat_to_eul(quat) (:343),
 * Memory.input_state for both histories (:379-380, in place, newest first), Ackermann (:391) and the scatter of
 * 4 steering angles / 6 wheel speeds into the [E,13] joint-target arrays at the indices of
 * robots/articulations/views/rover_view.py:45-46.  actions [E,2]; euler_pre / targets optional; actions_nn optional [E,2,3]:
 * self.actions_nn (:366: the newest action prepended, the oldest dropped), in place. */
ROVER_API int rover_pre_physics_step(rover_ctx *ctx, const float *actions, const float *quat, float *lin_hist,
                                     float *ang_hist, float *euler_pre, float *joint_pos_targets13,
                                     float *joint_vel_targets13, float *actions_nn, void *stream);
/* Ackermann tasks/utils/kinematics.py:13-67 on its own:
 * lin, ang [n] -> steering [n][6], velocities [n][6] in wheel order FL,FR,ML,MR,RL,RR */
ROVER_API int rover_ackermann(rover_ctx *ctx, const float *lin, const float *ang, int32_t n, float *steering,
                              float *velocities, void *stream);

/* ---- KNN map builder ("next" row f-3): tasks/utils/rover_utils.py:48-123 ------------------------------------ */
/* For every cell (x, y) of an X x Y map at `res` metres per cell (cell position = (x res, y res), rover_utils.py:75-81)
 * the K triangles whose centroid ((v0+v1+v2)/3, :68-70) is nearest in xy, ascending.  vertices [V,3] float32,
 * triangles [T,3] int32 (host or device pointers); map_idx_out [X,Y,K] int32 is a DEVICE pointer.  Ranking is exact f32
 * squared distance with ties broken by triangle id; the reference ranks fp16-rounded distances with torch.topk (ties
 * unspecified), so its maps agree with this one only up to that rounding.  Synchronous (init-time tool); needs no
 * prior set_* call.  Fails with ROVER_E_INVALID when T < K or a search ring holds more than 8192 candidates. */
ROVER_API int rover_build_knn_map(rover_ctx *ctx, const float *vertices, int32_t V, const int32_t *triangles, int32_t T,
                                  int32_t X, int32_t Y, float res, int32_t K, int32_t *map_idx_out);

/* The same builder ranking EXACTLY like the reference does (rover_utils.py:71-102): triangle centroids and cell coordinates
 * are fp16 tensors there, so per axis the difference is rounded to fp16, the norm is the f32 sqrt of the f32 sum of squares
 * rounded to fp16, and torch.topk picks the K smallest of those fp16 distances (tie order unspecified; here: by triangle id).
 * cell_x_f16 [X] / cell_y_f16 [Y] (IEEE half bits, host or device; NULL = fp16(float(i) * res), what ATen's CUDA arange yields)
 * are the coordinate tables of the reference's torch.arange(0, X*res, res, dtype=float16) (:75-76) — ATen's CPU arange
 * evaluates that in vector-width-dependent fp16 steps, so a map built by the reference on a CPU is reproduced by passing the
 * table that host produced.  Result per cell: the same multiset of fp16 distances as the reference's list, the same triangles
 * except among those tied with the K-th distance. */
ROVER_API int rover_build_knn_map_ref(rover_ctx *ctx, const float *vertices, int32_t V, const int32_t *triangles, int32_t T,
                                      int32_t X, int32_t Y, float res, int32_t K, const uint16_t *cell_x_f16,
                                      const uint16_t *cell_y_f16, int32_t *map_idx_out);

/* ---- policy-side consumer of the obs layout ("next" row f-4): learning/model.py:105-121 Layer = Linear + activation --- */
#define ROVER_ACT_NONE      0
#define ROVER_ACT_LEAKYRELU 1   /* nn.LeakyReLU(), slope 0.01 (cfg/trainSKRL/RoverPPOSKRL.yaml:5,9) */
#define ROVER_ACT_TANH      2   /* the actor head, model.py:182 */
#define ROVER_ACT_RELU      3
#define ROVER_ACT_ELU       4
/* y[:, 0:N] = act(x[:, 0:K] @ weight^T + bias); weight [N][K] (torch nn.Linear layout), bias [N] or NULL, N <= 256.
 * x / y are addressed as (pointer, row stride in floats), so a layer can read an obs slice (model.py:186-187) and write
 * into a column block of the concat buffer (:191-192) without copies.  f32-input MFMA, fp32 accumulate. */
ROVER_API int rover_linear_forward(rover_ctx *ctx, const float *x, int64_t x_stride, int32_t M, int32_t K,
                                   const float *weight, const float *bias, int32_t N, int32_t activation, float *y,
                                   int64_t y_stride, void *stream);

/* A chain of 2 or 4 such layers in ONE kernel: y = L_n(... L_1(x[:, 0:K0])), L_i(v) = act_i(W_i v + b_i) — an Encoder
 * (learning/model.py:122-150: 634 -> 80 -> 60) or the MLP with its head (:176-195: 124 -> 256 -> 160 -> 128 -> 2).  Only x and the
 * last layer's output touch HBM: the activations stay in the MFMA accumulator registers, which are the next layer's B operand as
 * they are (csrc/rover_mlp.hip).  weights[i] is nn.Linear's [widths[i]][widths[i-1]] (K0 for i = 0), biases[i] may be NULL.
 * Built tile shapes: 2 layers with widths <= 96, <= 64; 4 layers with widths <= 256, <= 160, <= 128, <= 16 and activation 0, 1 or 3
 * (none / LeakyReLU / ReLU) on the three hidden layers (else ROVER_E_INVALID: use rover_linear_forward per layer).  Same numerics as
 * rover_linear_forward up to the summation order inside a layer.
 * Small batches (M < 20 480 with a 2-layer chain) go through a split-k scratch buffer that the ctx owns and grows on demand: the
 * chain entry points of one ctx must therefore run on ONE stream at a time (two forwards overlapped on different streams need two
 * ctxs), and the first small-batch call of a given size must not happen inside a stream capture (it may hipMalloc and synchronise);
 * warm it up once before capturing. */
ROVER_API int rover_mlp_chain_forward(rover_ctx *ctx, const float *x, int64_t x_stride, int32_t M, int32_t K0, int32_t n_layers,
                                      const float *const *weights, const float *const *biases, const int32_t *widths,
                                      const int32_t *activations, float *y, int64_t y_stride, void *stream);

/* Two 2-layer chains over the SAME M rows — the two encoders of the actor / critic (learning/model.py:188-190: sparse and dense
 * heightmap slices of one obs row; neither depends on the other) — and, optionally, the copy of the proprioception columns into the
 * concat buffer (:191: dst[r][0:copy_cols] = src[r][0:copy_cols]; copy_cols = 0: none).  Small batches (M < 20 480, both chains of
 * the same built tile shape) run the two side by side: ONE launch for both first layers (split along k), one for both second layers
 * and the copy — a third of the actor forward's launches at the reference's default numEnvs 512 (cfg/task/Rover.yaml:11).  Otherwise
 * the chains run one after the other.  Results are those of two rover_mlp_chain_forward calls either way. */
typedef struct {
    const float *x; int64_t x_stride; int32_t K0, n_layers;
    const float *const *weights; const float *const *biases; const int32_t *widths; const int32_t *activations;
    float *y; int64_t y_stride;
} rover_chain_desc;
ROVER_API int rover_mlp_chain_pair_forward(rover_ctx *ctx, int32_t M, const rover_chain_desc *a, const rover_chain_desc *b,
                                           const float *copy_src, int64_t copy_src_stride, float *copy_dst, int64_t copy_dst_stride,
                                           int32_t copy_cols, void *stream);

/* ---- tuning knobs ------------------------------------------------------------------------------------- */
/* name = "raycast_variant": 0 = auto; 1 = one half-wave per ray in env order, every cell block streamed from HBM;
 *        2 = rays counting-sorted by (map, cell), one wave per run of sorted rays, the cell's triangles held in registers
 *        (needs K <= 256 on both maps); 3 = culled: the sorted rays of 2, but a conservative bounding-sphere + normal test
 *        (16 B per triangle, built at rover_set_knn_map) first proves for most (ray, triangle) pairs that ray_casting.py:59
 *        rejects them, and only the remaining candidates get the exact arithmetic (csrc/rover_cull.hip) — in f32 or, with
 *        ray_precision = 2, in the reference's as-shipped fp16 arithmetic (its own, wider proof margins).
 *        4 = staged: the proof of 3 on per-cell record rows ordered by a distance bound (16 suffix levels per cell: a ray tests only the
 *        prefix it cannot clear as a group; 16 B per pair for the sphere test, 8 B per pair — f32 proof — for the normal test), one lane
 *        per (ray, chunk of 8 pairs), then the same exact phase.
 *        All give bit-identical results.  auto: fp32 arithmetic (ray_precision 0, 1) — 4 from 24 576 rays per step, 1 below;
 *        ray_precision = 2 — 2 up to 24 576 rays per step; above that 4: in env order below 98 304 rays per step, behind the sort beyond
 *        (on an irregular terrain mesh — fewer than half of its cells with a usable far bound — 4 from two heightmap rays per terrain cell, 3 below).
 * name = "lane_env_order" (variant 4): 1 = no sort, the ray slots in env order; 0 = rays sorted by (map, cell); -1 (default) = auto: env
 *        order while a step's heightmap rays are fewer than 1.5 per terrain cell and the rovers fewer than one per 64 cells (ray_precision
 *        2: below 98 304 rays per step).
 * name = "lane_rocks" (variant 4, sorted): 1 = the rock rays through the staged kernel too (one ray-cast launch), 0 = through the culled one
 *        (3); -1 (default) = auto: 1.
 * name = "ray_precision": 0 (default) = the reference's fp32 mode, which the parity tests pin.
 *        1 = every ray origin / direction rounded to fp16 before the cell lookup and the ray maths, like the reference AS
 *        SHIPPED (Camera.dtype = float16: camera.py:55,212; rock_detect.py:319,371); f32 arithmetic after that.
 *        2 = AS SHIPPED: (1) plus every operation of ray_casting.py:31-59 rounded to fp16 the way ATen's Half kernels do,
 *        and fp16 collision thresholds (rover.py:667-668).  Bit-identical to the as-shipped reference on ray origins,
 *        distances, collision mask and done flags (ray-cast variants 2 and 3).
 * name = "bin_low_bits": width of the low digit of the ray bucket sort (2^bits map cells per bucket), 8..12, or 0 (default) = chosen by
 *        the library: 10, raised while that gives more than 4 096 buckets, lowered (to 8 at most) when that lets a sort entry fit one
 *        dword (num_envs x padded rays per env <= 2^(32 - bits)).  Results do not depend on it.
 * name = "raycast_early_out": 1 (default) = the binned kernel drops a whole packed pair of triangles per lane (the far half
 *        of a cell's K-nearest list; on the rocks map also the near half) when a conservative test on the numerators shows that
 *        every triangle of it fails the barycentric test; results are bit-identical with 0 (A/B and tests).
 * name = "cell_index_mode": how `(xy - shift) / 0.1` (camera.py:241, rock_detect.py:381, rover.py:590; a Python-float divisor)
 *        is evaluated.  0 = cpu_div (default): a correctly rounded division, what ATen's CPU kernel does and what the golden
 *        vectors (captured from the reference on CPU) pin.  1 = cuda_rcp: multiplication by 1.0f / 0.1f = 10.0f, what ATen's
 *        CUDA kernel does ("a * reciprocal(b)" for a CPU-scalar divisor) — the device the reference actually runs on.  The
 *        two differ only for coordinates within an ulp of a .5 tie of the cell grid (tests/test_oracle_golden.py).
 * name = "cull_queue_mb": most MiB the candidate queue of the culled ray cast may take (default 1536).  A wave of a launch owns a
 *        region of 1 024 entries (8 KB; a run that finds more finishes them and scans on), so a launch needs 8 KB per run of 64
 *        sorted rays: 712 MB at 65 536 envs x 63 rays.  Past the budget a step's ray cast is cut into several launches that
 *        re-use the regions (each extra launch costs ~25 us); an allocation failure is an error (ROVER_E_NOMEM), never a
 *        silent change of kernel.
 * name = "staged_tables": which proofs' tables of the staged ray cast (variant 4) the NEXT rover_set_knn_map calls build — bit 0 the f32
 *        proof (ray_precision 0 / 1), bit 1 the as-shipped fp16 proof (ray_precision 2); default 3.  About 4.3 KB per cell, map and proof
 *        at K = 200 (1.56 GB at 600 x 600 cells).  Tables that are not asked for, or that do not fit (the allocation failure is absorbed:
 *        the culled kernel, variant 3, then runs), leave variant 4 unavailable for that arithmetic: the auto choice never picks it, and
 *        asking for it by name ("raycast_variant" 4) is an error (ROVER_E_STATE / ROVER_E_NOMEM), never a silent change of kernel.
 * name = "raycast_run": sorted rays per wave for variants 2 and 3 (default 0 = auto: 32 on full batches, down to 4 on small
 *        ones; variant 3 caps it at 64). */
ROVER_API int rover_set_option(rover_ctx *ctx, const char *name, int64_t value);

/* ---- introspection (bench / roofline) ---------------------------------------------------------------- */
typedef struct {
    int32_t P, Ns, Nd, rays_per_env_padded;
    int32_t K[2], K8[2], X[2], Y[2];
    uint64_t table_bytes[2];       /* re-packed per-cell fp16 tables + cull tables */
    uint64_t workspace_bytes;
    int32_t raycast_variant;       /* the variant the next step will run: 1 env order, 2 binned, 3 culled, 4 staged (csrc/rover_cull.hip) */
    int32_t cell_index_mode;       /* option "cell_index_mode" in force: 0 cpu_div, 1 cuda_rcp (tests pin the mode a fixture was captured with) */
    int32_t ray_precision;         /* option "ray_precision" in force */
    int32_t raycast_sorted;        /* 1: the step sorts the rays by (map, cell) bin; 0: the ray cast walks the slots in env order */
    int32_t raycast_rocks_staged;  /* variant 4: 1 = the rocks part of the sorted list runs on the staged kernel too, 0 = on the culled one */
} rover_info;
ROVER_API int rover_get_info(const rover_ctx *ctx, rover_info *info);
/* Diagnostics of the culled ray cast (variant 3, csrc/rover_cull.hip).  Per map: how many triangles its conservative
 * rejection test can never reject (slivers, non-finite vertices: stored with a zero normal = "always a candidate") and how
 * many cells have no normal cone (their rays run both tests on every pair).  Of the LAST culled launch on this ctx: rays
 * scanned, (ray, lane-pair) candidates handed to the exact arithmetic (camera.py:84-117 evaluated K per ray), rays that ran
 * both tests, (map, cell) bins walked.  Synchronises the device (a test / bench call, not a step call). */
typedef struct {
    int64_t triangles[2];
    int64_t always_candidate_triangles[2];
    int64_t cells_without_cone[2];
    uint64_t rays, candidate_pairs, rays_both_tests, bins;
    uint64_t max_pairs_per_run;    /* most queue entries any one run of sorted rays produced */
    uint64_t queue_bytes;          /* size of the candidate queue allocation */
    uint64_t launches_per_step;    /* 1, unless the queue budget ("cull_queue_mb") forces a step's ray cast into slices */
    uint64_t rays_far_skipped;     /* rays whose scan skipped the farther half of their cell's triangles (proved clear as a group) */
    int64_t cells_with_far_bound[2];  /* per map: cells whose far bound is wide enough to hold for a usual ray (f32 proof tables) */
    uint64_t far_records_on_demand;   /* 1: the scan kernel in use fetches a bin's far records only when one of its rays tests them, and does not scan rays that clear their whole cell (rays_not_scanned) */
    uint64_t rays_not_scanned;        /* rays that cleared BOTH halves of their cell's triangles as groups (no candidate: the distance is the miss value) */
    uint64_t lane_items, lane_flushes;   /* staged ray cast (variant 4): (ray, chunk of 8 pairs) items tested, exact-phase rounds of runs (saturating at 63 per wave) */
} rover_cull_info;
ROVER_API int rover_get_cull_info(rover_ctx *ctx, rover_cull_info *out);
/* In-situ kernel timing: when enabled, rover_step / rover_get_observations bracket the ray-cast launch with
 * hipEvents on the caller's stream (ring of 256 pairs).  rover_get_profile synchronises those events and
 * returns the summed ray-cast time and the number of TIMED launches since the last rover_set_profiling(ctx, n > 0).
 * enable = n > 1 times every n-th launch only (the first one included): an event pair costs the stream ~12 us around
 * the kernel it brackets (two 6 us bubbles on MI355X), which a caller timing its own steps may not want in each of them. */
typedef struct { double raycast_ms; int32_t launches; uint64_t pairs_per_launch; } rover_profile;
ROVER_API int rover_set_profiling(rover_ctx *ctx, int32_t enable);
ROVER_API int rover_get_profile(rover_ctx *ctx, rover_profile *out);
/* Launch ONLY the ray-cast kernel on the ray records left by the last step (for hipEvent timing of the
 * roofline kernel in isolation; results are rewritten identically). */
ROVER_API int rover_replay_raycast(rover_ctx *ctx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_STEP_H */
