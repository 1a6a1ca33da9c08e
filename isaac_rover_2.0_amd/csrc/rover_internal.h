// rover_internal.h — structs shared by the kernels (rover_kernels.hip) and the C-ABI layer (rover_capi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Test (B) for a whole set of triangles by its normal cone (rover_cull.hip, header comment): the ray-side constant of the f32 proof.
#define ROVER_CONE_TAU 1.25e-2

namespace rover {

// One ray as the ray-cast kernel consumes it (32 B, two 16-byte loads).
struct RayRec {
    float sx, sy, sz;    // origin
    uint32_t cell;       // ix * Y + iy into the map the flags select
    float dx, dy, dz;    // -normalize(direction), ray_casting.py:31
    uint32_t flags;      // bit0: rocks map, bit1: valid; bits 16..31: the ray's normal-cone bound (rover_cull.hip), 0xffff = none
};
static_assert(sizeof(RayRec) == 32, "RayRec must be 32 bytes");

// One re-packed KNN map: per-cell contiguous fp16 block [X*Y][9][K8]
struct KnnDev {
    const uint16_t* table;
    int32_t X, Y, K, K8;         // K8: row pitch of one component inside a cell block (K rounded up to a multiple of 8)
    float cell, shift_x, shift_y;
    float inv_cell;              // 1.0f / cell rounded to f32: the factor of cell_index_mode 1 (cuda_rcp)
};

struct HeightDev {
    const float* hm;
    int32_t N0, N1;
    float hscale, vscale, shift_x, shift_y;
    float inv_hscale;            // 1.0f / hscale (cell_index_mode 1)
    int32_t rcp;                 // cell_index_mode
};

// stone-occupancy grid (built on the host at rover_set_stones): CSR lists of the stones that can matter per grid cell
struct StoneGridDev {
    const uint32_t* cell_start;  // [nx*ny + 1]
    const float4* stone_xyr;     // per list entry: (x, y, radius, stone id as float bits) — one independent 16-B load per stone
    float x0, y0, inv_cell;
    int32_t nx, ny;
};

struct PrepArgs {
    uint32_t E, P, R8;
    const float *pos, *quat, *joints, *target;
    const double* dist;          // [P][3]
    KnnDev terrain, rocks;
    RayRec* rays;                // [E*R8]
    float *euler, *heading;      // [E,3], [E]
    uint32_t* bin_out;           // optional [E*R8]: bin = (map, cell) key of every slot for the bucket sort (binned ray cast)
    uint32_t rocks_bin_offset;   // first bin of the rocks map (= terrain X*Y)
    int32_t precision;           // 0 fp32 mode; 1 fp16-rounded ray origins / directions; 2 as shipped (fp16 ray maths too)
    int32_t cell_rcp;            // cell_index_mode: 0 = (v - shift) / cell (ATen CPU), 1 = (v - shift) * (1 / cell) (ATen CUDA)
    const float* euler_in;       // optional [E,3]: the pose's euler angles as given (rover_get_depths) instead of quat -> euler
    // optional: the first pass of the bucket sort (bin_hist_fused) — counts[tile][bucket] += the block's keys per coarse bucket
    uint32_t* hist;              // [tiles][hist_buckets], all zero when the kernel starts (bucket_sort_kernel clears it again)
    uint32_t hist_low_bits, hist_buckets, hist_blocks_per_tile;     // a tile of the sort = the keys of this many 64-env blocks
};

// constants of the culled ray cast's rejection proof for the as-shipped fp16 arithmetic, derived from its one free parameter eta
struct CullProofH { double kappa, c_rho; float c_a, tau2; };
CullProofH cull_proof_h(double eta, double split);

struct LaneTables { float4* lvl; uint4* lrec; uint2* lid; };      // the staged ray cast's tables of one map for one proof (null: not built)

// raycast_culled_kernel (rover_cull.hip)
struct CullArgs {
    const RayRec* rays;
    const uint32_t* sorted;      // ray slots sorted by (map, cell)
    uint32_t n_sorted;
    uint32_t n_terrain;          // the first n_terrain sorted rays are terrain rays (bins are (map, cell): terrain first)
    const int32_t *idx0, *idx1;  // [cell][K8/4][4] triangle ids of the cell (-1 = empty slot)
    const uint4 *ctab0, *ctab1;  // [T] 16 B: bounding-sphere centre + scaled unit normal per triangle (phase 1)
    const uint16_t *rtab0, *rtab1; // [T] 20 B: the triangle's nine fp16 vertex components (exact arithmetic, phase 2)
    uint32_t kp0, kp1, run, n_blocks;
    float* out;                  // [E*R8] distances
    uint2* queue;                // candidate queue: one region of CULL_QCAP 8-byte entries per wave of a launch
    uint64_t queue_entries;      // its size: a step whose regions would not fit is cast in several launches
    int half;                    // the exact phase runs the reference's as-shipped fp16 arithmetic (ctab / qrow are then that proof's tables)
    float c_a_h, tau2_h;         // test constants of that proof (CullProofH)
    const float4 *far0, *far1;   // [cell][2]: the bound of the cell's far pairs (FarRec, rover_cull.hip) for the proof in force
    const float4 *near0, *near1; // [cell]: the same bound for the cell's near pairs (behind the far records in the same allocation)
    float k2_far;                // and the ray-side constant of the far skip
    int lazy_far;                // set up a bin's far pairs only when one of its rays tests them (few rays per bin)
    int skip_clear;              // (eager kernel) do not scan rays that clear their whole cell — the on-demand kernel always does
    uint4* stats;                // [waves] per-wave counters of the last launch {queue entries, rays, rays with both tests, bins}
};
uint32_t cull_stat_slots(uint64_t n_rays, uint32_t run);
// the staged ray cast (raycast variant 4, rover_cull.hip): lane = (ray, chunk of 8 pairs) over per-cell record rows in group-bound order
struct LaneArgs {
    const RayRec* rays;
    const uint32_t* sorted;
    uint32_t n_sorted, n_terrain;
    const float4* lvl[2];        // per map [cell][lane_lvl_stride()]: header, 16 suffix bounds (fp16), the suffixes' cones
    const uint4* lrec[2];        // [cell][2][pp]: pair records of test (A), then of test (B), in G order
    const uint2* lid[2];         // [cell][pp]: the pairs' triangle ids
    const uint16_t* rtab[2];
    uint32_t pp[2], run;
    float* out;
    uint4* stats;
    int half;                    // the tables are the as-shipped fp16 arithmetic's, the exact phase runs it
    float c_a_h, k2_far;         // test (A)'s constant of that proof; the level bound's ray-side constant (cull_far_k2)
};
uint32_t lane_pairs_per_row(uint32_t K8);
uint32_t lane_lvl_stride();
uint32_t lane_waves(uint32_t n_rays, uint32_t run);
hipError_t launch_raycast_lane(LaneArgs a, hipStream_t s);
uint64_t cull_queue_entries(uint64_t n_rays, uint32_t n_terrain, uint32_t run, uint64_t budget_bytes, uint32_t* n_launches);

// n / d for every 32-bit n with a multiply-high and two shifts (Granlund & Montgomery's round-up method): the compiler's own
// expansion of a division by a run-time value is ~30 (32-bit) / ~110 (64-bit) dependent instructions per thread.
struct FastDiv {
    uint32_t m, s1, s2;
#if defined(__HIPCC__)
    __device__ __forceinline__ uint32_t div(uint32_t n) const {
        const uint32_t t = __umulhi(m, n);
        return (t + ((n - t) >> s1)) >> s2;
    }
#endif
};
inline FastDiv make_fastdiv(uint32_t d) {           // d >= 1
    uint32_t L = 0;
    while (L < 32 && (1ull << L) < d) ++L;
    FastDiv f;
    f.m = (uint32_t)((((1ull << L) - d) << 32) / d + 1ull);
    f.s1 = L < 1 ? L : 1; f.s2 = L > 1 ? L - 1 : 0;
    return f;
}

struct ObsArgs {
    uint32_t E, W, R8;
    FastDiv w_div;               // by W (filled in by launch_assemble_obs)
    int64_t obs_stride;
    const float *pos, *target, *heading, *lin_hist, *ang_hist, *dist;
    int32_t fp16_div;            // as-shipped mode: round dist / 2 to fp16
    const int32_t* obs_idx;      // [Ns+Nd] ray index per heightmap column
    float* obs;
};

struct MetricsArgs {
    uint32_t E, R8;
    int32_t curriculum_level, max_episode_length;
    int64_t num_envs_global;
    int do_increment, do_collision, do_metrics, do_done;
    float pos_reward, heading_contraint_reward, motion_contraint_reward, goal_angle_reward, boogie_contraint_reward;
    float wheel_thr, body_thr;   // rover.py:667-668 thresholds (0.8 / 0.45; their fp16 roundings in the as-shipped mode)
    const float *pos, *target, *joints, *lin_hist, *ang_hist, *euler_pre, *heading, *dist;
    int64_t* progress;
    int64_t* rock_collision;
    float* rew;
    int64_t* reset;
    uint32_t* block_cnt;         // optional [ceil(E/256)]: per-block done count for the compaction
    float* ex_pos_reward; int64_t* ex_collision; float *ex_upright, *ex_heading, *ex_motion, *ex_goal_angle, *ex_lin, *ex_ang;
    uint8_t* done_u8;            // optional: reset != 0 as one byte per env (the form that travels in the multi-GPU gather)
    int64_t* stone_collision;    // optional additional output: stone_info occupancy mask at pos_xy (collision stage)
    float stone_margin;
    StoneGridDev sgrid;
    const float* info7;
};

struct ResetArgs {
    const int64_t* ids;          // compacted reset ids (global)
    int64_t id_offset;           // env_offset: global -> local
    uint32_t n_host;
    const int32_t* n_dev;        // optional: count in device memory
    const float* initial_pos3;
    float *pos3, *quat4, *joint_pos13, *joint_vel13, *base_pos3;
    int64_t *reset, *progress;
    const int32_t* yaw_deg;      // optional [n]
    uint64_t seed;
    const uint64_t* seed_dev;    // optional [1]: added to seed on the device (a captured graph's step counter)
};

struct GoalArgs {
    const float* info7; uint32_t S; HeightDev h; StoneGridDev grid;
    const int64_t* env_ids; int64_t id_offset;
    uint32_t n_host; const int32_t* n_dev;
    const float* initial_pos3; float* target3; float radius;
    const float* draws; int32_t max_draws; uint64_t seed;
    const uint64_t* seed_dev;    // optional [1]: added to seed on the device
    int32_t* t_acc;              // [n] scratch: iteration at which entry i was accepted (-1: id 0 from the start)
    int32_t* n_draws_used;
};

struct PrePhysicsArgs {
    uint32_t E;
    const float *actions, *quat;
    float *lin_hist, *ang_hist, *euler_pre, *pos_targets13, *vel_targets13;
    float* actions_nn;           // optional [E,2,3]: self.actions_nn (rover.py:366), newest first
};

struct LinearArgs {
    const float* x; int64_t x_stride;      // [M, K] rows at x_stride floats
    const float* w; const float* b;        // nn.Linear: weight [N][K], bias [N] (optional)
    float* y; int64_t y_stride;            // [M, N] rows at y_stride floats
    int32_t M, K, N, act;                  // act: 0 none, 1 leakyrelu(0.01), 2 tanh, 3 relu, 4 elu
};
hipError_t launch_linear_act(const LinearArgs& a, hipStream_t s);

// a chain of 2 or 4 layers in one kernel (rover_mlp.hip): y = L_n(... L_1(x)), L_i(v) = act_i(W_i v + b_i)
struct ChainArgs {
    const float* x; int64_t x_stride;      // [M, K0] rows at x_stride floats
    int32_t M, K0, n_layers;
    int32_t n[4];                          // output widths
    const float* w[4]; const float* b[4];  // nn.Linear: weight [n_i][n_{i-1}] (row-major, K0 for the first), bias [n_i] or NULL
    int32_t act[4];
    float* y; int64_t y_stride;            // [M, n_last] rows at y_stride floats
};
hipError_t launch_chain(const ChainArgs& a, hipStream_t s);
// small batches: a 2-layer chain with its first layer split along k (rover_mlp.hip); scratch holds chain_splitk_scratch_floats() floats
bool chain_wants_splitk(const ChainArgs& a);
size_t chain_splitk_scratch_floats(int M, int K0, int n0);
hipError_t launch_chain_splitk(const ChainArgs& a, float* scratch, hipStream_t s);
// two such chains over the same rows in one launch per stage (+ an optional column copy), when chain_pair_fits()
bool chain_pair_fits(const ChainArgs& a, const ChainArgs& b);
hipError_t launch_chain_splitk_pair(const ChainArgs& a, const ChainArgs& b, float* scratch_a, float* scratch_b, const float* copy_src,
                                    int64_t copy_src_stride, float* copy_dst, int64_t copy_dst_stride, int copy_cols, hipStream_t s);

hipError_t launch_repack(const int32_t* map_idx, const int32_t* tris, const uint16_t* verts, uint64_t n_cells, uint32_t K,
                         uint32_t K8, uint32_t T, uint32_t V, uint16_t* table, hipStream_t s);
hipError_t launch_prep(const PrepArgs& a, hipStream_t s);
hipError_t launch_raycast(const RayRec* rays, uint32_t n_rays, const uint16_t* tab0, const uint16_t* tab1, uint32_t kp0,
                          uint32_t kp1, float* out, hipStream_t s);
// true when prep_rays_kernel can count the sort's coarse buckets itself (PrepArgs::hist): a 64-env block's keys lie inside one tile of the sort
bool bin_hist_fused(uint32_t n_slots, uint32_t R8, uint32_t n_bins, uint32_t low_bits, uint32_t* blocks_per_tile);
hipError_t launch_bin_rays(const uint32_t* bins, uint32_t n_slots, uint32_t n_valid, uint32_t n_bins, uint32_t low_bits,
                           uint32_t* table, uint2* pairs, uint32_t* block_sums, uint32_t* sorted, bool hist_done, hipStream_t s);
hipError_t launch_raycast_binned(const RayRec* rays, const uint32_t* sorted, uint32_t n_sorted, const uint16_t* tab0,
                                 const uint16_t* tab1, uint32_t kp0, uint32_t kp1, uint32_t run, bool fp16_math, uint32_t early_out, float* out,
                                 hipStream_t s);
hipError_t launch_tri_centroids(const int32_t* tris, const uint16_t* verts, uint32_t T, uint32_t V, float2* out, hipStream_t s);
hipError_t launch_cull_build(const int32_t* map_idx, const int32_t* tris, const uint16_t* verts, uint64_t n_cells, uint32_t K,
                             uint32_t K8, uint32_t T, uint32_t T_int, uint32_t V, const uint32_t* order, const uint32_t* newid,
                             int32_t* idx4, uint4* ctab, uint4* ctab_h, uint16_t* rtab, uint32_t* qrow, uint32_t* qrow_h, float4* far,
                             float4* far_h, float* nz_scratch, uint32_t* counts, CullProofH ph, uint32_t Y, float cell_size, float shift_x,
                             float shift_y, LaneTables lane, LaneTables lane_h, hipStream_t s);
float cull_far_k2(int half, CullProofH ph);
hipError_t launch_raycast_culled(CullArgs a, hipStream_t s);
hipError_t launch_knn_centroids(const float* verts, const int32_t* tris, uint32_t T, uint32_t V, int ref, float* cx, float* cy,
                                hipStream_t s);
hipError_t launch_knn_bucket(const float* cx, const float* cy, uint32_t T, float ox, float oy, float inv_g, uint32_t nbx, uint32_t nby,
                             uint32_t* cursor, uint32_t* items, int count, hipStream_t s);
hipError_t launch_scan_exclusive(uint32_t* data, uint32_t n, uint32_t* block_sums, hipStream_t s);
hipError_t launch_knn_select(const float* cx, const float* cy, const uint32_t* bucket_start, const uint32_t* items, float ox, float oy,
                             float g, uint32_t nbx, uint32_t nby, uint32_t X, uint32_t Y, float res, uint32_t K, const float* cell_x,
                             const float* cell_y, int32_t* out, int32_t* overflow, hipStream_t s);
hipError_t launch_assemble_obs(const ObsArgs& a, hipStream_t s);
hipError_t launch_export_dist(const float* dist, const RayRec* rays, uint32_t E, uint32_t R8, uint32_t P, int precision, float* ray_dist,
                              float* wheel, float* body, float* ray_src, float* hit_pt, hipStream_t s);
hipError_t launch_export_rays(const RayRec* rays, const float* dist, uint32_t E, uint32_t R8, uint32_t P, float* src, float* dir, int32_t* cell,
                              float* out_dist, hipStream_t s);
hipError_t launch_import_rays(const float* src, const float* dir, uint32_t E, uint32_t R8, uint32_t P, const KnnDev& terrain, const KnnDev& rocks,
                              uint32_t rocks_bin_offset, int precision, int cell_rcp, RayRec* rays, uint32_t* bin_out, hipStream_t s,
                              uint32_t* not_unit = nullptr /* optional: counts the finite directions whose length is not 1 */);
hipError_t launch_obs_metrics(const ObsArgs& o, const MetricsArgs& m, hipStream_t s);     // both in one launch (rover_step)
hipError_t launch_metrics_done(const MetricsArgs& a, hipStream_t s);
hipError_t launch_compact(const int64_t* reset, uint32_t n, int64_t offset, uint32_t* block_cnt, bool counted, int64_t* ids,
                          int32_t* count, hipStream_t s);
hipError_t launch_quat_to_euler(const float* q, float* eul, uint32_t n, hipStream_t s);
hipError_t launch_clearance(const float* info7, uint32_t S, const float* xy, uint32_t n, float* out, hipStream_t s);
hipError_t launch_shift_spawns(const StoneGridDev& g, const float* info7, float* pos3, uint32_t n, int32_t max_iter, hipStream_t s);
hipError_t launch_sample_height(const HeightDev& h, const float* xy, uint32_t n, float* out, hipStream_t s);
hipError_t launch_generate_goals(const GoalArgs& a, uint32_t n_max, hipStream_t s);
hipError_t launch_reset_envs(const ResetArgs& a, uint32_t n_max, hipStream_t s);
hipError_t launch_pre_physics(const PrePhysicsArgs& a, hipStream_t s);
hipError_t launch_ackermann(const float* lin, const float* ang, uint32_t n, float* steer, float* vel, hipStream_t s);

}  // namespace rover
