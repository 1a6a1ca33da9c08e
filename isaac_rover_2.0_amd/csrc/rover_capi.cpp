// rover_capi.cpp — C-ABI layer of librover_step.so (include/rover_step.h): context, library-owned device
// tables, argument checks, kernel sequencing.  No torch, no exceptions across the boundary.
#include "../../include/rover_step.h"
#include "rover_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace rover;

struct rover_ctx {
    rover_cfg cfg{};
    std::string err;
    // re-packed maps
    KnnDev map[2]{};
    uint64_t table_bytes[2]{0, 0};
    bool have_map[2]{false, false};
    // tables of the culled ray cast (variant 3): the cell's triangle ids, and per triangle a bounding-sphere centre + scaled
    // unit normal (16 B) and the nine fp16 vertex components (20 B)
    int32_t* cull_idx[2]{nullptr, nullptr};
    uint4* cull_ctab[2]{nullptr, nullptr};
    uint4* cull_ctab_h[2]{nullptr, nullptr};    // the same for the as-shipped fp16 arithmetic's rejection proof (ray_precision 2)
    uint32_t* cull_qrow_h[2]{nullptr, nullptr};
    float4* cull_far[2]{nullptr, nullptr};      // [cell][2] far-pair bounds (f32 proof / fp16 proof)
    float4* cull_far_h[2]{nullptr, nullptr};
    uint16_t* cull_rtab[2]{nullptr, nullptr};
    uint32_t* cull_qrow[2]{nullptr, nullptr};
    uint64_t cull_bytes[2]{0, 0};
    // tables of the staged ray cast (variant 4; f32 proof): per cell the pair records in group-bound order, their ids, the suffix bounds
    LaneTables lane[2]{}, lane_h[2]{};  // per map: f32 proof / as-shipped fp16 proof
    uint32_t lane_pp[2]{0, 0};
    hipStream_t side = nullptr;         // variant 4: the rocks part's launch runs beside the terrain part's, on this stream, between two events
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int lane_side_stream = 0;           // ROVER_LANE_SIDE_STREAM=1: side by side (measured: 125 against 127 M env-steps/s one after the other — the two launches
                                        // slow each other down by more than the rocks launch's ramp and tail cost; kept as a switch)
    int lane_env_order = -1;            // variant 4 without the sort (the ray slots in env order): -1 auto (mid-size batches), 0 / 1 (option "lane_env_order")
    int staged_tables = 3;              // which proofs' staged-kernel tables rover_set_knn_map builds: bit 0 the f32 proof, bit 1 the as-shipped fp16 one (option
                                        // "staged_tables", before the maps are set; ~4.3 KB per cell, K = 200, map and proof)
    int lane_rocks = -1;                // variant 4: the rocks part of the sorted list through the staged kernel too: -1 auto, 0 / 1 (option "lane_rocks", ROVER_LANE_ROCKS)
    uint2* d_cull_queue = nullptr;      // candidate queue of the culled ray cast: one region of 1 024 entries per wave of a launch
    uint64_t cull_entries = 0;
    uint4* d_cull_stats = nullptr;      // per-wave counters of the last culled launch (rover_get_cull_info)
    uint32_t cull_stat_slots = 0;
    uint64_t stats_sig = 0;             // how the last launch that wrote the counters cast its rays (run_raycast)
    int64_t cull_always[2]{0, 0}, cull_nocone[2]{0, 0}, cull_tris[2]{0, 0};      // per map, counted when its tables were built
    int64_t cull_always_h[2]{0, 0}, cull_nocone_h[2]{0, 0};
    int64_t cull_farok[2]{0, 0}, cull_cells[2]{0, 0};      // cells whose far bound can hold for a usual ray (far_build_kernel) / cells
    double cull_eta_h = 0.08;           // free parameter of the fp16 proof (rover_cull.hip, cull_proof_h); ROVER_CULLH_ETA for experiments
    double cull_split_h = 8.0;          // how test (A)'s cross term is split between its |h|^2 and rho^2 parts (cull_proof_h); ROVER_CULLH_SPLIT
    uint64_t cull_budget = 1536ull << 20;  // option "cull_queue_mb": most bytes the queue may take (a step is cast in several launches beyond it)
    uint32_t cull_launches = 1;
    int cull_lazy = -1;                 // ROVER_CULL_LAZY: -1 auto, 0 / 1 force (experiments)
    uint32_t cull_run = 0;              // run length the queue was sized for
    // distribution
    double* d_dist = nullptr;       // [P][3]
    int32_t* d_obs_idx = nullptr;   // [Ns+Nd]
    int32_t P = 0, Ns = 0, Nd = 0;
    bool have_dist = false;
    // heightfield / stones
    HeightDev hf{};
    bool have_hf = false;
    float* d_stones = nullptr;
    int32_t S = 0;
    StoneGridDev sgrid{};
    bool have_stones = false;
    // per-step workspace
    uint32_t R8 = 0;
    RayRec* d_rays = nullptr;
    float* d_dist_out = nullptr;    // [E*R8]
    float* d_euler = nullptr;       // [E,3]
    float* d_heading = nullptr;     // [E]
    int64_t* d_ids_work = nullptr;  // [E]
    uint32_t* d_goal_work = nullptr;// [2][E] work lists of generate_goals
    uint32_t* d_block_cnt = nullptr;// [ceil(E/256)]
    // ray binning (raycast variant 2)
    uint32_t* d_bins = nullptr;         // [E*R8] bin key per slot
    uint32_t* d_bkt_table = nullptr;    // [n_buckets * n_blocks] counts -> offsets
    size_t bkt_table_bytes = 0;
    bool bkt_table_dirty = true;        // not known to be all zero (what prep_rays_kernel's fused histogram starts from): a step failed half way
    uint2* d_pairs = nullptr;           // [E*R8] (bin, slot) after the coarse partition
    uint32_t low_bits = 10;             // bins per sort bucket = 2^low_bits, in force (alloc_bins)
    uint32_t low_bits_opt = 0;          // option "bin_low_bits": 0 = chosen by the library, else 8..12
    int precision = 0;                  // option "ray_precision": 0 fp32 mode, 1 fp16 sources, 2 as shipped (fp16 maths)
    uint32_t* d_block_sums = nullptr;   // [4096] bucket totals + [4097] bucket starts
    uint32_t* d_sorted = nullptr;       // [E*R8] ray slots sorted by (map, cell)
    bool defer_obs = false, obs_pending = false;   // rover_step: assemble_obs waits for do_metrics and shares its launch
    ObsArgs pending_obs{};
    uint32_t n_bins = 0;
    int variant = 0;                    // 0 = auto
    int last_variant = 1;
    bool sorted_valid = false;
    uint32_t run = 0;                   // option "raycast_run": 0 = auto (effective_run)
    uint32_t early_out = 1;             // option "raycast_early_out": conservative whole-pair rejection (bit-identical results)
    int32_t cell_rcp = 0;               // option "cell_index_mode": 0 cpu_div (x / 0.1), 1 cuda_rcp (x * (1 / 0.1))
    float* d_mlp_scratch = nullptr;     // partial sums of the split-k small-batch encoder path (rover_mlp_chain_forward)
    size_t mlp_scratch_floats = 0;
    uint64_t workspace_bytes = 0;
    bool ws_ok = false, bins_ok = false;   // false after a failed (re)allocation: the step entry points refuse to run
    bool rays_valid = false;            // the ray workspace holds a finished ray cast (rover_replay_raycast)
    bool obs_valid = false;             // ... and euler / heading hold the state of a rover_get_observations (rover_calculate_metrics reads them)
    // in-situ ray-cast timing (rover_set_profiling)
    bool profiling = false;
    int32_t prof_every = 1;             // time every prof_every-th ray-cast launch (an event pair costs ~12 us of stream time)
    int32_t prof_seen = 0;              // launches since profiling was switched on
    std::vector<hipEvent_t> ev0, ev1;
    int32_t prof_launches = 0;
    double prof_ms = 0.0;
    int32_t prof_pending = 0;
};

static const int kProfRing = 256;

static int prof_drain(rover_ctx* c) {
    for (int i = 0; i < c->prof_pending; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(c->ev1[i]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]);
        if (e != hipSuccess) { c->prof_pending = 0; return ROVER_E_HIP; }
        c->prof_ms += ms;
    }
    c->prof_pending = 0;
    return ROVER_OK;
}

static thread_local std::string g_create_err;

static int fail(rover_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) return fail((c), ROVER_E_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// Makes the ctx's device current for the duration of one entry point and restores the caller's device afterwards
// (the reference's task pins everything to one device, rover.py:90; a library must not change the caller's).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            changed = err == hipSuccess;
        }
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define USE_DEVICE(c)                                                                                          \
    DeviceGuard device_guard__((c)->cfg.device);                                                               \
    if (device_guard__.err != hipSuccess)                                                                      \
        return fail((c), ROVER_E_HIP, "hipSetDevice(%d): %s", (c)->cfg.device, hipGetErrorString(device_guard__.err))

template <typename T>
static void dfree(T*& p) {
    if (p) { (void)hipFree((void*)p); p = nullptr; }
}

static uint64_t valid_rays(const rover_ctx* c) { return (uint64_t)c->cfg.num_envs * (26u + (uint64_t)c->P); }

// Auto choice, measured on MI355X at K = 200, 37 + 26 rays (round 4, one call per size, whole step): the culled ray cast passes the
// env-order kernel between 512 and 1 024 envs (32 k / 64 k rays: 9.8 vs 11.0 M env-steps/s at 512 — the four sort launches cost more than
// the culling saves —, 16.5 vs 14.6 M at 1 024, 25.8 vs 17.0 M at 2 048); in the as-shipped fp16 arithmetic it is ahead of the binned
// kernel from 512 envs on (9.1 vs 8.1, 14.0 vs 12.4, 20.6 vs 17.2 M).  (Until round 4 the switch sat at 131 072 rays: round 2's
// measurement, when the ray cast behind the sort was the every-triangle kernel.)
#define ROVER_AUTO_CULL_RAYS_F32 49152u
#define ROVER_AUTO_CULL_RAYS_F16 24576u
// The staged ray cast: from 24 576 rays (f32 arithmetic), in env order — no sort — while a terrain cell holds fewer than 1.5 heightmap rays
// and 48 cells or more hold one rover (lane_env_order).  Whole step, M env-steps/s, 37 + 26 rays, one call (tools/sweep_small.sh,
// profiles/r05_final_sweep.log), env-order kernel / culled / staged behind the sort / staged in env order: 512 envs 11.4 / 10.5 / 8.2 / 17.0,
// 1 024: 14.7 / 17.6 / 15.4 / 29.3, 2 048: 17.4 / 27.3 / 26.5 / 37.7, 4 096: 18.3 / 39.9 / 41.1 / 49.1, 8 192: 18.9 / 54.9 / 59.2 / 57.7,
// 16 384: 19.2 / 72.2 / 79.7 / 66.9, 32 768: 19.3 / 98.4 / 113.6 / 71.1, 65 536: 19.2 / 121.7 / 142.1 / 73.5; 120 + 26 rays at 4 096 envs
// 8.4 / 26.1 / 30.8 / 35.0, at 65 536 envs 8.5 / 56.2 / 69.9 / 45.4.
#define ROVER_AUTO_LANE_RAYS 24576u
#define ROVER_AUTO_LANE_ENV_RAYS_F16 98304u   // as shipped: below this many rays the staged kernel in env order (lane_env_order) is ahead of the culled one
static bool lane_tables_ok(const rover_ctx* c) {
    const LaneTables* t = c->precision == 2 ? c->lane_h : c->lane;
    return t[0].lrec && t[1].lrec;
}
static int effective_variant(const rover_ctx* c) {
    const bool v2_ok = c->map[0].K8 <= 256 && c->map[1].K8 <= 256;      // 64 lanes x 4 triangles
    if (c->variant == 1 || !v2_ok) return 1;
    const bool v4_ok = lane_tables_ok(c);     // the staged kernel's tables of the proof in force, on both maps
    if (c->variant == 0 && c->precision != 2 && c->have_dist && valid_rays(c) < (v4_ok ? ROVER_AUTO_LANE_RAYS : ROVER_AUTO_CULL_RAYS_F32 + 1u)) return 1;
    if (c->variant == 0 && c->precision == 2 && c->have_dist && valid_rays(c) <= ROVER_AUTO_CULL_RAYS_F16) return 2;      // small batches, as shipped: binned
    // variant 3 (culled): its exact phase runs either arithmetic (f32 / as shipped), each with its own proof tables
    const bool v3_ok = c->cull_idx[0] && c->cull_idx[1];
    if (c->variant == 2 || !v3_ok) return 2;
    // variant 4 (staged, rover_cull.hip: lane = (ray, chunk of 8 pairs) over per-cell record rows): either arithmetic, each with its
    // proof's tables.  Auto: the table above; the native 1 634 + 26 rays at 512 envs on an irregular mesh 2.13 / 2.98 (culled / staged), the
    // irregular mesh at 65 536 envs 83.5 / 113.8.  As shipped (fp16 proof: a third of the rays lie off their cell's narrow cone and test
    // every pair both ways) the culled kernel stays ahead at 37 + 26 rays from 2 048 envs on (23.1 / 22.2 there, 31.7 / 27.1 at 4 096, 54.2 / 46.8 at
    // 16 384, 91.2 / 87.1 at 65 536) but not below (env order, no sort: 512 envs 9.7 / 13.4, 1 024: 15.4 / 18.5, 1 536: 20.0 / 21.8) and not on dense ray
    // sets — ten or more heightmap rays per terrain cell (behind the sort: 120 + 26 rays at 65 536 envs 47.1 / 51.6, the native 1 634 + 26 rays at
    // 4 096 envs 3.65 / 3.86; below ten: native rays at 512 envs 2.25 / 1.95).  On an irregular terrain mesh — fewer than half of the cells
    // with a usable far bound: the culled kernel then scans most cells whole — from two rays per cell: 37 + 26 rays at 4 096 / 16 384 / 65 536
    // envs (0.4 / 1.7 / 6.7 rays per cell) 25.5 / 20.7, 36.1 / 34.3, 48.1 / 58.7; the native rays at 512 envs (2.3) 1.46 / 1.61.
    if (v4_ok && c->variant == 4) return 4;
    if (v4_ok && c->variant == 0 && c->have_dist) {
        if (c->precision != 2) { if (valid_rays(c) >= ROVER_AUTO_LANE_RAYS) return 4; }
        // (round 6 — the rocks part in the staged launch, eta = 0.08, cross-term split 8 —, binned / culled / staged behind the sort / staged in env order,
        //  M env-steps/s at 37 + 26 rays: 512 envs 8.7 / 9.9 / 9.2 / 14.3; 1 536: 15.1 / 20.6 / 21.1 / 23.9; 2 048: 17.5 / 23.7 / 24.3 / 24.8; 4 096: 21.9 / 32.6 / 33.7 /
        //  30.3; 16 384: 39.8 / 55.9 / 57.6 / 41.0; 65 536: 51.0 / 93.3 / 95.4 / 44.7; 120 + 26 rays 22.6 / 45.4 / 51.1 / 27.9; native rays at 4 096 envs 1.95 / 3.57 / 4.04 /
        //  2.46, at 512 envs 1.56 / 2.24 / 2.44 / 2.10: on a regular terrain mesh the staged kernel from the binned kernel's range on.  Irregular terrain
        //  mesh: 4 096 envs 22.1 / 25.4 / 21.6 / 21.3; 16 384: 39.7 / 35.8 / 35.9 / 23.8; 65 536: 49.7 / 47.9 / 59.9 / 25.2; native rays at 512 envs 1.52 / 1.41 / 1.62 /
        //  1.42: staged from two heightmap rays per terrain cell, as before.)
        else if (valid_rays(c) < ROVER_AUTO_LANE_ENV_RAYS_F16 || 2 * c->cull_farok[0] >= c->cull_cells[0] ||
                 (uint64_t)c->cfg.num_envs * (uint64_t)c->P >= 2ull * (uint64_t)c->cull_cells[0]) return 4;
    }
    return 3;
}

// Sorted rays per wave.  Long runs amortise a cell's set-up (65 536 envs, ray cast only: run 12 -> 1.183 ms, 16 -> 1.171,
// 24 -> 1.153, 32 -> 1.144, 64 -> 1.150); small batches need more, shorter waves to fill 256 CUs (4 096 envs: run 4 ->
// 0.2145 ms per step, run 16 -> 0.2264 ms).
static uint32_t effective_run(const rover_ctx* c) {
    if (c->run) return c->run;
    const uint64_t r = valid_rays(c) / 65536u;
    // the culled ray cast (round 3, one call each: 4 096 envs run 4 / 8 / 16 / 32 -> 0.155 / 0.151 / 0.161 / 0.173 ms per step;
    // 8 192 envs 0.246 / 0.227 / 0.236 / 0.249; 16 384 envs 0.407 / 0.335 / 0.331 / 0.343): small batches want many short-lived waves
    // (round 4, on the final kernels, whole step in M env-steps/s at 37 + 26 rays, one call: 1 024 envs run 4 / 8 / 16 / 32 -> 16.0 / 16.5 / 16.2 /
    //  13.4; 2 048: 23.9 / 25.9 / 25.7 / 24.0; 4 096 run 8 / 16 / 32: 37.0 / 37.9 / 36.5; 8 192 run 8 / 16 / 32 / 64: 49.0 / 52.4 / 52.8 / 51.4;
    //  16 384 run 16 / 32 / 64: 68.8 / 71.9 / 72.3; 32 768 run 32 / 64: 96.0 / 97.9; 49 152: 109.1 / 112.5 — since the one-wave workgroups
    //  and the LDS-first queue of round 3 longer runs win earlier than the table above, measured before them, said)
    //  A wave's life is longer where a ray has more candidates — an irregular mesh (8.6 pairs per ray against 3.6), the as-shipped fp16
    //  arithmetic (8.1) — and there shorter runs still balance better (irregular mesh, run 8 / 16 / 32 / 64: 4 096 envs 30.2 / 29.5 / 27.1 / 25.6;
    //  8 192: 39.0 / 39.7 / 37.8 / 36.7; 16 384: 49.1 / 53.0 / 52.7 / 51.3; 32 768: 58.8 / 66.0 / 68.2 / 67.2; 65 536: 66.0 / 77.7 / 82.4 / 81.8;
    //  fp16 at 8 192 envs: 36.3 / 36.0 / 34.8 / 34.1): they keep the older table.  The native ray set on the regular mesh: 512 envs run 16 / 32 /
    //  64 -> 2.56 / 2.61 / 2.46, 1 024 envs 3.10 / 3.33 / 3.32.)
    // the staged kernel behind the sort wants long runs — a chunk read is shared by the run's rays that test it, a round is fuller —
    // (whole step, M env-steps/s, runs of 16 / 32 / 64: 8 192 envs 54.5 / 57.6 / 56.0; 16 384: 68.5 / 77.0 / 79.3; 32 768: 82.8 / 98.1 / 110.5;
    //  65 536: 91.9 / 111.8 / 133.0; 120 + 26 rays 38.5 / 53.9 / 65.0; irregular mesh 67.8 / 92.4 / 104.7)
    if (effective_variant(c) == 4) return r < 12 ? 32u : 64u;
    if (effective_variant(c) >= 3) {
        const bool quick_rays = c->precision != 2 && 2 * c->cull_farok[0] >= c->cull_cells[0];      // regular mesh (most cells have a far bound), f32 arithmetic
        if (quick_rays) return r < 3 ? 8u : (r < 6 ? 16u : (r < 20 ? 32u : 64u));                   // powers of two: 63 instead of 64 cost 6 %
        return r < 12 ? 8u : (r < 24 ? 16u : (r < 48 ? 32u : 64u));
    }
    return (uint32_t)(r < 4 ? 4 : (r > 32 ? 32 : r));
}

static uint32_t bucket_count(const rover_ctx* c) { return (c->n_bins + (1u << c->low_bits) - 1u) >> c->low_bits; }

static int alloc_cull_queue(rover_ctx* c);

static int alloc_bins(rover_ctx* c) {
    c->bins_ok = false;
    if (!c->have_map[0] || !c->have_map[1]) return ROVER_OK;
    const uint64_t nb = (uint64_t)c->map[0].X * c->map[0].Y + (uint64_t)c->map[1].X * c->map[1].Y;
    if (nb > 0xfffffffeull) return fail(c, ROVER_E_INVALID, "too many map cells for ray binning");
    c->n_bins = (uint32_t)nb;
    c->low_bits = c->low_bits_opt ? c->low_bits_opt : 10u;
    while (c->low_bits < 12u && bucket_count(c) > 4096u) ++c->low_bits;
    if (!c->low_bits_opt && c->have_dist) {
        // One-dword sort entries (low bin bits | slot id) need n_slots <= 2^(32 - low_bits).  A dense ray set that misses that at 1 024 bins
        // per bucket (65 536 envs x 152 slots: 24 bits of slot id) sorts faster with fewer bins per bucket and packed entries than with
        // two-dword entries (configs[4]: the four sort passes 158 -> 110 us) — as long as the buckets stay <= 4 096.
        const uint64_t n_slots = (uint64_t)c->cfg.num_envs * c->R8;
        uint32_t lb = c->low_bits;
        while (lb > 8u && n_slots > (1ull << (32u - lb)) && ((c->n_bins + (1u << (lb - 1u)) - 1u) >> (lb - 1u)) <= 4096u) --lb;
        if (n_slots <= (1ull << (32u - lb))) c->low_bits = lb;
    }
    if (!c->d_block_sums) HIP_TRY(c, hipMalloc((void**)&c->d_block_sums, (2 * 4096 + 8) * sizeof(uint32_t)));   // bucket totals + bucket starts
    if (c->have_dist) {                                   // table size depends on E*R8 too
        dfree(c->d_bkt_table);
        const uint64_t n_blocks = ((uint64_t)c->cfg.num_envs * c->R8 + 4095) / 4096;
        c->bkt_table_bytes = ((uint64_t)bucket_count(c) * n_blocks + 1) * sizeof(uint32_t);
        HIP_TRY(c, hipMalloc((void**)&c->d_bkt_table, c->bkt_table_bytes));
        // zero from the start, here and not in the first step: a first step that is only CAPTURED (hipGraph) would record the clearing
        // without running it, and an eager step after it would count into whatever the allocation held
        HIP_TRY(c, hipMemset(c->d_bkt_table, 0, c->bkt_table_bytes));
        c->bkt_table_dirty = false;
        c->bins_ok = true;
    }
    return alloc_cull_queue(c);       // sized here, not in the step: hipMalloc is not allowed while a stream is capturing
}

// candidate queue of the culled ray cast (one bounded region per resident wave) + its per-wave counters, for the options in force
static int alloc_cull_queue(rover_ctx* c) {
    if (!c->ws_ok || !c->have_dist || !c->have_map[0] || !c->have_map[1] || effective_variant(c) < 3) return ROVER_OK;
    const uint32_t run = effective_run(c);
    const uint64_t entries = cull_queue_entries(valid_rays(c), (uint32_t)c->cfg.num_envs * (uint32_t)c->P, run, c->cull_budget, &c->cull_launches);
    // (the per-wave counters are sized by the RAY count, the queue — once capped by the budget — is not: a second
    //  rover_set_distribution with more rays must grow the counters even when the queue keeps its size)
    // (by the PADDED slot count: in env order the staged kernel walks every slot of every env, and its runs are never shorter than `run`)
    const uint32_t slots = rover::cull_stat_slots((uint64_t)c->cfg.num_envs * c->R8, run < 16u ? run : 16u);
    if (c->d_cull_queue && c->d_cull_stats && entries == c->cull_entries && run == c->cull_run && slots == c->cull_stat_slots) return ROVER_OK;
    dfree(c->d_cull_queue); dfree(c->d_cull_stats);
    c->cull_stat_slots = 0;
    c->cull_entries = 0;
    // no fallback to another kernel: a queue that cannot be allocated is an error the caller sees
    hipError_t e = hipMalloc((void**)&c->d_cull_queue, entries * sizeof(uint2));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->d_cull_queue = nullptr;
        return fail(c, ROVER_E_NOMEM, "culled ray cast: candidate queue of %llu bytes: %s", (unsigned long long)(entries * sizeof(uint2)), hipGetErrorString(e));
    }
    c->cull_entries = entries; c->cull_run = run;
    HIP_TRY(c, hipMalloc((void**)&c->d_cull_stats, (size_t)slots * sizeof(uint4)));
    c->cull_stat_slots = slots;
    c->stats_sig = 0;
    HIP_TRY(c, hipMemset(c->d_cull_stats, 0, (size_t)c->cull_stat_slots * sizeof(uint4)));
    return ROVER_OK;
}

static int alloc_workspace(rover_ctx* c) {
    c->ws_ok = false;
    dfree(c->d_rays); dfree(c->d_dist_out); dfree(c->d_euler); dfree(c->d_heading); dfree(c->d_sorted);
    dfree(c->d_bins); dfree(c->d_pairs);
    const uint64_t E = (uint64_t)c->cfg.num_envs;
    c->R8 = (uint32_t)(((26 + c->P) + 7) / 8 * 8);
    const uint64_t n = E * c->R8;
    if (n > 0xffffffffull) return fail(c, ROVER_E_INVALID, "num_envs * rays_per_env = %llu exceeds 2^32", (unsigned long long)n);
    HIP_TRY(c, hipMalloc((void**)&c->d_rays, n * sizeof(RayRec)));
    HIP_TRY(c, hipMalloc((void**)&c->d_dist_out, n * sizeof(float)));
    HIP_TRY(c, hipMalloc((void**)&c->d_euler, E * 3 * sizeof(float)));
    HIP_TRY(c, hipMalloc((void**)&c->d_heading, E * sizeof(float)));
    HIP_TRY(c, hipMalloc((void**)&c->d_sorted, n * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void**)&c->d_bins, n * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void**)&c->d_pairs, n * sizeof(uint2)));
    HIP_TRY(c, hipMemset(c->d_euler, 0, E * 3 * sizeof(float)));
    HIP_TRY(c, hipMemset(c->d_heading, 0, E * sizeof(float)));
    c->workspace_bytes = n * (sizeof(RayRec) + sizeof(float) + 2 * sizeof(uint32_t) + sizeof(uint2)) + E * (4 * sizeof(float) + sizeof(int64_t));
    c->rays_valid = false;
    c->obs_valid = false;
    c->ws_ok = true;
    return alloc_bins(c);
}


// ---- internal triangle numbering of the culled ray cast's tables (rover_set_knn_map) -----------------------------------------
// The caller's triangle order means nothing (a decimated .ply lists triangles in no spatial order), but the kernel gains from
// two properties of the ids it works with: (i) consecutive ids are neighbours in space (one gather instruction of a wave then
// touches few cache lines, and bins that follow each other re-use each other's records in L2), (ii) a lane's packed pair holds two
// triangles that are candidates for the same rays (a queue entry then carries two candidates: fewer entries, fewer exact-phase
// lanes).  So: triangles are matched into spatial partners — mutual nearest centroids first (on a grid mesh exactly the two
// halves of every mesh cell), then greedily the nearest unmatched neighbour — and a pair gets the ids 2p, 2p + 1, pairs in
// Morton order of their first member; an unmatched triangle gets 2p and leaves 2p + 1 a hole.  order[new] = old (0xffffffff =
// hole), newid[old] = new.  A min over the same set of triangles does not depend on the numbering: results are unchanged.
static void cull_numbering(const std::vector<float2>& cen, std::vector<uint32_t>& order, std::vector<uint32_t>& newid) {
    const uint32_t T = (uint32_t)cen.size(), NONE = 0xffffffffu;
    auto finite = [](const float2& p) { return std::fabs(p.x) < 1e30f && std::fabs(p.y) < 1e30f; };
    float bx0 = 0.f, bx1 = 0.f, by0 = 0.f, by1 = 0.f;
    bool any = false;
    uint32_t n_fin = 0;
    for (const float2& p : cen) {
        if (!finite(p)) continue;
        ++n_fin;
        if (!any) { bx0 = bx1 = p.x; by0 = by1 = p.y; any = true; }
        bx0 = p.x < bx0 ? p.x : bx0; bx1 = p.x > bx1 ? p.x : bx1; by0 = p.y < by0 ? p.y : by0; by1 = p.y > by1 ? p.y : by1;
    }
    // Morton rank (16 bits per axis of the bounding box; broken triangles last)
    const double sx = bx1 > bx0 ? 65535.0 / ((double)bx1 - bx0) : 0.0, sy = by1 > by0 ? 65535.0 / ((double)by1 - by0) : 0.0;
    auto spread = [](uint32_t v) {                     // 16 bits -> every second bit of 32
        v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu;
        v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    std::vector<uint64_t> key((size_t)T);
    for (uint32_t t = 0; t < T; ++t) {
        uint32_t m = 0xffffffffu;
        if (finite(cen[t])) m = (spread((uint32_t)(((double)cen[t].x - bx0) * sx)) << 1) | spread((uint32_t)(((double)cen[t].y - by0) * sy));
        key[t] = ((uint64_t)m << 32) | t;
    }
    std::sort(key.begin(), key.end());
    // uniform hash grid over the centroids, ~4 per bucket at mean density
    const double area = ((double)bx1 - bx0 + 1e-6) * ((double)by1 - by0 + 1e-6);
    double h = std::sqrt(area * 4.0 / (double)(n_fin ? n_fin : 1));
    uint32_t gx = (uint32_t)(((double)bx1 - bx0) / h) + 1, gy = (uint32_t)(((double)by1 - by0) / h) + 1;
    while ((uint64_t)gx * gy > (1u << 24)) { h *= 2.0; gx = (uint32_t)(((double)bx1 - bx0) / h) + 1; gy = (uint32_t)(((double)by1 - by0) / h) + 1; }
    auto bucket = [&](const float2& p, uint32_t& ix, uint32_t& iy) {
        ix = (uint32_t)(((double)p.x - bx0) / h); iy = (uint32_t)(((double)p.y - by0) / h);
        ix = ix >= gx ? gx - 1 : ix; iy = iy >= gy ? gy - 1 : iy;
    };
    std::vector<uint32_t> start((size_t)gx * gy + 1, 0), items((size_t)n_fin);
    for (uint32_t t = 0; t < T; ++t) if (finite(cen[t])) { uint32_t ix, iy; bucket(cen[t], ix, iy); ++start[(size_t)ix * gy + iy + 1]; }
    for (size_t b = 0; b < (size_t)gx * gy; ++b) start[b + 1] += start[b];
    {
        std::vector<uint32_t> cur(start.begin(), start.end() - 1);
        for (uint32_t t = 0; t < T; ++t) if (finite(cen[t])) { uint32_t ix, iy; bucket(cen[t], ix, iy); items[cur[(size_t)ix * gy + iy]++] = t; }
    }
    std::vector<uint32_t> partner((size_t)T, NONE);
    // nearest other centroid in the 3 x 3 buckets around t among those for which ok(u); ties by id
    auto nearest = [&](uint32_t t, auto&& ok) {
        uint32_t ix, iy, best = NONE;
        bucket(cen[t], ix, iy);
        double bd = 1e300;
        for (uint32_t a = ix ? ix - 1 : 0; a <= (ix + 1 < gx ? ix + 1 : gx - 1); ++a)
            for (uint32_t b = iy ? iy - 1 : 0; b <= (iy + 1 < gy ? iy + 1 : gy - 1); ++b)
                for (uint32_t k = start[(size_t)a * gy + b]; k < start[(size_t)a * gy + b + 1]; ++k) {
                    const uint32_t u = items[k];
                    if (u == t || !ok(u)) continue;
                    const double dx = (double)cen[u].x - cen[t].x, dy = (double)cen[u].y - cen[t].y, d = dx * dx + dy * dy;
                    if (d < bd || (d == bd && u < best)) { bd = d; best = u; }
                }
        return best;
    };
    {   // pass 1: mutual nearest neighbours
        std::vector<uint32_t> nn((size_t)T, NONE);
        for (uint32_t t = 0; t < T; ++t) if (finite(cen[t])) nn[t] = nearest(t, [](uint32_t) { return true; });
        for (uint32_t t = 0; t < T; ++t) if (nn[t] != NONE && nn[t] > t && nn[nn[t]] == t) { partner[t] = nn[t]; partner[nn[t]] = t; }
    }
    // pass 2, in Morton order: the nearest still unmatched neighbour
    for (uint32_t r = 0; r < T; ++r) {
        const uint32_t t = (uint32_t)key[r];
        if (partner[t] != NONE || !finite(cen[t])) continue;
        const uint32_t u = nearest(t, [&](uint32_t v) { return partner[v] == NONE; });
        if (u != NONE) { partner[t] = u; partner[u] = t; }
    }
    order.clear();
    order.reserve((size_t)T + T / 8);
    std::vector<uint8_t> done((size_t)T, 0);
    for (uint32_t r = 0; r < T; ++r) {
        const uint32_t t = (uint32_t)key[r];
        if (done[t]) continue;
        const uint32_t u = partner[t];
        done[t] = 1;
        newid[t] = (uint32_t)order.size(); order.push_back(t);
        if (u != NONE) { done[u] = 1; newid[u] = (uint32_t)order.size(); order.push_back(u); }
        else order.push_back(NONE);
    }
}

extern "C" {

#ifndef ROVER_SRC_HASH
#define ROVER_SRC_HASH "unknown"
#endif
const char* rover_version(void) { return "rover_step 0.3 (gfx950) src-" ROVER_SRC_HASH; }

const char* rover_last_error(const rover_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int rover_create(const rover_cfg* cfg, rover_ctx** out) {
    if (!cfg || !out) return fail(nullptr, ROVER_E_INVALID, "rover_create: null argument");
    if (cfg->num_envs <= 0) return fail(nullptr, ROVER_E_INVALID, "rover_create: num_envs must be > 0 (got %d)", cfg->num_envs);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, ROVER_E_HIP, "rover_create: no HIP device available (%s)", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, ROVER_E_INVALID, "rover_create: device %d out of range (have %d)", cfg->device, ndev);
    rover_ctx* c = new (std::nothrow) rover_ctx();
    if (!c) return fail(nullptr, ROVER_E_NOMEM, "rover_create: out of host memory");
    c->cfg = *cfg;
    if (c->cfg.num_envs_global <= 0) c->cfg.num_envs_global = c->cfg.num_envs;
    if (c->cfg.max_episode_length <= 0) c->cfg.max_episode_length = 3000;
    if (const char* v = getenv("ROVER_LANE_SIDE_STREAM")) c->lane_side_stream = atoi(v) != 0 ? 1 : 0;
    if (const char* v = getenv("ROVER_LANE_ENV_ORDER")) c->lane_env_order = atoi(v) != 0 ? 1 : 0;
    if (const char* v = getenv("ROVER_LANE_ROCKS")) c->lane_rocks = atoi(v) != 0 ? 1 : 0;
    if (const char* v = getenv("ROVER_RAYCAST_VARIANT")) { int x = atoi(v); c->variant = (x >= 1 && x <= 4) ? x : 0; }
    if (const char* v = getenv("ROVER_CULL_LAZY")) c->cull_lazy = atoi(v);
    if (const char* v = getenv("ROVER_CULLH_ETA")) { const double x = atof(v); if (x >= 0.02 && x <= 0.5) c->cull_eta_h = x; }
    if (const char* v = getenv("ROVER_CULLH_SPLIT")) { const double x = atof(v); if (x >= 0.5 && x <= 64.0) c->cull_split_h = x; }
    if (const char* v = getenv("ROVER_CULL_QUEUE_MB")) { const long mb = atol(v); if (mb >= 1) c->cull_budget = (uint64_t)mb << 20; }
    if (const char* v = getenv("ROVER_BIN_LOW_BITS")) { int b = atoi(v); if (b >= 8 && b <= 12) c->low_bits_opt = (uint32_t)b; }
    if (const char* v = getenv("ROVER_RAYCAST_RUN")) { int r = atoi(v); if (r >= 1 && r <= 4096) c->run = (uint32_t)r; }
    DeviceGuard guard(cfg->device);
    e = guard.err;
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_block_cnt, ((size_t)cfg->num_envs / 256 + 2) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_goal_work, 2 * (size_t)cfg->num_envs * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_ids_work, (size_t)cfg->num_envs * sizeof(int64_t));
    if (e != hipSuccess) { delete c; return fail(nullptr, ROVER_E_HIP, "rover_create: %s", hipGetErrorString(e)); }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (c->side) (void)hipStreamDestroy(c->side);
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        if (c->ev_join) (void)hipEventDestroy(c->ev_join);
        c->side = nullptr; c->ev_fork = nullptr; c->ev_join = nullptr;      // (no side stream: the launches run one after the other)
    }
    *out = c;
    return ROVER_OK;
}

void rover_destroy(rover_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->cfg.device);
    for (int w = 0; w < 2; ++w) { uint16_t* t = const_cast<uint16_t*>(c->map[w].table); dfree(t); dfree(c->cull_idx[w]); dfree(c->cull_ctab[w]); dfree(c->cull_rtab[w]); dfree(c->cull_qrow[w]); dfree(c->cull_ctab_h[w]); dfree(c->cull_qrow_h[w]); dfree(c->cull_far[w]); dfree(c->cull_far_h[w]);
                                  dfree(c->lane[w].lvl); dfree(c->lane[w].lrec); dfree(c->lane[w].lid); dfree(c->lane_h[w].lvl); dfree(c->lane_h[w].lrec); dfree(c->lane_h[w].lid); }
    dfree(c->d_dist); dfree(c->d_obs_idx);
    { float* h = const_cast<float*>(c->hf.hm); dfree(h); }
    dfree(c->d_stones);
    { uint32_t* p = const_cast<uint32_t*>(c->sgrid.cell_start); dfree(p); float4* q = const_cast<float4*>(c->sgrid.stone_xyr); dfree(q); }
    dfree(c->d_rays); dfree(c->d_dist_out); dfree(c->d_euler); dfree(c->d_heading); dfree(c->d_ids_work);
    dfree(c->d_bins); dfree(c->d_bkt_table); dfree(c->d_pairs); dfree(c->d_block_sums); dfree(c->d_sorted);
    dfree(c->d_block_cnt);
    dfree(c->d_goal_work);
    dfree(c->d_cull_queue); dfree(c->d_cull_stats); dfree(c->d_mlp_scratch);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (auto& e : c->ev0) (void)hipEventDestroy(e);
    for (auto& e : c->ev1) (void)hipEventDestroy(e);
    delete c;
}

int rover_set_knn_map(rover_ctx* c, int which, const int32_t* map_idx, int32_t X, int32_t Y, int32_t K, const int32_t* tris,
                      int32_t T, const uint16_t* verts, int32_t V, float cell, float shift_x, float shift_y) {
    if (!c) return ROVER_E_INVALID;
    if (which != ROVER_MAP_TERRAIN && which != ROVER_MAP_ROCKS) return fail(c, ROVER_E_INVALID, "set_knn_map: which=%d", which);
    if (!map_idx || !tris || !verts) return fail(c, ROVER_E_INVALID, "set_knn_map: null table pointer");
    if (X <= 0 || Y <= 0 || K <= 0 || T <= 0 || V <= 0 || !(cell > 0.0f))
        return fail(c, ROVER_E_INVALID, "set_knn_map: bad shape X=%d Y=%d K=%d T=%d V=%d cell=%g", X, Y, K, T, V, (double)cell);
    if ((uint64_t)X * (uint64_t)Y > 0xffffffffull) return fail(c, ROVER_E_INVALID, "set_knn_map: X*Y exceeds 2^32 cells");
    USE_DEVICE(c);
    const uint64_t n_cells = (uint64_t)X * Y;
    const uint32_t K8 = (uint32_t)((K + 7) / 8 * 8);
    const uint64_t bytes = n_cells * 9ull * K8 * sizeof(uint16_t);
    int32_t *d_idx = nullptr, *d_tris = nullptr;
    uint16_t *d_verts = nullptr, *d_table = nullptr;
    auto cleanup = [&]() { dfree(d_idx); dfree(d_tris); dfree(d_verts); };
    hipError_t e;
    if ((e = hipMalloc((void**)&d_idx, n_cells * K * sizeof(int32_t))) != hipSuccess ||
        (e = hipMalloc((void**)&d_tris, (uint64_t)T * 3 * sizeof(int32_t))) != hipSuccess ||
        (e = hipMalloc((void**)&d_verts, (uint64_t)V * 3 * sizeof(uint16_t))) != hipSuccess ||
        (e = hipMalloc((void**)&d_table, bytes)) != hipSuccess) {
        cleanup(); dfree(d_table);
        return fail(c, ROVER_E_NOMEM, "set_knn_map: hipMalloc (%llu B table): %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    if ((e = hipMemcpy(d_idx, map_idx, n_cells * K * sizeof(int32_t), hipMemcpyDefault)) != hipSuccess ||
        (e = hipMemcpy(d_tris, tris, (uint64_t)T * 3 * sizeof(int32_t), hipMemcpyDefault)) != hipSuccess ||
        (e = hipMemcpy(d_verts, verts, (uint64_t)V * 3 * sizeof(uint16_t), hipMemcpyDefault)) != hipSuccess ||
        (e = launch_repack(d_idx, d_tris, d_verts, n_cells, (uint32_t)K, K8, (uint32_t)T, (uint32_t)V, d_table, nullptr)) != hipSuccess ||
        (e = hipDeviceSynchronize()) != hipSuccess) {
        cleanup(); dfree(d_table);
        return fail(c, ROVER_E_HIP, "set_knn_map: %s", hipGetErrorString(e));
    }
    // tables of the culled ray cast (64 lanes x 4 triangles; triangle ids and the map bit share 32 bits of a queue entry)
    int32_t* d_cidx = nullptr;
    uint4 *d_ctab = nullptr, *d_ctab_h = nullptr;
    uint16_t* d_rtab = nullptr;
    uint32_t *d_qrow = nullptr, *d_qrow_h = nullptr;
    float4 *d_far = nullptr, *d_far_h = nullptr;
    float* d_nz = nullptr;
    LaneTables lt{}, lth{};
    const uint32_t lane_pp = lane_pairs_per_row(K8);
    uint32_t* d_cnt = nullptr;
    uint32_t h_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t cull_bytes = 0;
    uint32_t *d_order = nullptr, *d_newid = nullptr;
    if (K8 <= 256 && (uint32_t)T < 0x1ffffffu) {
        const uint64_t b_idx = n_cells * K8 * sizeof(int32_t);
        uint32_t T_int = 0;
        auto drop = [&]() { cleanup(); dfree(d_cidx); dfree(d_ctab); dfree(d_ctab_h); dfree(d_rtab); dfree(d_qrow); dfree(d_qrow_h); dfree(d_far); dfree(d_far_h); dfree(d_nz); dfree(d_cnt);
                            dfree(d_order); dfree(d_newid); dfree(d_table); dfree(lt.lvl); dfree(lt.lrec); dfree(lt.lid); dfree(lth.lvl); dfree(lth.lrec); dfree(lth.lid); };
        // internal triangle numbering (spatial partners get ids 2p, 2p + 1, pairs ordered along a Morton curve): cull_numbering()
        std::vector<uint32_t> order, newid((size_t)T);
        {
            float2* d_cen = nullptr;
            std::vector<float2> cen((size_t)T);
            if ((e = hipMalloc((void**)&d_cen, (uint64_t)T * sizeof(float2))) != hipSuccess ||
                (e = launch_tri_centroids(d_tris, d_verts, (uint32_t)T, (uint32_t)V, d_cen, nullptr)) != hipSuccess ||
                (e = hipMemcpy(cen.data(), d_cen, (uint64_t)T * sizeof(float2), hipMemcpyDeviceToHost)) != hipSuccess) {
                dfree(d_cen); drop();
                return fail(c, ROVER_E_HIP, "set_knn_map: triangle centroids: %s", hipGetErrorString(e));
            }
            dfree(d_cen);
            cull_numbering(cen, order, newid);
        }
        T_int = (uint32_t)order.size();
        if (T_int >= 0x3ffffffu) { drop(); return fail(c, ROVER_E_INVALID, "set_knn_map: too many triangles for the culled ray cast's 26-bit ids"); }
        const uint64_t b_ct = (uint64_t)T_int * sizeof(uint4), b_rt = (uint64_t)T_int * 20u;
        const uint64_t b_lane = n_cells * ((uint64_t)lane_lvl_stride() * sizeof(float4) + (uint64_t)lane_pp * (2 * sizeof(uint4) + sizeof(uint2)));
        // The staged kernel's tables are optional: which proofs get them is the "staged_tables" option, and an allocation that fails drops them
        // (the culled kernel then runs, effective_variant) — unless variant 4 was asked for by name.
        const bool want_f = (c->staged_tables & 1) != 0, want_h = (c->staged_tables & 2) != 0;
        auto lane_alloc = [&](LaneTables& t) -> hipError_t {
            hipError_t le;
            if ((le = hipMalloc((void**)&t.lvl, n_cells * (uint64_t)lane_lvl_stride() * sizeof(float4))) != hipSuccess ||
                (le = hipMalloc((void**)&t.lrec, n_cells * 2ull * lane_pp * sizeof(uint4))) != hipSuccess ||
                (le = hipMalloc((void**)&t.lid, n_cells * (uint64_t)lane_pp * sizeof(uint2))) != hipSuccess) {
                dfree(t.lvl); dfree(t.lrec); dfree(t.lid);
                t = LaneTables{};
                (void)hipGetLastError();
            }
            return le;
        };
        hipError_t le = hipSuccess;
        if (want_f) le = lane_alloc(lt);
        if (want_h && le == hipSuccess) le = lane_alloc(lth);
        if (le != hipSuccess) {
            dfree(lt.lvl); dfree(lt.lrec); dfree(lt.lid); lt = LaneTables{};
            if (c->variant == 4) {
                drop();
                return fail(c, ROVER_E_NOMEM, "set_knn_map: the staged ray cast's tables (%llu B per proof) do not fit and raycast_variant 4 was requested: %s",
                            (unsigned long long)b_lane, hipGetErrorString(le));
            }
        }
        cull_bytes = b_idx + 2 * b_ct + b_rt + 2 * n_cells * sizeof(uint32_t) + 2 * n_cells * 48u + ((lt.lrec ? 1 : 0) + (lth.lrec ? 1 : 0)) * b_lane;
        if ((e = hipMalloc((void**)&d_cidx, b_idx)) != hipSuccess || (e = hipMalloc((void**)&d_ctab, b_ct)) != hipSuccess ||
            (e = hipMalloc((void**)&d_ctab_h, b_ct)) != hipSuccess || (e = hipMalloc((void**)&d_qrow_h, n_cells * sizeof(uint32_t))) != hipSuccess ||
            (e = hipMalloc((void**)&d_rtab, b_rt)) != hipSuccess || (e = hipMalloc((void**)&d_qrow, n_cells * sizeof(uint32_t))) != hipSuccess ||
            (e = hipMalloc((void**)&d_far, n_cells * 48u)) != hipSuccess || (e = hipMalloc((void**)&d_far_h, n_cells * 48u)) != hipSuccess ||
            (e = hipMalloc((void**)&d_nz, (uint64_t)T_int * sizeof(float))) != hipSuccess ||
            (e = hipMalloc((void**)&d_cnt, 8 * sizeof(uint32_t))) != hipSuccess ||
            (e = hipMalloc((void**)&d_order, (uint64_t)T_int * sizeof(uint32_t))) != hipSuccess ||
            (e = hipMalloc((void**)&d_newid, (uint64_t)T * sizeof(uint32_t))) != hipSuccess ||
            (e = hipMemcpy(d_order, order.data(), (uint64_t)T_int * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(d_newid, newid.data(), (uint64_t)T * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemset(d_cnt, 0, 8 * sizeof(uint32_t))) != hipSuccess ||
            (e = launch_cull_build(d_idx, d_tris, d_verts, n_cells, (uint32_t)K, K8, (uint32_t)T, T_int, (uint32_t)V, d_order, d_newid, d_cidx,
                                   d_ctab, d_ctab_h, d_rtab, d_qrow, d_qrow_h, d_far, d_far_h, d_nz, d_cnt, cull_proof_h(c->cull_eta_h, c->cull_split_h), (uint32_t)Y, cell,
                                   shift_x, shift_y, lt, lth, nullptr)) != hipSuccess ||
            (e = hipDeviceSynchronize()) != hipSuccess ||
            (e = hipMemcpy(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost)) != hipSuccess) {
            drop();
            return fail(c, ROVER_E_HIP, "set_knn_map: cull tables (%llu B): %s", (unsigned long long)cull_bytes, hipGetErrorString(e));
        }
    }
    cleanup();
    dfree(d_nz); dfree(d_cnt); dfree(d_order); dfree(d_newid);
    c->cull_always[which] = h_cnt[0]; c->cull_nocone[which] = h_cnt[1]; c->cull_tris[which] = T;
    c->cull_always_h[which] = h_cnt[2]; c->cull_nocone_h[which] = h_cnt[3];
    c->cull_farok[which] = h_cnt[4]; c->cull_cells[which] = (int64_t)n_cells;
    uint16_t* old = const_cast<uint16_t*>(c->map[which].table);
    dfree(old);
    dfree(c->cull_idx[which]); dfree(c->cull_ctab[which]); dfree(c->cull_rtab[which]); dfree(c->cull_qrow[which]);
    dfree(c->cull_ctab_h[which]); dfree(c->cull_qrow_h[which]); dfree(c->cull_far[which]); dfree(c->cull_far_h[which]);
    dfree(c->lane[which].lvl); dfree(c->lane[which].lrec); dfree(c->lane[which].lid);
    dfree(c->lane_h[which].lvl); dfree(c->lane_h[which].lrec); dfree(c->lane_h[which].lid);
    c->lane[which] = lt; c->lane_h[which] = lth; c->lane_pp[which] = lane_pp;
    c->cull_far[which] = d_far; c->cull_far_h[which] = d_far_h;
    c->cull_idx[which] = d_cidx; c->cull_ctab[which] = d_ctab; c->cull_rtab[which] = d_rtab; c->cull_qrow[which] = d_qrow; c->cull_bytes[which] = cull_bytes;
    c->cull_ctab_h[which] = d_ctab_h; c->cull_qrow_h[which] = d_qrow_h;
    c->map[which] = KnnDev{d_table, X, Y, K, (int32_t)K8, cell, shift_x, shift_y, 1.0f / cell};
    c->table_bytes[which] = bytes + cull_bytes;
    c->have_map[which] = true;
    c->rays_valid = false;
    return alloc_bins(c);
}

int rover_set_distribution(rover_ctx* c, const double* pts, int32_t P, const int64_t* sparse_idx, int32_t Ns,
                           const int64_t* dense_idx, int32_t Nd) {
    if (!c) return ROVER_E_INVALID;
    if (!pts || P <= 0 || Ns < 0 || Nd < 0 || (Ns > 0 && !sparse_idx) || (Nd > 0 && !dense_idx))
        return fail(c, ROVER_E_INVALID, "set_distribution: bad arguments (P=%d Ns=%d Nd=%d)", P, Ns, Nd);
    USE_DEVICE(c);
    std::vector<double> hp((size_t)P * 3);
    std::vector<int64_t> hs((size_t)Ns), hd((size_t)Nd);
    HIP_TRY(c, hipMemcpy(hp.data(), pts, hp.size() * sizeof(double), hipMemcpyDefault));
    if (Ns) HIP_TRY(c, hipMemcpy(hs.data(), sparse_idx, hs.size() * sizeof(int64_t), hipMemcpyDefault));
    if (Nd) HIP_TRY(c, hipMemcpy(hd.data(), dense_idx, hd.size() * sizeof(int64_t), hipMemcpyDefault));
    std::vector<int32_t> idx;
    idx.reserve((size_t)Ns + Nd);
    for (int64_t v : hs) idx.push_back((int32_t)v);
    for (int64_t v : hd) idx.push_back((int32_t)v);
    for (int32_t v : idx)
        if (v < 0 || v >= P) return fail(c, ROVER_E_INVALID, "set_distribution: index %d outside [0,%d)", v, P);
    c->have_dist = false;
    dfree(c->d_dist); dfree(c->d_obs_idx);
    HIP_TRY(c, hipMalloc((void**)&c->d_dist, hp.size() * sizeof(double)));
    HIP_TRY(c, hipMemcpy(c->d_dist, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc((void**)&c->d_obs_idx, (idx.size() + 1) * sizeof(int32_t)));
    HIP_TRY(c, hipMemset(c->d_obs_idx, 0, (idx.size() + 1) * sizeof(int32_t)));      // assemble_obs_kernel reads entry 0 from every lane
    if (!idx.empty()) HIP_TRY(c, hipMemcpy(c->d_obs_idx, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    c->P = P; c->Ns = Ns; c->Nd = Nd;
    c->have_dist = true;
    return alloc_workspace(c);
}

int rover_set_heightfield(rover_ctx* c, const float* hm, int32_t N0, int32_t N1, float hscale, float vscale, float sx, float sy) {
    if (!c) return ROVER_E_INVALID;
    if (!hm || N0 <= 0 || N1 <= 0 || !(hscale > 0.0f)) return fail(c, ROVER_E_INVALID, "set_heightfield: bad arguments");
    USE_DEVICE(c);
    float* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, (uint64_t)N0 * N1 * sizeof(float)));
    hipError_t e = hipMemcpy(d, hm, (uint64_t)N0 * N1 * sizeof(float), hipMemcpyDefault);
    if (e != hipSuccess) { dfree(d); return fail(c, ROVER_E_HIP, "set_heightfield: %s", hipGetErrorString(e)); }
    float* old = const_cast<float*>(c->hf.hm);
    dfree(old);
    c->hf = HeightDev{d, N0, N1, hscale, vscale, sx, sy, 1.0f / hscale, c->cell_rcp};
    c->have_hf = true;
    return ROVER_OK;
}

int rover_set_stones(rover_ctx* c, const float* info7, int32_t S) {
    if (!c) return ROVER_E_INVALID;
    if (S < 0 || (S > 0 && !info7)) return fail(c, ROVER_E_INVALID, "set_stones: bad arguments");
    USE_DEVICE(c);
    std::vector<float> h((size_t)S * 7);
    if (S) HIP_TRY(c, hipMemcpy(h.data(), info7, h.size() * sizeof(float), hipMemcpyDefault));
    // stone-occupancy grid: 2 m cells over the stones' bounding box padded by r_max + 1.4 m (the largest threshold the
    // task uses, rover.py:660) + margin; a cell lists the stones whose inflated disc reaches it.
    const float cell = 2.0f, reach = 1.4f + 0.05f;
    float x0 = 0.f, y0 = 0.f, x1 = 1.f, y1 = 1.f, rmax = 0.f;
    bool any = false, poisoned = false;
    for (int s = 0; s < S; ++s) {
        float x = h[7 * s], y = h[7 * s + 1], r = h[7 * s + 6];
        // torch.min propagates NaN (rover.py:538): one NaN stone makes nearest_rock NaN for every query, and
        // NaN <= thr is False -> nothing ever collides.  Reproduced by leaving every cell list empty.
        if (!(x == x) || !(y == y) || !(r == r)) { poisoned = true; continue; }
        if (!any) { x0 = x1 = x; y0 = y1 = y; any = true; }
        x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1; y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
        rmax = r > rmax ? r : rmax;
    }
    const float pad = rmax + reach;
    x0 -= pad; y0 -= pad; x1 += pad; y1 += pad;
    int nx = (int)((x1 - x0) / cell) + 1, ny = (int)((y1 - y0) / cell) + 1;
    if (nx < 1) nx = 1; if (ny < 1) ny = 1;
    if ((int64_t)nx * ny > (int64_t)1 << 24) return fail(c, ROVER_E_INVALID, "set_stones: stone extent too large for the 2 m grid");
    std::vector<std::vector<uint32_t>> lists((size_t)nx * ny);
    for (int s = 0; s < S && any && !poisoned; ++s) {
        float x = h[7 * s], y = h[7 * s + 1], r = h[7 * s + 6];
        if (!(x == x) || !(y == y) || !(r == r)) continue;
        float rr = r + reach;
        int cx0 = (int)((x - rr - x0) / cell), cx1 = (int)((x + rr - x0) / cell);
        int cy0 = (int)((y - rr - y0) / cell), cy1 = (int)((y + rr - y0) / cell);
        cx0 = cx0 < 0 ? 0 : cx0; cy0 = cy0 < 0 ? 0 : cy0; cx1 = cx1 >= nx ? nx - 1 : cx1; cy1 = cy1 >= ny ? ny - 1 : cy1;
        for (int cx = cx0; cx <= cx1; ++cx)
            for (int cy = cy0; cy <= cy1; ++cy) lists[(size_t)cx * ny + cy].push_back((uint32_t)s);   // stone order kept
    }
    std::vector<uint32_t> start((size_t)nx * ny + 1, 0), idx;
    for (size_t k = 0; k < lists.size(); ++k) { start[k + 1] = start[k] + (uint32_t)lists[k].size(); idx.insert(idx.end(), lists[k].begin(), lists[k].end()); }
    std::vector<float4> xyr(idx.size() + 1);
    for (size_t k = 0; k < idx.size(); ++k) {
        const uint32_t sidx = idx[k];
        float fid; memcpy(&fid, &sidx, sizeof fid);
        xyr[k] = float4{h[7 * (size_t)sidx], h[7 * (size_t)sidx + 1], h[7 * (size_t)sidx + 6], fid};
    }
    uint32_t* d_start = nullptr;
    float4* d_idx = nullptr;
    float* d_info = nullptr;
    hipError_t e;
    if ((e = hipMalloc((void**)&d_start, start.size() * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void**)&d_idx, xyr.size() * sizeof(float4))) != hipSuccess ||
        (e = hipMalloc((void**)&d_info, ((uint64_t)S * 7 + 1) * sizeof(float))) != hipSuccess ||
        (e = hipMemcpy(d_start, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_idx, xyr.data(), xyr.size() * sizeof(float4), hipMemcpyHostToDevice)) != hipSuccess ||
        (S && (e = hipMemcpy(d_info, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess)) {
        dfree(d_start); dfree(d_idx); dfree(d_info);
        return fail(c, ROVER_E_HIP, "set_stones: %s", hipGetErrorString(e));        // the previous stone tables stay in place
    }
    dfree(c->d_stones);
    { uint32_t* p = const_cast<uint32_t*>(c->sgrid.cell_start); dfree(p); float4* q = const_cast<float4*>(c->sgrid.stone_xyr); dfree(q); }
    c->d_stones = d_info;
    c->sgrid = StoneGridDev{d_start, d_idx, x0, y0, 1.0f / cell, nx, ny};
    c->S = S;
    c->have_stones = true;
    return ROVER_OK;
}

int rover_set_curriculum_level(rover_ctx* c, int32_t level) {
    if (!c) return ROVER_E_INVALID;
    c->cfg.curriculum_level = level;
    return ROVER_OK;
}

// ---- step ------------------------------------------------------------------------------------------
static int check_ready(rover_ctx* c);
static int effective_variant(const rover_ctx* c);
static int check_precision(rover_ctx* c) {
    if (c->precision == 2 && effective_variant(c) < 2)
        return fail(c, ROVER_E_STATE, "ray_precision 2 (as shipped, fp16 maths) needs ray-cast variant 2 or 3 (K <= 256 on both maps)");
    // a variant asked for by name is the one that runs, or the call fails: the staged kernel needs the tables of the arithmetic in force
    // (K > 256 on a map is the documented exception: every variant then runs as the streaming kernel 1, and rover_get_info says so)
    if (c->variant == 4 && c->map[0].K8 <= 256 && c->map[1].K8 <= 256 && !lane_tables_ok(c))
        return fail(c, ROVER_E_STATE, "raycast_variant 4 (staged) was requested but its tables for ray_precision %d are not there (option "
                                      "staged_tables, or they did not fit when the maps were set)", c->precision);
    return ROVER_OK;
}

static int check_ready(rover_ctx* c) {
    if (!c->have_map[0] || !c->have_map[1]) return fail(c, ROVER_E_STATE, "terrain and rocks maps must be set (rover_set_knn_map)");
    if (!c->have_dist) return fail(c, ROVER_E_STATE, "ray distribution must be set (rover_set_distribution)");
    if (!c->ws_ok || !c->bins_ok) return fail(c, ROVER_E_STATE, "the step workspace is not allocated (an earlier rover_set_* call failed)");
    return check_precision(c);
}

static CullArgs cull_args(const rover_ctx* c, uint32_t n_valid) {
    CullArgs a{};
    a.rays = c->d_rays; a.sorted = c->d_sorted; a.n_sorted = n_valid;
    a.n_terrain = (uint32_t)c->cfg.num_envs * (uint32_t)c->P;
    const bool h = c->precision == 2;     // the as-shipped fp16 arithmetic: its own proof tables, the fp16 exact phase
    a.idx0 = c->cull_idx[0]; a.idx1 = c->cull_idx[1];
    a.ctab0 = h ? c->cull_ctab_h[0] : c->cull_ctab[0]; a.ctab1 = h ? c->cull_ctab_h[1] : c->cull_ctab[1];
    a.rtab0 = c->cull_rtab[0]; a.rtab1 = c->cull_rtab[1];
    a.half = h ? 1 : 0;
    const CullProofH ph = cull_proof_h(c->cull_eta_h, c->cull_split_h);
    a.c_a_h = ph.c_a; a.tau2_h = ph.tau2;
    a.far0 = h ? c->cull_far_h[0] : c->cull_far[0]; a.far1 = h ? c->cull_far_h[1] : c->cull_far[1];
    a.near0 = a.far0 + 2ull * (uint64_t)c->cull_cells[0]; a.near1 = a.far1 + 2ull * (uint64_t)c->cull_cells[1];
    a.k2_far = cull_far_k2(a.half, ph);
    // few rays per (map, cell) bin: most bins have no ray that tests the far pairs, and setting them up lazily halves a bin's set-up
    // (32 768 envs x 63 rays 69.7 -> 76 M env-steps/s, 4 096 envs 26.9 -> 30 M; with 146 rays per env a bin holds 14 rays, nearly every
    // bin needs its far pairs and the second, dependent gather costs 3 %: eager there)
    // The kernel that fetches a bin's far records only when a ray needs them pays where most bins skip them: small ray sets (few rays
    // per bin) on a terrain map whose cells mostly have a useful far bound (a regular grid: all of them; an irregular mesh with
    // triangles that span many cells: few — there the second, dependent round of gathers cost 3 %).
    // (round 4: what decides is the rays per bin, not the size of the ray set — 120 + 26 rays at 16 384 / 4 096 envs hold 5.8 / 2.1 rays per
    //  bin and gain 2.8 / 3.7 % from the on-demand kernel; the estimate is heightmap rays per terrain cell for rovers spread over the map.
    //  The native 1 634-point set is dense — 3.2 rays per bin already at 512 envs — and keeps the eager kernel: -1 ... -5 % otherwise.)
    const bool few_per_bin = c->P <= 260 && (uint64_t)c->cfg.num_envs * (uint64_t)c->P < 8ull * (uint64_t)c->cull_cells[0];
    const bool lazy_auto = (26 + c->P < 100 || few_per_bin) && 2 * c->cull_farok[0] >= c->cull_cells[0];
    a.lazy_far = (c->cull_lazy < 0 ? lazy_auto : c->cull_lazy != 0) ? 1 : 0;
    // rays that clear their whole cell are left out of the scan where some do: a mesh whose cells mostly have a usable bound, and rock
    // rays (the ones that qualify) at least a tenth of the ray set (120 + 26 rays: 12 % of the rays, +3.8 %; the native 1 634 + 26: none)
    a.skip_clear = (2 * c->cull_farok[0] >= c->cull_cells[0] && 26 + c->P <= 260) ? 1 : 0;
    // The as-shipped fp16 arithmetic takes the same two choices since round 4 (its proof's group bound holds less often — 39 % of the rays
    // skip the far pairs, 13 % are not scanned at all, against 84 % / 25 % — but what holds is free: 65 536 envs 85.4 -> 88.3 M
    // env-steps/s, 16 384 envs 50.0 -> 53.3 M, alternating in one call); without the whole-cell skip its kernel stays the eager one.
    if (h && !a.skip_clear) a.lazy_far = 0;
    a.kp0 = (uint32_t)c->map[0].K8; a.kp1 = (uint32_t)c->map[1].K8;
    a.run = effective_run(c);
    a.out = c->d_dist_out;
    a.queue = c->d_cull_queue;
    a.stats = c->d_cull_stats;
    a.queue_entries = c->cull_entries;
    return a;
}

// The staged ray cast needs no bins: where a (map, cell) bin holds a ray or none — small and mid-size batches — the sort's three launches
// (18 us) buy it nothing and it walks the ray slots in env order (a run = 64 consecutive slots: a rover's 37 heightmap rays still share cells).
// Measured (MI355X, 37 + 26 rays, whole step, one call per size): see ROVER_AUTO_ENVORDER_* below.
static bool lane_env_order(const rover_ctx* c, int variant) {
    if (variant != 4) return false;
    if (c->lane_env_order >= 0) return c->lane_env_order != 0;
    // what decides is the heightmap rays per terrain cell (rovers spread over the map): below ~1.5 the sort buys no sharing (4 096 envs x 120
    // rays: 1.37, env order 35.3 against 30.9 M env-steps/s behind the sort; 16 384 x 37: 1.68, 67.1 / 80.1) — and the rovers per cell: from one per
    // 48 cells a cell's rays come from several rovers and only the sort brings them together (8 192 x 37: 57.8 / 59.3; 4 096 x 37: 49.2 / 41.1)
    // (round 6, the rocks part in the staged launch too: behind the sort / env order 4 096 envs 48.9 / 49.9, 8 192 envs 68.6 / 58.3, 120 + 26 rays at
    //  4 096 envs 35.3 / 36.4 — the sort pays from one rover per ~64 cells)
    // (as shipped the staged kernel is the auto choice for small batches in env order and for dense ray sets behind the sort: effective_variant)
    if (c->precision == 2) return c->have_dist && valid_rays(c) < ROVER_AUTO_LANE_ENV_RAYS_F16;
    return c->have_dist && 2ull * (uint64_t)c->cfg.num_envs * (uint64_t)c->P < 3ull * (uint64_t)c->cull_cells[0] &&
           64ull * (uint64_t)c->cfg.num_envs < (uint64_t)c->cull_cells[0];
}

// variant 4 behind the sort: the rocks part of the sorted list through the staged kernel too?  Yes, since round 6, in either arithmetic: f32 — 4-byte
// test-(B) records, one launch 0.31 ms against 0.23 + 0.14; as shipped 91.0 against 87.2 M env-steps/s (and 96.9 with the round's proof constants).
static bool lane_rocks_too(const rover_ctx* c) { return c->lane_rocks < 0 ? true : c->lane_rocks != 0; }

// the ray-cast launch(es) of a step for the variant in force, on the ray records / sorted list in the workspace
static int run_raycast(rover_ctx* c, int variant, uint32_t n_valid, hipStream_t s) {
    const uint32_t E = (uint32_t)c->cfg.num_envs;
    if (variant >= 3 && c->d_cull_stats) {
        // the per-wave counters of rover_get_cull_info: a launch writes the slots of its own waves; when the way the rays are cast changed
        // since the last launch (another kernel, order or run length: another number of waves) the slots are cleared first
        const uint64_t sig = (uint64_t)variant | ((uint64_t)lane_env_order(c, variant) << 8) | ((uint64_t)lane_rocks_too(c) << 9) | ((uint64_t)effective_run(c) << 16) |
                             ((uint64_t)c->precision << 32);
        if (sig != c->stats_sig) {
            HIP_TRY(c, hipMemsetAsync(c->d_cull_stats, 0, (size_t)c->cull_stat_slots * sizeof(uint4), s));
            c->stats_sig = sig;
        }
    }
    if (variant == 4) {
        LaneArgs l{};
        l.rays = c->d_rays; l.sorted = c->d_sorted; l.n_sorted = n_valid; l.n_terrain = E * (uint32_t)c->P;
        const bool lh = c->precision == 2;
        for (int w = 0; w < 2; ++w) {
            const LaneTables& t = lh ? c->lane_h[w] : c->lane[w];
            l.lvl[w] = t.lvl; l.lrec[w] = t.lrec; l.lid[w] = t.lid; l.rtab[w] = c->cull_rtab[w]; l.pp[w] = c->lane_pp[w];
        }
        {
            const CullProofH ph = cull_proof_h(c->cull_eta_h, c->cull_split_h);
            l.half = lh ? 1 : 0; l.c_a_h = ph.c_a; l.k2_far = cull_far_k2(l.half, ph);
        }
        l.run = effective_run(c); l.out = c->d_dist_out; l.stats = c->d_cull_stats;
        if (lane_env_order(c, variant)) {      // every slot (padding included), in env order, one launch
            l.sorted = nullptr; l.n_sorted = E * c->R8; l.n_terrain = l.n_sorted;
            // slots per wave: enough waves to fill 1 024 SIMDs (4 096 envs x 64 slots in runs of 64 are one wave per SIMD).  Whole step, M
            // env-steps/s, runs of 8 / 16 / 32 / 64: 512 envs 15.0 / 16.3 / 14.9 / 12.7; 1 024: 21.2 / 26.7 / 26.3 / 22.1; 2 048: 27.5 / 35.3 / 37.1 /
            // 36.6; 4 096: - / 45.0 / 49.8 / 47.2; 8 192: - / 52.6 / 60.1 / 59.2; 120 + 26 rays at 4 096 envs: 19.9 / - / 35.5 / 33.2
            l.run = c->run ? (c->run > 64u ? 64u : c->run) : ((uint64_t)E * c->R8 >= (1ull << 20) ? 64u : ((uint64_t)E * c->R8 >= (1ull << 17) ? 32u : 16u));
            HIP_TRY(c, launch_raycast_lane(l, s));
            return ROVER_OK;
        }
        // The rocks part too?  On a regular rocks mesh no: its rays are few per bin and a tenth of them lie off every cone (the horizontal body
        // rays, which test every pair of their cell both ways) — the staged kernel reads a cell's whole 6.6 KB of (A) and (B) rows for one such
        // ray where the culled kernel reads 800 bytes of ids and gathers: a tie at 65 536 envs (363-372 us in one launch against 226-233 +
        // 137-140).  On an irregular rocks mesh — most cells without a usable far bound — yes (465 us against 327 + 270 with the first version).
        if (lane_rocks_too(c)) {
            HIP_TRY(c, launch_raycast_lane(l, s));
        } else {
            // the terrain rays (the first E x P of the sorted list: terrain bins sort first) through the staged kernel, the rock rays — few per
            // bin, a tenth of them off every cone (the horizontal body rays) — through the culled one, which reads 800 bytes of ids per bin
            // where the staged kernel reads the 3.5 KB of a cell's whole row for one such ray
            CullArgs a = cull_args(c, n_valid);
            l.n_sorted = l.n_terrain < n_valid ? l.n_terrain : n_valid;
            a.sorted += l.n_sorted; a.n_sorted -= l.n_sorted; a.n_terrain = 0;
            a.stats += lane_waves(l.n_sorted, l.run);
            // the two launches touch disjoint rays, results and counters and could run side by side (fork / join by events); measured, that is
            // no faster than one after the other (lane_side_stream)
            const bool beside = c->lane_side_stream && c->side && a.n_sorted;
            if (beside) {
                HIP_TRY(c, hipEventRecord(c->ev_fork, s));
                HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
                HIP_TRY(c, launch_raycast_culled(a, c->side));
                HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
            }
            HIP_TRY(c, launch_raycast_lane(l, s));
            if (beside) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
            else if (a.n_sorted) HIP_TRY(c, launch_raycast_culled(a, s));
        }
    } else if (variant == 3)
        HIP_TRY(c, launch_raycast_culled(cull_args(c, n_valid), s));
    else if (variant == 2)
        HIP_TRY(c, launch_raycast_binned(c->d_rays, c->d_sorted, n_valid, c->map[0].table, c->map[1].table,
                                         (uint32_t)c->map[0].K8, (uint32_t)c->map[1].K8, effective_run(c), c->precision == 2, c->early_out, c->d_dist_out, s));
    else
        HIP_TRY(c, launch_raycast(c->d_rays, E * c->R8, c->map[0].table, c->map[1].table, (uint32_t)c->map[0].K8,
                                  (uint32_t)c->map[1].K8, c->d_dist_out, s));
    return ROVER_OK;
}

// The ray pipeline of a step: env records + ray records, the bucket sort by (map, cell), the ray cast -> d_dist_out [E][R8].
// euler_in != NULL (rover_get_depths): the poses come as euler angles, quat / joints / target may be NULL, and the ctx's euler / heading
// state of the last observation is left alone.
static int cast_rays(rover_ctx* c, const float* pos, const float* quat, const float* joints, const float* target, const float* euler_in,
                     hipStream_t s, const float* import_src = nullptr, const float* import_dir = nullptr) {
    const uint32_t E = (uint32_t)c->cfg.num_envs;
    PrepArgs p{};
    p.E = E; p.P = (uint32_t)c->P; p.R8 = c->R8;
    p.pos = pos; p.quat = quat; p.joints = joints; p.target = target; p.euler_in = euler_in;
    p.dist = c->d_dist; p.terrain = c->map[0]; p.rocks = c->map[1];
    p.rays = c->d_rays; p.euler = euler_in ? nullptr : c->d_euler; p.heading = euler_in ? nullptr : c->d_heading;
    const int variant = effective_variant(c);
    // (the queue is sized by every call that changes its size — never here: no hipMalloc inside a step / a stream capture)
    if (variant >= 3 && (!c->d_cull_queue || !c->d_cull_stats || c->cull_run != effective_run(c)))
        return fail(c, ROVER_E_STATE, "the culled ray cast's candidate queue is not allocated for the options in force");
    const uint32_t n_valid = E * (26u + (uint32_t)c->P);
    p.rocks_bin_offset = (uint32_t)((uint64_t)c->map[0].X * c->map[0].Y);
    const bool sorts = variant >= 2 && !lane_env_order(c, variant);
    if (sorts) p.bin_out = c->d_bins;
    p.precision = c->precision;
    p.cell_rcp = c->cell_rcp;
    // the sort's first pass (keys per coarse bucket and tile) inside prep_rays_kernel where a 64-env block's keys lie in one tile: the
    // table is zero between steps (allocation, then the sort's last kernel) — unless a step failed half way
    // (caller-supplied rays, rover_cast_rays: import_rays_kernel writes the records and keys, the sort counts its keys itself)
    const bool hist_fused = !import_src && sorts && bin_hist_fused(E * c->R8, c->R8, c->n_bins, c->low_bits, &p.hist_blocks_per_tile);
    if (hist_fused) {
        if (c->bkt_table_dirty) HIP_TRY(c, hipMemsetAsync(c->d_bkt_table, 0, c->bkt_table_bytes, s));
        c->bkt_table_dirty = true;
        p.hist = c->d_bkt_table; p.hist_low_bits = c->low_bits; p.hist_buckets = bucket_count(c);
    }
    if (import_src) {
        // caller-supplied directions: the culled / staged ray cast's proofs need them of unit length (what -normalize() gives)
        uint32_t* const not_unit = variant >= 3 ? c->d_block_cnt + (size_t)c->cfg.num_envs / 256 + 1 : nullptr;      // (the spare word behind the block counts)
        if (not_unit) HIP_TRY(c, hipMemsetAsync(not_unit, 0, sizeof(uint32_t), s));
        HIP_TRY(c, launch_import_rays(import_src, import_dir, E, c->R8, (uint32_t)c->P, c->map[0], c->map[1], p.rocks_bin_offset, c->precision,
                                      c->cell_rcp, c->d_rays, sorts ? c->d_bins : nullptr, s, not_unit));
        if (not_unit) {
            uint32_t bad = 0;
            HIP_TRY(c, hipMemcpyAsync(&bad, not_unit, sizeof bad, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            if (bad) {
                c->rays_valid = false;
                return fail(c, ROVER_E_INVALID, "cast_rays: %u directions are not of unit length (|d|^2 within %g of 1: pass -normalize(direction) as "
                            "rover_export_rays returns it, or use raycast_variant 1 / 2, which evaluate every triangle)", bad, c->precision == 2 ? 4.0e-3 : 1.0e-5);
            }
        }
    } else {
        HIP_TRY(c, launch_prep(p, s));
    }
    if (sorts)
        HIP_TRY(c, launch_bin_rays(c->d_bins, E * c->R8, n_valid, c->n_bins, c->low_bits, c->d_bkt_table, c->d_pairs,
                                   c->d_block_sums, c->d_sorted, hist_fused, s));
    // fused histogram: bucket_sort_kernel has cleared the table again; otherwise the table (if the sort ran) holds this step's offsets
    if (sorts) c->bkt_table_dirty = !hist_fused;        // (variant 1 does not touch the table)
    const bool timed = c->profiling && (c->prof_seen++ % c->prof_every) == 0;
    if (timed) {
        if (c->prof_pending == kProfRing && prof_drain(c)) return fail(c, ROVER_E_HIP, "profiling: event drain failed");
        HIP_TRY(c, hipEventRecord(c->ev0[c->prof_pending], s));
    }
    if (int r = run_raycast(c, variant, n_valid, s)) return r;
    if (timed) {
        HIP_TRY(c, hipEventRecord(c->ev1[c->prof_pending], s));
        ++c->prof_pending;
        ++c->prof_launches;
    }
    c->last_variant = variant;
    c->sorted_valid = sorts;
    c->rays_valid = true;
    return ROVER_OK;
}

static int do_observations(rover_ctx* c, const rover_step_in* in, const rover_step_out* out, hipStream_t s) {
    if (!in->pos || !in->quat || !in->joints || !in->target || !in->lin_hist || !in->ang_hist)
        return fail(c, ROVER_E_INVALID, "get_observations: null input pointer");
    if (!out->obs) return fail(c, ROVER_E_INVALID, "get_observations: obs is required");
    const uint32_t E = (uint32_t)c->cfg.num_envs, W = (uint32_t)(4 + c->Ns + c->Nd);
    const int64_t stride = out->obs_stride ? out->obs_stride : (int64_t)W;
    if (stride < (int64_t)W) return fail(c, ROVER_E_INVALID, "obs_stride %lld < row width %u", (long long)stride, W);
    if (int r = cast_rays(c, in->pos, in->quat, in->joints, in->target, nullptr, s)) return r;
    c->obs_valid = true;
    ObsArgs o{};
    o.E = E; o.W = W; o.R8 = c->R8; o.obs_stride = stride;
    o.pos = in->pos; o.target = in->target; o.heading = c->d_heading; o.lin_hist = in->lin_hist; o.ang_hist = in->ang_hist;
    o.dist = c->d_dist_out; o.obs_idx = c->d_obs_idx; o.obs = out->obs; o.fp16_div = c->precision == 2;
    if (c->defer_obs) { c->pending_obs = o; c->obs_pending = true; }
    else HIP_TRY(c, launch_assemble_obs(o, s));
    if (out->ray_dist || out->wheel_dist || out->body_dist || out->ray_src || out->hit_pt)
        HIP_TRY(c, launch_export_dist(c->d_dist_out, c->d_rays, E, c->R8, (uint32_t)c->P, c->precision, out->ray_dist, out->wheel_dist,
                                      out->body_dist, out->ray_src, out->hit_pt, s));
    if (out->euler) HIP_TRY(c, hipMemcpyAsync(out->euler, c->d_euler, (uint64_t)E * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out->heading_diff) HIP_TRY(c, hipMemcpyAsync(out->heading_diff, c->d_heading, (uint64_t)E * sizeof(float), hipMemcpyDeviceToDevice, s));
    return ROVER_OK;
}

static int do_metrics(rover_ctx* c, const rover_step_in* in, const rover_step_out* out, int inc, int coll, int met, int done,
                      hipStream_t s, bool count_done = false) {
    if (!in->pos || !in->target) return fail(c, ROVER_E_INVALID, "metrics/done: null input pointer");
    if ((inc || met || done) && !in->progress) return fail(c, ROVER_E_INVALID, "metrics/done: progress is required");
    if (!out->rock_collision) return fail(c, ROVER_E_INVALID, "metrics/done: rock_collision is required");
    if (met && (!in->joints || !in->lin_hist || !in->ang_hist || !out->rew))
        return fail(c, ROVER_E_INVALID, "calculate_metrics: joints, lin_hist, ang_hist and rew are required");
    if (done && (!in->euler_pre || !out->reset)) return fail(c, ROVER_E_INVALID, "is_done: euler_pre and reset are required");
    if (coll && out->stone_collision) {
        if (!c->have_stones) return fail(c, ROVER_E_STATE, "stone_collision: rover_set_stones first");
        if (!(out->stone_margin <= 1.4f)) return fail(c, ROVER_E_INVALID, "stone_margin %g exceeds the 1.4 m reach of the stone grid", (double)out->stone_margin);
    }
    MetricsArgs m{};
    m.E = (uint32_t)c->cfg.num_envs; m.R8 = c->R8;
    m.curriculum_level = c->cfg.curriculum_level; m.max_episode_length = c->cfg.max_episode_length;
    m.num_envs_global = c->cfg.num_envs_global;
    m.do_increment = inc; m.do_collision = coll; m.do_metrics = met; m.do_done = done;
    m.pos_reward = c->cfg.pos_reward; m.heading_contraint_reward = c->cfg.heading_contraint_reward;
    m.motion_contraint_reward = c->cfg.motion_contraint_reward; m.goal_angle_reward = c->cfg.goal_angle_reward;
    m.boogie_contraint_reward = c->cfg.boogie_contraint_reward;
    m.wheel_thr = c->precision == 2 ? 0.7998046875f : 0.8f;                // fp16(0.8), fp16(0.45): Python scalars compared
    m.body_thr = c->precision == 2 ? 0.449951171875f : 0.45f;              // against fp16 tensors (rover.py:667-668)
    m.pos = in->pos; m.target = in->target; m.joints = in->joints; m.lin_hist = in->lin_hist; m.ang_hist = in->ang_hist;
    m.euler_pre = in->euler_pre; m.heading = c->d_heading; m.dist = c->d_dist_out;
    m.progress = in->progress; m.rock_collision = out->rock_collision; m.rew = out->rew; m.reset = out->reset;
    m.ex_pos_reward = out->ex_pos_reward; m.ex_collision = out->ex_collision_penalty; m.ex_upright = out->ex_uprightness_penalty;
    m.ex_heading = out->ex_heading_contraint_penalty; m.ex_motion = out->ex_motion_contraint_penalty;
    m.block_cnt = count_done ? c->d_block_cnt : nullptr;
    m.stone_collision = coll ? out->stone_collision : nullptr; m.stone_margin = out->stone_margin;
    m.done_u8 = done ? out->done_u8 : nullptr;
    m.sgrid = c->sgrid; m.info7 = c->d_stones;
    m.ex_goal_angle = out->ex_goal_angle_penalty; m.ex_lin = out->ex_torque_penalty_driving; m.ex_ang = out->ex_torque_penalty_steering;
    if (c->obs_pending) {
        c->obs_pending = false;
        HIP_TRY(c, launch_obs_metrics(c->pending_obs, m, s));
    } else {
        HIP_TRY(c, launch_metrics_done(m, s));
    }
    return ROVER_OK;
}

int rover_get_observations(rover_ctx* c, const rover_step_in* in, const rover_step_out* out, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!in || !out) return fail(c, ROVER_E_INVALID, "get_observations: null struct");
    if (int r = check_ready(c)) return r;
    USE_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    if (int r = do_observations(c, in, out, s)) return r;
    if (!out->rock_collision) return ROVER_OK;
    // check_collision (rover.py:292-293) is part of get_observations: run only the collision stage
    return do_metrics(c, in, out, 0, 1, 0, 0, s);
}

int rover_calculate_metrics(rover_ctx* c, const rover_step_in* in, const rover_step_out* out, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!in || !out) return fail(c, ROVER_E_INVALID, "calculate_metrics: null struct");
    if (int r = check_ready(c)) return r;
    if (!c->obs_valid) return fail(c, ROVER_E_STATE, "calculate_metrics: call rover_get_observations first (rover.py:479 reads self.heading_diff)");
    USE_DEVICE(c);
    return do_metrics(c, in, out, 0, 0, 1, 0, (hipStream_t)stream);
}

int rover_is_done(rover_ctx* c, const rover_step_in* in, const rover_step_out* out, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!in || !out) return fail(c, ROVER_E_INVALID, "is_done: null struct");
    if (int r = check_ready(c)) return r;
    USE_DEVICE(c);
    return do_metrics(c, in, out, 0, 0, 0, 1, (hipStream_t)stream);
}

int rover_get_depths(rover_ctx* c, const float* positions, const float* rotations_euler, float* distances, float* points,
                     float* sources, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!positions || !rotations_euler) return fail(c, ROVER_E_INVALID, "get_depths: positions and rotations are required");
    if (int r = check_ready(c)) return r;
    USE_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    if (int r = cast_rays(c, positions, nullptr, nullptr, nullptr, rotations_euler, s)) return r;
    if (distances || points || sources)
        HIP_TRY(c, launch_export_dist(c->d_dist_out, c->d_rays, (uint32_t)c->cfg.num_envs, c->R8, (uint32_t)c->P, c->precision, distances,
                                      nullptr, nullptr, sources, points, s));
    return ROVER_OK;
}

int rover_get_collisions(rover_ctx* c, const float* positions, const float* rotations_euler, const float* joints, float* wheel_dist,
                         float* body_dist, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!positions || !rotations_euler) return fail(c, ROVER_E_INVALID, "get_collisions: positions and rotations are required");
    if (int r = check_ready(c)) return r;
    USE_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    if (int r = cast_rays(c, positions, nullptr, joints, nullptr, rotations_euler, s)) return r;
    if (wheel_dist || body_dist)
        HIP_TRY(c, launch_export_dist(c->d_dist_out, c->d_rays, (uint32_t)c->cfg.num_envs, c->R8, (uint32_t)c->P, c->precision, nullptr,
                                      wheel_dist, body_dist, nullptr, nullptr, s));
    return ROVER_OK;
}

int rover_export_rays(rover_ctx* c, float* src, float* dir, int32_t* cell, float* dist, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (int r = check_ready(c)) return r;
    if (!c->rays_valid) return fail(c, ROVER_E_STATE, "export_rays: no ray records yet (run a step first)");
    USE_DEVICE(c);
    HIP_TRY(c, launch_export_rays(c->d_rays, c->d_dist_out, (uint32_t)c->cfg.num_envs, c->R8, (uint32_t)c->P, src, dir, cell, dist,
                                  (hipStream_t)stream));
    return ROVER_OK;
}

int rover_cast_rays(rover_ctx* c, const float* src, const float* dir, float* dist, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!src || !dir || !dist) return fail(c, ROVER_E_INVALID, "cast_rays: src, dir and dist are required");
    if (int r = check_ready(c)) return r;
    USE_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    if (int r = cast_rays(c, nullptr, nullptr, nullptr, nullptr, nullptr, s, src, dir)) return r;
    HIP_TRY(c, launch_export_rays(c->d_rays, c->d_dist_out, (uint32_t)c->cfg.num_envs, c->R8, (uint32_t)c->P, nullptr, nullptr, nullptr, dist, s));
    return ROVER_OK;
}

int rover_compact_resets(rover_ctx* c, const int64_t* reset, int64_t* ids, int32_t* n_reset, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!reset || !ids || !n_reset) return fail(c, ROVER_E_INVALID, "compact_resets: null pointer");
    USE_DEVICE(c);
    HIP_TRY(c, launch_compact(reset, (uint32_t)c->cfg.num_envs, (int64_t)c->cfg.env_offset, c->d_block_cnt, false, ids, n_reset,
                              (hipStream_t)stream));
    return ROVER_OK;
}

int rover_step(rover_ctx* c, const rover_step_in* in, const rover_step_out* out, uint32_t flags, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!in || !out) return fail(c, ROVER_E_INVALID, "step: null struct");
    if (int r = check_ready(c)) return r;
    USE_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    if ((flags & ROVER_STEP_COMPACT) && (!out->reset_ids || !out->n_reset))
        return fail(c, ROVER_E_INVALID, "step: ROVER_STEP_COMPACT needs reset_ids and n_reset");
    c->defer_obs = true;              // the obs pass is launched by do_metrics, in one grid with the metrics pass
    c->obs_pending = false;
    const int ro = do_observations(c, in, out, s);
    c->defer_obs = false;
    if (ro) { c->obs_pending = false; return ro; }
    const bool compact = (flags & ROVER_STEP_COMPACT) != 0;
    if (int r = do_metrics(c, in, out, (flags & ROVER_STEP_INCREMENT_PROGRESS) ? 1 : 0, 1, 1, 1, s, compact)) {
        c->obs_pending = false;
        return r;
    }
    if (compact)
        HIP_TRY(c, launch_compact(out->reset, (uint32_t)c->cfg.num_envs, (int64_t)c->cfg.env_offset, c->d_block_cnt, true,
                                  out->reset_ids, out->n_reset, s));
    return ROVER_OK;
}

int rover_quat_to_euler(rover_ctx* c, const float* quat, float* euler, int32_t n, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!quat || !euler || n < 0) return fail(c, ROVER_E_INVALID, "quat_to_euler: bad arguments");
    if (n == 0) return ROVER_OK;
    USE_DEVICE(c);
    HIP_TRY(c, launch_quat_to_euler(quat, euler, (uint32_t)n, (hipStream_t)stream));
    return ROVER_OK;
}

// ---- reset path ------------------------------------------------------------------------------------
int rover_clearance(rover_ctx* c, const float* xy, int32_t n, float* out, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!c->have_stones) return fail(c, ROVER_E_STATE, "clearance: rover_set_stones first");
    if (n < 0 || (n > 0 && (!xy || !out))) return fail(c, ROVER_E_INVALID, "clearance: bad arguments");
    if (n == 0) return ROVER_OK;
    USE_DEVICE(c);
    HIP_TRY(c, launch_clearance(c->d_stones, (uint32_t)c->S, xy, (uint32_t)n, out, (hipStream_t)stream));
    return ROVER_OK;
}

int rover_shift_spawns(rover_ctx* c, float* pos3, int32_t n, int32_t max_iter, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!c->have_stones) return fail(c, ROVER_E_STATE, "shift_spawns: rover_set_stones first");
    if (n < 0 || (n > 0 && !pos3) || max_iter < 0) return fail(c, ROVER_E_INVALID, "shift_spawns: bad arguments");
    if (n == 0) return ROVER_OK;
    USE_DEVICE(c);
    HIP_TRY(c, launch_shift_spawns(c->sgrid, c->d_stones, pos3, (uint32_t)n, max_iter, (hipStream_t)stream));
    return ROVER_OK;
}

int rover_sample_height(rover_ctx* c, const float* xy, int32_t n, float* out, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!c->have_hf) return fail(c, ROVER_E_STATE, "sample_height: rover_set_heightfield first");
    if (n < 0 || (n > 0 && (!xy || !out))) return fail(c, ROVER_E_INVALID, "sample_height: bad arguments");
    if (n == 0) return ROVER_OK;
    USE_DEVICE(c);
    HIP_TRY(c, launch_sample_height(c->hf, xy, (uint32_t)n, out, (hipStream_t)stream));
    return ROVER_OK;
}

int rover_generate_goals(rover_ctx* c, const int64_t* env_ids, int32_t n, const float* initial_pos3, float* target3, float radius,
                         const float* draws, int32_t max_draws, uint64_t seed, int32_t* n_draws_used, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!c->have_stones || !c->have_hf) return fail(c, ROVER_E_STATE, "generate_goals: rover_set_stones and rover_set_heightfield first");
    if (n < 0 || n > c->cfg.num_envs || (n > 0 && (!env_ids || !initial_pos3 || !target3)) || max_draws <= 0)
        return fail(c, ROVER_E_INVALID, "generate_goals: bad arguments (n=%d, max_draws=%d)", n, max_draws);
    if (n == 0) return ROVER_OK;
    USE_DEVICE(c);
    GoalArgs g{c->d_stones, (uint32_t)c->S, c->hf, c->sgrid, env_ids, 0, (uint32_t)n, nullptr, initial_pos3, target3, radius, draws,
               max_draws, seed, nullptr, (int32_t*)c->d_goal_work, n_draws_used};
    HIP_TRY(c, launch_generate_goals(g, (uint32_t)n, (hipStream_t)stream));
    return ROVER_OK;
}

int rover_reset_envs(rover_ctx* c, const rover_reset_io* io, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!io) return fail(c, ROVER_E_INVALID, "reset_envs: null struct");
    if (!io->reset_ids || !io->initial_pos3 || !io->pos3 || !io->quat4 || !io->reset || !io->progress)
        return fail(c, ROVER_E_INVALID, "reset_envs: reset_ids, initial_pos3, pos3, quat4, reset and progress are required");
    if (!io->n_reset_dev && (io->n_reset_host < 0 || io->n_reset_host > c->cfg.num_envs))
        return fail(c, ROVER_E_INVALID, "reset_envs: n_reset_host=%d out of range", io->n_reset_host);
    if (io->draws && io->n_reset_dev) return fail(c, ROVER_E_INVALID, "reset_envs: caller-supplied draws need n_reset_host");
    // yaw_deg[i] is read for every i < n: with the count on the device n can be any value up to num_envs
    if (io->yaw_deg && io->yaw_deg_len < (io->n_reset_dev ? c->cfg.num_envs : io->n_reset_host))
        return fail(c, ROVER_E_INVALID, "reset_envs: yaw_deg holds %d entries, %d may be read", io->yaw_deg_len,
                    io->n_reset_dev ? c->cfg.num_envs : io->n_reset_host);
    if (io->target3 && (!c->have_stones || !c->have_hf))
        return fail(c, ROVER_E_STATE, "reset_envs: goal validation needs rover_set_stones and rover_set_heightfield");
    if (!io->n_reset_dev && io->n_reset_host == 0) return ROVER_OK;
    USE_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    ResetArgs a{};
    a.ids = io->reset_ids; a.id_offset = c->cfg.env_offset; a.n_host = (uint32_t)io->n_reset_host; a.n_dev = io->n_reset_dev;
    a.initial_pos3 = io->initial_pos3; a.pos3 = io->pos3; a.quat4 = io->quat4; a.joint_pos13 = io->joint_pos13;
    a.joint_vel13 = io->joint_vel13; a.base_pos3 = io->base_pos3; a.reset = io->reset; a.progress = io->progress;
    a.yaw_deg = io->yaw_deg; a.seed = io->seed; a.seed_dev = io->seed_dev;
    const uint32_t n_max = io->n_reset_dev ? (uint32_t)c->cfg.num_envs : (uint32_t)io->n_reset_host;
    HIP_TRY(c, launch_reset_envs(a, n_max, s));
    if (io->target3) {
        GoalArgs g{c->d_stones, (uint32_t)c->S, c->hf, c->sgrid, io->reset_ids, (int64_t)c->cfg.env_offset, (uint32_t)io->n_reset_host,
                   io->n_reset_dev, io->initial_pos3, io->target3, io->radius > 0.f ? io->radius : 8.0f, io->draws,
                   io->max_draws > 0 ? io->max_draws : 256, io->seed, io->seed_dev, (int32_t*)c->d_goal_work, io->n_draws_used};
        HIP_TRY(c, launch_generate_goals(g, n_max, s));
    }
    return ROVER_OK;
}

int rover_pre_physics_step(rover_ctx* c, const float* actions, const float* quat, float* lin_hist, float* ang_hist,
                           float* euler_pre, float* pos_targets13, float* vel_targets13, float* actions_nn, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!actions || !lin_hist || !ang_hist) return fail(c, ROVER_E_INVALID, "pre_physics_step: actions and both histories are required");
    if (euler_pre && !quat) return fail(c, ROVER_E_INVALID, "pre_physics_step: euler_pre needs quat");
    USE_DEVICE(c);
    PrePhysicsArgs a{(uint32_t)c->cfg.num_envs, actions, quat, lin_hist, ang_hist, euler_pre, pos_targets13, vel_targets13, actions_nn};
    HIP_TRY(c, launch_pre_physics(a, (hipStream_t)stream));
    return ROVER_OK;
}

int rover_ackermann(rover_ctx* c, const float* lin, const float* ang, int32_t n, float* steering, float* velocities, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (n < 0 || (n > 0 && (!lin || !ang || !steering || !velocities))) return fail(c, ROVER_E_INVALID, "ackermann: bad arguments");
    if (n == 0) return ROVER_OK;
    USE_DEVICE(c);
    HIP_TRY(c, launch_ackermann(lin, ang, (uint32_t)n, steering, velocities, (hipStream_t)stream));
    return ROVER_OK;
}

int rover_get_info(const rover_ctx* c, rover_info* info) {
    if (!c || !info) return ROVER_E_INVALID;
    memset(info, 0, sizeof *info);
    info->P = c->P; info->Ns = c->Ns; info->Nd = c->Nd; info->rays_per_env_padded = (int32_t)c->R8;
    for (int w = 0; w < 2; ++w) {
        info->K[w] = c->map[w].K; info->K8[w] = c->map[w].K8; info->X[w] = c->map[w].X; info->Y[w] = c->map[w].Y;
        info->table_bytes[w] = c->table_bytes[w];
    }
    // the per-step workspace: ray records, distances, sort buffers, env records, and the culled ray cast's queue + counters
    info->workspace_bytes = c->workspace_bytes + (c->d_cull_queue ? c->cull_entries * sizeof(uint2) : 0) +
                            (c->d_cull_stats ? (uint64_t)c->cull_stat_slots * sizeof(uint4) : 0);
    info->raycast_variant = (c->have_map[0] && c->have_map[1]) ? effective_variant(c) : 0;
    info->cell_index_mode = c->cell_rcp; info->ray_precision = c->precision;
    info->raycast_sorted = info->raycast_variant >= 2 && !lane_env_order(c, info->raycast_variant);
    info->raycast_rocks_staged = info->raycast_variant == 4 && (!info->raycast_sorted || lane_rocks_too(c)) ? 1 : 0;
    return ROVER_OK;
}

int rover_get_cull_info(rover_ctx* c, rover_cull_info* out) {
    if (!c || !out) return ROVER_E_INVALID;
    memset(out, 0, sizeof *out);
    for (int w = 0; w < 2; ++w) {             // (of the proof tables the precision in force uses)
        out->triangles[w] = c->cull_tris[w];
        out->always_candidate_triangles[w] = c->precision == 2 ? c->cull_always_h[w] : c->cull_always[w];
        out->cells_without_cone[w] = c->precision == 2 ? c->cull_nocone_h[w] : c->cull_nocone[w];
    }
    for (int w = 0; w < 2; ++w) out->cells_with_far_bound[w] = c->cull_farok[w];
    out->far_records_on_demand = (c->have_dist && c->have_map[0]) ? (uint64_t)cull_args(c, 0).lazy_far : 0;
    out->queue_bytes = c->d_cull_queue ? c->cull_entries * sizeof(uint2) : 0;
    out->launches_per_step = c->d_cull_queue ? c->cull_launches : 0;
    if (!c->d_cull_stats || c->last_variant < 3) return ROVER_OK;
    USE_DEVICE(c);
    HIP_TRY(c, hipDeviceSynchronize());
    std::vector<uint4> h(c->cull_stat_slots);
    HIP_TRY(c, hipMemcpy(h.data(), c->d_cull_stats, h.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    for (const uint4& v : h) {
        out->candidate_pairs += v.x; out->rays += v.y & 0xffu; out->rays_far_skipped += v.y >> 8; out->rays_both_tests += v.z & 0xffu; out->rays_not_scanned += v.z >> 8; out->bins += v.w & 0xffu;
        out->lane_items += (v.w >> 8) & 0x3ffffu; out->lane_flushes += v.w >> 26;      // (zero in the words the culled kernel's waves write)
        out->max_pairs_per_run = v.x > out->max_pairs_per_run ? v.x : out->max_pairs_per_run;
    }
    return ROVER_OK;
}

// IEEE binary16 <-> binary32 on the host (round to nearest even), for the reference-ranking coordinate tables
static float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else { int e = -1; uint32_t m = man; do { ++e; m <<= 1; } while (!(m & 0x400u)); u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13); }
    } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &u, sizeof f); return f;
}
static uint16_t float_to_half_bits(float f) {
    uint32_t u; memcpy(&u, &f, sizeof u);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                        // rounds to infinity
    if (u < 0x38800000u) {                                                         // subnormal half or zero
        if (u < 0x33000000u) return (uint16_t)sign;
        const int shift = 126 - (int)(u >> 23);                                    // 14 .. 24
        const uint32_t m = (u & 0x7fffffu) | 0x800000u;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (u - 0x38000000u) >> 13;
    const uint32_t rem = u & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}

static int build_knn_map_impl(rover_ctx* c, const float* vertices, int32_t V, const int32_t* triangles, int32_t T, int32_t X, int32_t Y,
                              float res, int32_t K, int ref, const uint16_t* cell_x_f16, const uint16_t* cell_y_f16,
                              int32_t* map_idx_out);

int rover_build_knn_map(rover_ctx* c, const float* vertices, int32_t V, const int32_t* triangles, int32_t T, int32_t X, int32_t Y,
                        float res, int32_t K, int32_t* map_idx_out) {
    return build_knn_map_impl(c, vertices, V, triangles, T, X, Y, res, K, 0, nullptr, nullptr, map_idx_out);
}

int rover_build_knn_map_ref(rover_ctx* c, const float* vertices, int32_t V, const int32_t* triangles, int32_t T, int32_t X, int32_t Y,
                            float res, int32_t K, const uint16_t* cell_x_f16, const uint16_t* cell_y_f16, int32_t* map_idx_out) {
    return build_knn_map_impl(c, vertices, V, triangles, T, X, Y, res, K, 1, cell_x_f16, cell_y_f16, map_idx_out);
}

static int build_knn_map_impl(rover_ctx* c, const float* vertices, int32_t V, const int32_t* triangles, int32_t T, int32_t X, int32_t Y,
                              float res, int32_t K, int ref, const uint16_t* cell_x_f16, const uint16_t* cell_y_f16,
                              int32_t* map_idx_out) {
    if (!c) return ROVER_E_INVALID;
    if (!vertices || !triangles || !map_idx_out || V <= 0 || T <= 0 || X <= 0 || Y <= 0 || K <= 0 || !(res > 0.0f))
        return fail(c, ROVER_E_INVALID, "build_knn_map: bad arguments");
    if (T < K) return fail(c, ROVER_E_INVALID, "build_knn_map: the mesh has %d triangles, fewer than K=%d", T, K);
    if (K > 4096) return fail(c, ROVER_E_INVALID, "build_knn_map: K=%d exceeds the builder's limit of 4096", K);
    USE_DEVICE(c);
    float *d_v = nullptr, *d_cx = nullptr, *d_cy = nullptr, *d_cell = nullptr;
    int32_t *d_t = nullptr, *d_over = nullptr;
    uint32_t *d_cur = nullptr, *d_items = nullptr, *d_bs = nullptr, *d_start = nullptr;
    auto cleanup = [&]() { dfree(d_v); dfree(d_cx); dfree(d_cy); dfree(d_t); dfree(d_over); dfree(d_cur); dfree(d_items); dfree(d_bs); dfree(d_cell); dfree(d_start); };
#define KNN_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) { cleanup(); return fail(c, ROVER_E_HIP, "build_knn_map: %s: %s", #expr, hipGetErrorString(e__)); } \
    } while (0)
    KNN_TRY(hipMalloc((void**)&d_v, (size_t)V * 3 * sizeof(float)));
    KNN_TRY(hipMalloc((void**)&d_t, (size_t)T * 3 * sizeof(int32_t)));
    KNN_TRY(hipMalloc((void**)&d_cx, (size_t)T * sizeof(float)));
    KNN_TRY(hipMalloc((void**)&d_cy, (size_t)T * sizeof(float)));
    KNN_TRY(hipMalloc((void**)&d_items, (size_t)T * sizeof(uint32_t)));
    KNN_TRY(hipMalloc((void**)&d_over, sizeof(int32_t)));
    KNN_TRY(hipMalloc((void**)&d_bs, 8192 * sizeof(uint32_t)));
    KNN_TRY(hipMemcpy(d_v, vertices, (size_t)V * 3 * sizeof(float), hipMemcpyDefault));
    KNN_TRY(hipMemcpy(d_t, triangles, (size_t)T * 3 * sizeof(int32_t), hipMemcpyDefault));
    KNN_TRY(hipMemset(d_over, 0, sizeof(int32_t)));
    KNN_TRY(launch_knn_centroids(d_v, d_t, (uint32_t)T, (uint32_t)V, ref, d_cx, d_cy, nullptr));
    if (ref) {
        // cell coordinates as fp16 values: the caller's tables (what the reference's torch.arange(0, X res, res, dtype=float16)
        // gave on the host that built the map), or fp16(float(i) * res) — ATen's CUDA arange kernel, the reference's own device
        std::vector<float> cell((size_t)X + (size_t)Y);
        std::vector<uint16_t> hx((size_t)X), hy((size_t)Y);
        if (cell_x_f16) KNN_TRY(hipMemcpy(hx.data(), cell_x_f16, hx.size() * sizeof(uint16_t), hipMemcpyDefault));
        if (cell_y_f16) KNN_TRY(hipMemcpy(hy.data(), cell_y_f16, hy.size() * sizeof(uint16_t), hipMemcpyDefault));
        for (int32_t i = 0; i < X; ++i) cell[(size_t)i] = half_bits_to_float(cell_x_f16 ? hx[(size_t)i] : float_to_half_bits((float)i * res));
        for (int32_t j = 0; j < Y; ++j) cell[(size_t)X + j] = half_bits_to_float(cell_y_f16 ? hy[(size_t)j] : float_to_half_bits((float)j * res));
        KNN_TRY(hipMalloc((void**)&d_cell, cell.size() * sizeof(float)));
        KNN_TRY(hipMemcpy(d_cell, cell.data(), cell.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // bucket grid over the centroids' bounding box; bucket edge ~ the radius that holds K/4 centroids at mean density
    std::vector<float> hx((size_t)T), hy((size_t)T);
    KNN_TRY(hipMemcpy(hx.data(), d_cx, (size_t)T * sizeof(float), hipMemcpyDeviceToHost));
    KNN_TRY(hipMemcpy(hy.data(), d_cy, (size_t)T * sizeof(float), hipMemcpyDeviceToHost));
    float x0 = hx[0], x1 = hx[0], y0 = hy[0], y1 = hy[0];
    for (int32_t t = 0; t < T; ++t) {
        if (hx[t] == hx[t]) { x0 = hx[t] < x0 ? hx[t] : x0; x1 = hx[t] > x1 ? hx[t] : x1; }
        if (hy[t] == hy[t]) { y0 = hy[t] < y0 ? hy[t] : y0; y1 = hy[t] > y1 ? hy[t] : y1; }
    }
    const double area = ((double)x1 - x0 + 1e-6) * ((double)y1 - y0 + 1e-6);
    double gsz = std::sqrt(area * (double)K / (4.0 * (double)T));
    if (gsz < (double)res) gsz = (double)res;
    // A cell's search gathers whole rings of buckets into LDS (8192 candidates at most).  On a strongly non-uniform mesh (a
    // decimated terrain: millimetre triangles on the rocks, metre-sized ones between them) a ring sized for the MEAN density can
    // hold more than that next to a dense patch: the bucket edge is halved and the search repeated (same result: the ranking
    // does not depend on the bucket size).
    hipError_t e2 = hipSuccess;
    int32_t over = 0;
    for (int attempt = 0; attempt < 6; ++attempt, gsz *= 0.5) {
        uint32_t nbx = (uint32_t)(((double)x1 - x0) / gsz) + 1, nby = (uint32_t)(((double)y1 - y0) / gsz) + 1;
        while ((uint64_t)nbx * nby > (1u << 22)) { gsz *= 2.0; nbx = (uint32_t)(((double)x1 - x0) / gsz) + 1; nby = (uint32_t)(((double)y1 - y0) / gsz) + 1; attempt = 99; }
        const float g = (float)gsz, inv_g = 1.0f / g;
        const uint32_t nb = nbx * nby;
        dfree(d_cur); dfree(d_start);
        KNN_TRY(hipMalloc((void**)&d_cur, ((size_t)nb + 1) * sizeof(uint32_t)));
        KNN_TRY(hipMemset(d_cur, 0, ((size_t)nb + 1) * sizeof(uint32_t)));
        KNN_TRY(hipMemset(d_over, 0, sizeof(int32_t)));
        KNN_TRY(launch_knn_bucket(d_cx, d_cy, (uint32_t)T, x0, y0, inv_g, nbx, nby, d_cur, d_items, 1, nullptr));
        KNN_TRY(launch_scan_exclusive(d_cur, nb + 1, d_bs, nullptr));
        e2 = hipMalloc((void**)&d_start, ((size_t)nb + 1) * sizeof(uint32_t));
        if (e2 == hipSuccess) e2 = hipMemcpy(d_start, d_cur, ((size_t)nb + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice);
        if (e2 == hipSuccess) e2 = launch_knn_bucket(d_cx, d_cy, (uint32_t)T, x0, y0, inv_g, nbx, nby, d_cur, d_items, 0, nullptr);
        if (e2 == hipSuccess) e2 = launch_knn_select(d_cx, d_cy, d_start, d_items, x0, y0, g, nbx, nby, (uint32_t)X, (uint32_t)Y, res,
                                                     (uint32_t)K, d_cell, d_cell ? d_cell + X : nullptr, map_idx_out, d_over, nullptr);
        if (e2 == hipSuccess) e2 = hipDeviceSynchronize();
        over = 0;
        if (e2 == hipSuccess) e2 = hipMemcpy(&over, d_over, sizeof over, hipMemcpyDeviceToHost);
        if (e2 != hipSuccess || !over) break;
    }
    cleanup();
#undef KNN_TRY
    if (e2 != hipSuccess) return fail(c, ROVER_E_HIP, "build_knn_map: %s", hipGetErrorString(e2));
    if (over) return fail(c, ROVER_E_INVALID, "build_knn_map: a search ring held more than 8192 candidate triangles (mesh too dense for K=%d)", K);
    return ROVER_OK;
}

int rover_linear_forward(rover_ctx* c, const float* x, int64_t x_stride, int32_t M, int32_t K, const float* weight, const float* bias,
                         int32_t N, int32_t activation, float* y, int64_t y_stride, void* stream) {
    if (!c) return ROVER_E_INVALID;
    // K = 0: a layer over an empty obs slice (model.py builds Encoder(0, ...) when a heightmap part is absent) = act(bias)
    if ((K > 0 && (!x || !weight)) || !y || M < 0 || K < 0 || N <= 0 || N > 256 || x_stride < K || y_stride < N || activation < 0 || activation > 4)
        return fail(c, ROVER_E_INVALID, "linear_forward: bad arguments (M=%d K=%d N=%d act=%d)", M, K, N, activation);
    if (M == 0) return ROVER_OK;
    USE_DEVICE(c);
    LinearArgs a{x, x_stride, weight, bias, y, y_stride, M, K, N, activation};
    HIP_TRY(c, launch_linear_act(a, (hipStream_t)stream));
    return ROVER_OK;
}

// validated ChainArgs of one chain (0 = ok, else the error is recorded)
static int chain_args_of(rover_ctx* c, const float* x, int64_t x_stride, int32_t M, int32_t K0, int32_t n_layers, const float* const* weights,
                         const float* const* biases, const int32_t* widths, const int32_t* activations, float* y, int64_t y_stride, ChainArgs* out) {
    if (!x || !y || !weights || !biases || !widths || !activations || M < 0 || K0 <= 0 || x_stride < K0 || (n_layers != 2 && n_layers != 4))
        return fail(c, ROVER_E_INVALID, "mlp_chain_forward: bad arguments (M=%d K0=%d layers=%d)", M, K0, n_layers);
    ChainArgs a{};
    a.x = x; a.x_stride = x_stride; a.M = M; a.K0 = K0; a.n_layers = n_layers; a.y = y; a.y_stride = y_stride;
    for (int i = 0; i < n_layers; ++i) {
        if (!weights[i] || widths[i] <= 0 || widths[i] > 256 || activations[i] < 0 || activations[i] > 4)
            return fail(c, ROVER_E_INVALID, "mlp_chain_forward: layer %d: width %d activation %d", i, widths[i], activations[i]);
        a.w[i] = weights[i]; a.b[i] = biases[i]; a.n[i] = widths[i]; a.act[i] = activations[i];
    }
    if (y_stride < a.n[n_layers - 1]) return fail(c, ROVER_E_INVALID, "mlp_chain_forward: y_stride %lld < width %d", (long long)y_stride, a.n[n_layers - 1]);
    *out = a;
    return ROVER_OK;
}

// the split-k scratch buffer, grown when a larger batch comes (not inside a stream capture — size the first call before capturing)
static int mlp_scratch_reserve(rover_ctx* c, size_t need, hipStream_t s) {
    if (need <= c->mlp_scratch_floats) return ROVER_OK;
    HIP_TRY(c, hipStreamSynchronize(s));          // kernels still reading the old buffer
    dfree(c->d_mlp_scratch);
    c->mlp_scratch_floats = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_mlp_scratch, need * sizeof(float)));
    c->mlp_scratch_floats = need;
    return ROVER_OK;
}

static int chain_run(rover_ctx* c, const ChainArgs& a, hipStream_t s) {
    if (a.M == 0) return ROVER_OK;
    if (chain_wants_splitk(a)) {          // small batches: first layer split along k through a scratch buffer
        if (int r = mlp_scratch_reserve(c, chain_splitk_scratch_floats(a.M, a.K0, a.n[0]), s)) return r;
        HIP_TRY(c, launch_chain_splitk(a, c->d_mlp_scratch, s));
        return ROVER_OK;
    }
    hipError_t e = launch_chain(a, s);
    if (e == hipErrorInvalidValue) return fail(c, ROVER_E_INVALID, "mlp_chain_forward: net outside the built tile shapes (<= 96 -> <= 64, or <= 256 -> <= 160 -> <= 128 -> <= 16 with hidden activations none / LeakyReLU / ReLU)");
    HIP_TRY(c, e);
    return ROVER_OK;
}

int rover_mlp_chain_forward(rover_ctx* c, const float* x, int64_t x_stride, int32_t M, int32_t K0, int32_t n_layers,
                            const float* const* weights, const float* const* biases, const int32_t* widths, const int32_t* activations,
                            float* y, int64_t y_stride, void* stream) {
    if (!c) return ROVER_E_INVALID;
    ChainArgs a{};
    if (int r = chain_args_of(c, x, x_stride, M, K0, n_layers, weights, biases, widths, activations, y, y_stride, &a)) return r;
    USE_DEVICE(c);
    return chain_run(c, a, (hipStream_t)stream);
}

int rover_mlp_chain_pair_forward(rover_ctx* c, int32_t M, const rover_chain_desc* da, const rover_chain_desc* db, const float* copy_src,
                                 int64_t copy_src_stride, float* copy_dst, int64_t copy_dst_stride, int32_t copy_cols, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (!da || !db || copy_cols < 0 || (copy_cols > 0 && (!copy_src || !copy_dst || copy_src_stride < copy_cols || copy_dst_stride < copy_cols)))
        return fail(c, ROVER_E_INVALID, "mlp_chain_pair_forward: bad arguments (copy_cols=%d)", copy_cols);
    ChainArgs a{}, b{};
    if (int r = chain_args_of(c, da->x, da->x_stride, M, da->K0, da->n_layers, da->weights, da->biases, da->widths, da->activations, da->y, da->y_stride, &a)) return r;
    if (int r = chain_args_of(c, db->x, db->x_stride, M, db->K0, db->n_layers, db->weights, db->biases, db->widths, db->activations, db->y, db->y_stride, &b)) return r;
    if (M == 0) return ROVER_OK;
    USE_DEVICE(c);
    hipStream_t s = (hipStream_t)stream;
    if (chain_pair_fits(a, b)) {
        const size_t fa = chain_splitk_scratch_floats(a.M, a.K0, a.n[0]), fb = chain_splitk_scratch_floats(b.M, b.K0, b.n[0]);
        if (int r = mlp_scratch_reserve(c, fa + fb, s)) return r;
        HIP_TRY(c, launch_chain_splitk_pair(a, b, c->d_mlp_scratch, c->d_mlp_scratch + fa, copy_src, copy_src_stride, copy_dst, copy_dst_stride, copy_cols, s));
        return ROVER_OK;
    }
    if (copy_cols > 0)
        HIP_TRY(c, hipMemcpy2DAsync(copy_dst, (size_t)copy_dst_stride * sizeof(float), copy_src, (size_t)copy_src_stride * sizeof(float),
                                    (size_t)copy_cols * sizeof(float), (size_t)M, hipMemcpyDeviceToDevice, s));
    if (int r = chain_run(c, a, s)) return r;
    return chain_run(c, b, s);
}

int rover_set_option(rover_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return ROVER_E_INVALID;
    USE_DEVICE(c);                                 // some options (re)allocate device workspace
    if (!strcmp(name, "raycast_variant")) {
        if (value < 0 || value > 4) return fail(c, ROVER_E_INVALID, "raycast_variant must be 0 (auto), 1 (env order), 2 (binned), 3 (culled) or 4 (staged)");
        if (value == 4 && c->have_map[0] && c->have_map[1] && c->map[0].K8 <= 256 && c->map[1].K8 <= 256 && !lane_tables_ok(c))
            return fail(c, ROVER_E_STATE, "raycast_variant 4 (staged) needs its tables for the arithmetic in force: they were not built (option "
                                          "staged_tables, or they did not fit when the maps were set)");
        c->variant = (int)value;
        return alloc_cull_queue(c);
    }
    if (!strcmp(name, "staged_tables")) {
        if (value < 0 || value > 3) return fail(c, ROVER_E_INVALID, "staged_tables must be 0 (none), 1 (f32 proof), 2 (as-shipped fp16 proof) or 3 (both)");
        c->staged_tables = (int)value;       // takes effect at the next rover_set_knn_map
        return ROVER_OK;
    }
    if (!strcmp(name, "lane_env_order")) {
        if (value < -1 || value > 1) return fail(c, ROVER_E_INVALID, "lane_env_order must be -1 (auto), 0 or 1");
        c->lane_env_order = (int)value;
        return ROVER_OK;
    }
    if (!strcmp(name, "lane_rocks")) {
        if (value < -1 || value > 1) return fail(c, ROVER_E_INVALID, "lane_rocks must be -1 (auto), 0 or 1");
        c->lane_rocks = (int)value;
        return ROVER_OK;
    }
    if (!strcmp(name, "ray_precision")) {
        if (value < 0 || value > 2) return fail(c, ROVER_E_INVALID, "ray_precision must be 0 (fp32), 1 (fp16 sources) or 2 (as shipped)");
        c->precision = (int)value;
        c->rays_valid = false;
        c->obs_valid = false;           // (what an observation means changed: rover_calculate_metrics wants a fresh rover_get_observations)
        return alloc_cull_queue(c);
    }
    if (!strcmp(name, "bin_low_bits")) {
        if (value != 0 && (value < 8 || value > 12)) return fail(c, ROVER_E_INVALID, "bin_low_bits must be 0 (chosen by the library) or in [8, 12]");
        c->low_bits_opt = (uint32_t)value;
        return alloc_bins(c);
    }
    if (!strcmp(name, "raycast_early_out")) {
        if (value < 0 || value > 1) return fail(c, ROVER_E_INVALID, "raycast_early_out must be 0 or 1");
        c->early_out = (uint32_t)value;
        return ROVER_OK;
    }
    if (!strcmp(name, "cell_index_mode")) {
        if (value < 0 || value > 1) return fail(c, ROVER_E_INVALID, "cell_index_mode must be 0 (cpu_div) or 1 (cuda_rcp)");
        c->cell_rcp = (int32_t)value;
        c->hf.rcp = c->cell_rcp;
        c->rays_valid = false;
        c->obs_valid = false;
        return ROVER_OK;
    }
    if (!strcmp(name, "cull_queue_mb")) {
        if (value < 1 || value > (1 << 20)) return fail(c, ROVER_E_INVALID, "cull_queue_mb must be in [1, 1048576]");
        c->cull_budget = (uint64_t)value << 20;
        return alloc_cull_queue(c);
    }
    if (!strcmp(name, "raycast_run")) {
        if (value < 0 || value > 4096) return fail(c, ROVER_E_INVALID, "raycast_run must be 0 (auto) or in [1, 4096]");
        c->run = (uint32_t)value;
        return alloc_cull_queue(c);
    }
    return fail(c, ROVER_E_INVALID, "unknown option '%s'", name);
}

int rover_set_profiling(rover_ctx* c, int32_t enable) {
    if (!c) return ROVER_E_INVALID;
    USE_DEVICE(c);
    if (enable && c->ev0.empty()) {
        // created into locals and swapped in only when all 2 x 256 exist: a partial failure leaves the ctx without events
        std::vector<hipEvent_t> e0, e1;
        hipError_t e = hipSuccess;
        for (int i = 0; i < kProfRing && e == hipSuccess; ++i) {
            hipEvent_t a = nullptr, b = nullptr;
            e = hipEventCreate(&a);
            if (e == hipSuccess) { e0.push_back(a); e = hipEventCreate(&b); }
            if (e == hipSuccess) e1.push_back(b);
        }
        if (e != hipSuccess) {
            for (auto& x : e0) (void)hipEventDestroy(x);
            for (auto& x : e1) (void)hipEventDestroy(x);
            return fail(c, ROVER_E_HIP, "set_profiling: hipEventCreate: %s", hipGetErrorString(e));
        }
        c->ev0.swap(e0); c->ev1.swap(e1);
    }
    if (c->prof_pending) (void)prof_drain(c);
    c->profiling = enable != 0;
    if (enable) { c->prof_ms = 0.0; c->prof_launches = 0; c->prof_seen = 0; c->prof_every = enable > 1 ? enable : 1; }
    return ROVER_OK;
}

int rover_get_profile(rover_ctx* c, rover_profile* out) {
    if (!c || !out) return ROVER_E_INVALID;
    USE_DEVICE(c);
    if (prof_drain(c)) return fail(c, ROVER_E_HIP, "get_profile: event drain failed");
    out->raycast_ms = c->prof_ms;
    out->launches = c->prof_launches;
    out->pairs_per_launch = (uint64_t)c->cfg.num_envs * ((uint64_t)c->P * (uint64_t)c->map[0].K + 26ull * (uint64_t)c->map[1].K);
    return ROVER_OK;
}

int rover_replay_raycast(rover_ctx* c, void* stream) {
    if (!c) return ROVER_E_INVALID;
    if (int r = check_ready(c)) return r;
    if (!c->rays_valid) return fail(c, ROVER_E_STATE, "replay_raycast: no ray records yet (run a step first)");
    USE_DEVICE(c);
    int v = effective_variant(c);
    if (v >= 2 && !c->sorted_valid && !lane_env_order(c, v)) v = 1;       // no sorted list from the last step: only an env-order kernel can replay
    const uint32_t n_valid = (uint32_t)c->cfg.num_envs * (26u + (uint32_t)c->P);
    hipStream_t s = (hipStream_t)stream;
    if (v >= 3 && (!c->d_cull_queue || !c->d_cull_stats || c->cull_run != effective_run(c)))
        return fail(c, ROVER_E_STATE, "the culled ray cast's candidate queue is not allocated for the options in force");
    if (int r = run_raycast(c, v, n_valid, s)) return r;
    return ROVER_OK;
}

}  // extern "C"
