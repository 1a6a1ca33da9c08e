// rover_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the rover env.step() hot path.
//
// Reference (pure PyTorch; paths relative to omniisaacgymenvs/):
//   tasks/rover.py:272-336 get_observations, :460-531 calculate_metrics, :610-647 is_done, :663-668 check_collision
//   tasks/utils/camera/camera.py:60-145,165-212,233-264     (terrain ray cast)
//   tasks/utils/camera/ray_casting.py:3-66                   (ray/triangle)
//   tasks/utils/rock_detection/rock_detect.py:52-149,160-371 (wheel/body rays)
//   tasks/utils/math/tensor_quat_to_euler.py:6-31
//
// Built with -ffp-contract=off: every f32 +,-,*,/ and sqrt is one IEEE rounding, in the reference's own
// evaluation order, so that given identical rays the ray kernel agrees bit for bit with the CPU oracle and
// differs from the reference's fp32 mode only through sin/cos/atan2/asin ulps.
//
// Kernels (one fused step = the launches marked * — seven with the default 37 + 26 rays, where prep_rays_kernel counts the
// sort's coarse buckets itself and bucket_hist_kernel is not launched; the ray cast of full batches is rover_cull.hip's):
//   repack_knn_kernel        init: (map_idx, tris, verts) -> per-cell contiguous fp16 block [cell][9][K8], near/far halves per lane
// * prep_rays_kernel         64 envs x a range of ray slots per block: quat -> euler, heading, the pose's and the wheels' sin/cos and the
//                            direction per ray kind once per block, then per slot origin, cell id, bin key [+ the sort's first pass] (A2, A4, A6)
// * bucket_hist / rowscan / scatter / sort   bucket sort of the ray slots by (map, cell), LDS atomics only (3 - 4 launches)
// * cull_scan_kernel         (rover_cull.hip) the culled ray cast: 1 wave / run of sorted rays                      (A4, A5) <- roofline kernel
//   raycast_binned_kernel    variant 2: 1 wave / run of sorted rays, 4 triangles per lane in registers, every triangle evaluated,
//                            conservative early out (round 1's roofline kernel; the bit-for-bit reference of the culled one)
//   raycast_binned_h_kernel  the same in the reference's as-shipped fp16 arithmetic (ray_precision 2)
//   raycast_kernel           variant 1: env order, half-wave per ray, streams the cell blocks (small batches, K8 > 256)
// * obs_metrics_kernel       assemble_obs (1 thread / 4 obs elements, coalesced rows) + metrics_done (1 thread / env: collision mask,
//                            stone mask, reward, extras, done, done count) in one launch                            (A1, A7, A8, A10)
//   assemble_obs_kernel / metrics_done_kernel   the same two passes on their own (the reference's method split)
// * compact_write_kernel     ballot/popc ordered stream compaction of the reset ids                                 (A12)
//   clearance / shift_spawns / sample_height / goals_draw / goals_env0 / reset_envs kernels                          (A9, f-2)
//   pre_physics / ackermann / quat_to_euler kernels                                                                  (f-1)
//   knn_centroid / knn_bucket / knn_select kernels                                                                   (f-3)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rover_internal.h"
#include "rover_raymath.h"

namespace rover {

// ---------------------------------------------------------------------------------------------------
// shared device maths (same operation order as oracle/rover_oracle.c, which cites the reference lines)
// ---------------------------------------------------------------------------------------------------

// tensor_quat_to_euler.py:17-29, one angle at a time (prep_rays_kernel gives each to a wave of its own)
__device__ __forceinline__ float quat_roll(const float* __restrict__ q) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float sinr = 2.0f * (w * x + y * z);
    float cosr = 1.0f - (2.0f * (x * x + y * y));
    return atan2f(sinr, cosr);
}
__device__ __forceinline__ float quat_pitch(const float* __restrict__ q) {
    const float half_pi = 3.1415927410125732f / 2.0f;
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float sinp = 2.0f * (w * y - z * x);
    float t = sinp - 1.0f;
    return (t >= 0.0f) ? copysignf(half_pi, sinp) : asinf(sinp);
}
__device__ __forceinline__ float quat_yaw(const float* __restrict__ q) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float siny = 2.0f * (w * z + x * y);
    float cosy = 1.0f - (2.0f * (y * y + z * z));
    return atan2f(siny, cosy);
}
__device__ __forceinline__ void quat_to_euler(const float* __restrict__ q, float& roll, float& pitch, float& yaw) {
    roll = quat_roll(q); pitch = quat_pitch(q); yaw = quat_yaw(q);
}

struct Trig6 { float sx, cx, sy, cy, sz, cz; };

__device__ __forceinline__ Trig6 euler_trig(float roll, float pitch, float yaw) {
    Trig6 t;
    t.sx = sinf(-roll);  t.cx = cosf(-roll);
    t.sy = sinf(-pitch); t.cy = cosf(-pitch);
    t.sz = sinf(-yaw);   t.cz = cosf(-yaw);
    return t;
}

// rock_detect.py:305-307 / :356-358 — f32 body transform
__device__ __forceinline__ void body_xf(float x, float y, float z, const Trig6& t, float px, float py, float pz,
                                        float& ox, float& oy, float& oz) {
    float A = y * t.cx + z * t.sx;
    float C = z * t.cx - y * t.sx;
    float B = x * t.cy - t.sy * C;
    ox = px + t.sz * A + t.cz * B;
    oy = py + t.cz * A - t.sz * B;
    oz = pz + x * t.sy + t.cy * C;
}

// rock_detect.py:256-258,275-277 then body_xf: wheel-local point -> world (translations zeroed for directions)
__device__ __forceinline__ void wheel_chain(float x, float y, float z, const float* t0, const float* t1,
                                            float sst, float cst, float ssx, float csx, float ssy, float csy,
                                            const Trig6& t, float px, float py, float pz,
                                            float& ox, float& oy, float& oz) {
    float x1 = t0[0] + x * cst + y * sst;
    float y1 = t0[1] + y * cst - x * sst;
    float z1 = t0[2] + z;
    float c1 = z1 * csx - y1 * ssx;
    float x2 = t1[0] + x1 * csy - ssy * c1;
    float y2 = t1[1] + y1 * csx + z1 * ssx;
    float z2 = t1[2] + x1 * ssy + csy * c1;
    body_xf(x2, y2, z2, t, px, py, pz, ox, oy, oz);
}

// camera.py:241-253: clamp bound is the dim-0 size for both axes; torch.round is half-to-even.
// `/ horizontal_scale` (a Python float): ATen's CPU kernel divides (rcp = 0, what the golden vectors pin); ATen's CUDA
// kernel multiplies by 1 / scale rounded to f32 (rcp = 1, option cell_index_mode) — they differ next to .5 ties only.
__device__ __forceinline__ uint32_t cell_coord(float v, float shift, float cell, float inv_cell, int32_t rcp, int32_t dim0) {
    float s = rcp ? (v - shift) * inv_cell : (v - shift) / cell;
    float hi = (float)(dim0 - 1);
    s = (s < 0.0f) ? 0.0f : s;
    s = (s > hi) ? hi : s;
    s = rintf(s);
    if (!(s == s)) return 0u;          // NaN pose: the reference raises; map to cell 0 (all tests then miss)
    return (uint32_t)s;
}

// The ray's normal-cone bound for the culled ray cast (rover_cull.hip), a 16-bit fraction rounded up, 0xffff = none; (dx, dy, dz) = the
// ray record's direction.  f32 proof: a cell whose triangles all have |N_z| / |N| above ROVER_CONE_TAU |d_z| + |d_xy| meets test (B) as a whole.
// As-shipped fp16 arithmetic (precision 2): (B)'s threshold is per triangle, the cell stores the largest angle from the vertical a ray
// may have (as a fraction of pi / 2) and the ray its angle beta (d is normalised in fp16 there: re-normalised here).
__device__ __forceinline__ uint32_t ray_cone_bound(float dx, float dy, float dz, int precision) {
    uint32_t qq = 0xffffu;
    if (precision == 2) {
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        const float beta = acosf(fminf(1.0f, fabsf(dz) * inv)) * 0.63661977f + 4.0e-5f;      // / (pi / 2), rounded up
        if (beta < 0.9999f) qq = (uint32_t)ceilf(beta * 65535.0f);                            // NaN -> 0xffff
    } else {
        const float qm = (float)ROVER_CONE_TAU * fabsf(dz) + sqrtf(dx * dx + dy * dy) + 2.0e-5f;
        if (qm < 0.9999f) qq = (uint32_t)ceilf(qm * 65535.0f);                                // NaN -> 0xffff
    }
    return qq;
}

// -(F.normalize(dir)), ray_casting.py:31
__device__ __forceinline__ void neg_normalize(float dx, float dy, float dz, float& ox, float& oy, float& oz) {
    float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    if (nrm < 1e-12f) nrm = 1e-12f;
    ox = -(dx / nrm); oy = -(dy / nrm); oz = -(dz / nrm);
}

// ---------------------------------------------------------------------------------------------------
// init: re-pack the reference's three tables into per-cell contiguous fp16 blocks [cell][9][K8]
// component q = 3*vertex + coord; padding triangles (K..K8) are NaN so every test on them fails.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) repack_knn_kernel(const int32_t* __restrict__ map_idx, const int32_t* __restrict__ tris,
                                                         const uint16_t* __restrict__ verts, uint64_t n_cells, uint32_t K,
                                                         uint32_t K8, uint32_t T, uint32_t V, uint16_t* __restrict__ table) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (cell, k8)
    if (i >= n_cells * K8) return;
    uint64_t cell = i / K8;
    uint32_t slot = (uint32_t)(i % K8);
    // Slot order inside a cell block: lane l of the binned ray cast reads slots 4l..4l+3 as two packed pairs.  Its first
    // pair holds the list entries 2l, 2l+1 (the NEARER half of the K-nearest list), its second pair the entries
    // K8/2 + 2l, K8/2 + 2l + 1 (the farther half): a ray almost never reaches the far half, so the wave can drop that
    // pair as a whole after a cheap conservative test (cast_pairs).  The min over a cell is order-free, so the other
    // consumers (raycast_kernel) do not care.
    const uint32_t lane = slot >> 2, j = slot & 3u;
    const uint32_t k = (j < 2u) ? 2u * lane + j : (K8 >> 1) + 2u * lane + (j - 2u);
    uint16_t v[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) v[q] = 0x7e00u;                         // fp16 NaN
    if (k < K) {
        uint32_t t = (uint32_t)map_idx[cell * K + k];
        if (t < T) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                uint32_t vi = (uint32_t)tris[3 * (uint64_t)t + a];
                if (vi < V) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[3 * a + c] = verts[3 * (uint64_t)vi + c];
                }
            }
        }
    }
    uint16_t* dst = table + cell * 9ull * K8 + slot;
#pragma unroll
    for (int q = 0; q < 9; ++q) dst[(uint64_t)q * K8] = v[q];
}

// ---------------------------------------------------------------------------------------------------
// prep: one thread per (env, slot).  slots 0..23 wheel rays, 24..25 body rays, 26..26+P-1 terrain rays,
// rest padding to a multiple of 8 (a ray-cast workgroup then never straddles two envs).
// ---------------------------------------------------------------------------------------------------
__constant__ float c_wheel_ray[5][3] = {{0.215 / 2, 0.130 / 2, 0.1}, {0.215 / 2, -0.130 / 2, 0.1},
                                        {-0.215 / 2, 0.130 / 2, 0.1}, {-0.215 / 2, -0.130 / 2, 0.1}, {0, 0, -1}};
__constant__ float c_wp0[6][3] = {{0.286, 0.385, -0.197}, {0.286, -0.385, -0.197}, {-0.146, 0.447, -0.197},
                                  {-0.146, -0.447, -0.197}, {-0.440, 0.385, -0.197}, {-0.440, -0.385, -0.197}};
__constant__ float c_wp1[6][3] = {{0.153, 0, 0.03}, {0.153, 0, 0.03}, {0.153, 0, 0.03}, {0.153, -0.0, 0.03},
                                  {0, 0, 0.03}, {0, 0, 0.03}};
__constant__ float c_body_pt[2][3] = {{0.340, 0, -0.01}, {-0.485, 0, -0.01}};

// per (env, slot): ray origin, unit direction, cell id, bin key.
// A workgroup takes 64 envs and a range of slot groups (8 slots each): in a group wave w works on ONE slot (8 g + w) of the 64 envs, so the
// three kinds of slot — wheel rays (f32 joint chain), body rays, heightmap rays (f64 transform) — never share a wave and the slot's constants
// are wave-uniform.  (One wave per env, lane = slot, ran all three branches in every wave: 373 VALU instructions for work of ~120, 46 us.)
// The 64 x 8 records of a group leave through LDS: per env the 8 slots are 256 contiguous bytes.
//
// What depends on the POSE only is built once per block, before the loop over its slot groups (the kernel is bound by its arithmetic —
// 20 M VALU instructions per launch at 65 536 envs x 64 slots when every block of 8 slots rebuilt it, 37 us; its stores are not what it
// waits for: 16-byte records made it 1 us faster):
//  phase 1  the tasks {roll, pitch, yaw (+ heading), one per wheel the block's slots touch} dealt over the 8 waves: an euler angle
//           (tensor_quat_to_euler.py:17-29) with its sin / cos, or the six sin / cos of a wheel's steer / suspension joints
//           (rock_detect.py:248-272); the block of slot range 0 writes euler / heading out.  (Until round 4 a kernel of its own,
//           prep_env_kernel, wrote a 240-byte record per env that this kernel read back: one launch more.)
//  phase 2  a ray's DIRECTION is its kind's — the four rays of a wheel share one (rock_detect.py:305-319), the two body rays one
//           (:356-371), all heightmap rays of a rover one (camera.py:179-181,202-212): wave k < 6 builds wheel k's, wave 6 the body
//           rays', wave 7 the heightmap rays' record — un-normalised direction, the fp16 rounding of the as-shipped modes, -normalize
//           (ray_casting.py:31), the ray's normal-cone bound for the culled ray cast.
//  phase 3  per slot group: origin, cell, bin key; the records leave through LDS.
#define PREP_SLOTS 8
__global__ void __launch_bounds__(64 * PREP_SLOTS) prep_rays_kernel(PrepArgs a, uint32_t groups_per_block) {
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t e0 = blockIdx.x * 64u, e = e0 + lane;
    const uint32_t n_groups = a.R8 / PREP_SLOTS, g0 = blockIdx.y * groups_per_block, g1 = min(g0 + groups_per_block, n_groups);
    const uint32_t s_lo = g0 * PREP_SLOTS, s_hi = g1 * PREP_SLOTS;                        // the block's slots
    const uint32_t wh_lo = min(6u, s_lo >> 2), wh_hi = min(6u, s_hi >> 2);                // the wheels they touch
    const bool live = e < a.E;
    const uint32_t n_real = 26u + a.P;
    const uint32_t ec = min(e, a.E - 1u);                                                 // loads come from clamped (always valid) addresses
    const float px = a.pos[3ull * ec], py = a.pos[3ull * ec + 1], pz = a.pos[3ull * ec + 2];
    // the distribution point of a heightmap slot: a SCALAR load (the address is wave-uniform) — as a vector load its s_waitcnt vmcnt
    // would also wait for the previous slot group's stores to be acknowledged
    auto dist_of = [&](uint32_t slot, double& x, double& y, double& z) {
        const double* p = a.dist + 3ull * (slot - 26u);
        asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x8\n\ts_load_dwordx2 %2, %3, 0x10\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(x), "=&s"(y), "=&s"(z) : "s"(p) : "memory");
    };
    __shared__ float2 s_trig[3][64];            // (sin, cos) of -roll, -pitch, -yaw of the block's 64 envs
    __shared__ float2 s_wheel[6][3][64];        // [wheel][(sin, cos) of -steer | susX | susY][env]
    __shared__ float4 s_dir[8][64];             // [kind] {dx, dy, dz, flags: map | valid << 1 | cone bound << 16}
    // optionally the first pass of the bucket sort (launch_bin_rays): the block's keys counted per coarse bucket in LDS, then added to the
    // row of the sort tile the block's 64 envs lie in (a few blocks per row; the table is all zero when the kernel starts)
    extern __shared__ uint32_t s_hist[];        // [a.hist_buckets] when a.hist
    if (a.hist) for (uint32_t i = threadIdx.x; i < a.hist_buckets; i += 64u * PREP_SLOTS) s_hist[i] = 0u;
    const uint32_t n_tasks = 3u + (wh_hi - wh_lo);
    for (uint32_t task = w; task < n_tasks; task += PREP_SLOTS) {
        if (task < 3u) {
            // one atan2 / asin and one sin / cos pair (yaw: also the heading): a third of the dependent chain each (one wave doing all
            // three angles kept the other seven waiting 1.5 us at the barrier)
            float ang;
            if (a.euler_in) {                   // (wave-uniform) the pose's euler angles as given: rover_get_depths / rover_get_collisions
                ang = a.euler_in[3ull * ec + task];
            } else {
                const float* qp = a.quat + 4ull * ec;
                const float q[4] = {qp[0], qp[1], qp[2], qp[3]};
                ang = task == 0u ? quat_roll(q) : (task == 1u ? quat_pitch(q) : quat_yaw(q));
            }
            s_trig[task][lane] = make_float2(sinf(-ang), cosf(-ang));
            if (blockIdx.y == 0u && live) {
                if (a.euler) a.euler[3ull * e + task] = ang;
                if (task == 2u && a.heading) {
                    const float* tp = a.target ? a.target : a.pos;
                    const float tx = tp[3ull * ec] - px, ty = tp[3ull * ec + 1] - py;
                    const float hx = cosf(ang), hy = sinf(ang);                           // heading_diff, rover.py:279-283
                    a.heading[e] = -atan2f(tx * hy - ty * hx, tx * hx + ty * hy);
                }
            }
        } else {
            const uint32_t wh = wh_lo + (task - 3u);
            const float* j = a.joints ? a.joints + 13ull * ec : nullptr;
            const float j0 = j ? j[0] : 0.0f, j1 = j ? j[1] : 0.0f, j2 = j ? j[2] : 0.0f, j4 = j ? j[4] : 0.0f, j6 = j ? j[6] : 0.0f,
                        j7 = j ? j[7] : 0.0f, j8 = j ? j[8] : 0.0f;
            const float steer = (wh == 0) ? j4 : (wh == 1) ? j6 : (wh == 4) ? -j7 : (wh == 5) ? j8 : 0.0f;         // :248
            const float susY = (wh == 0 || wh == 2) ? -j0 : (wh == 1 || wh == 3) ? j1 : 0.0f;                      // :263
            const float susX = (wh >= 4) ? -j2 : 0.0f;                                                           // :264
            s_wheel[wh][0][lane] = make_float2(sinf(-steer), cosf(-steer));
            s_wheel[wh][1][lane] = make_float2(sinf(susX), cosf(susX));
            s_wheel[wh][2][lane] = make_float2(sinf(susY), cosf(susY));
        }
    }
    __syncthreads();
    Trig6 t;
    {
        const float2 tr = s_trig[0][lane], tp2 = s_trig[1][lane], ty2 = s_trig[2][lane];
        t.sx = tr.x; t.cx = tr.y; t.sy = tp2.x; t.cy = tp2.y; t.sz = ty2.x; t.cz = ty2.y;
    }
    // the pose in float64 for the heightmap rays (camera.py:165-212 works in the distribution tensor's float64; widened where it is
    // used: eighteen registers held through the loop would cost the kernel its eighth wave per SIMD)
#define PREP_POSE_F64                                                                                                                    \
    const double dsx = (double)t.sx, dcx = (double)t.cx, dsy = (double)t.sy, dcy = (double)t.cy, dsz = (double)t.sz, dcz = (double)t.cz; \
    const double X = (double)px, Y = (double)py, Z = (double)pz
    if (w < 6u ? (w >= wh_lo && w < wh_hi) : (w == 6u ? (s_lo < 26u && s_hi > 24u) : s_hi > 26u)) {
        float ux, uy, uz;
        uint32_t fl;
        if (w < 6u) {                           // wheel w: rock_detect.py:305-319
            const float2 d0 = s_wheel[w][0][lane], d1 = s_wheel[w][1][lane], d2 = s_wheel[w][2][lane];
            const float zero3[3] = {0.0f, 0.0f, 0.0f};
            wheel_chain(c_wheel_ray[4][0], c_wheel_ray[4][1], c_wheel_ray[4][2], zero3, zero3, d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, t, 0.0f, 0.0f,
                        0.0f, ux, uy, uz);
            fl = 3u;
        } else if (w == 6u) {                   // the body rays: rock_detect.py:356-371
            float qx, qy, qz;
            body_xf(0.0f, 1.0f, 0.0f, t, px, py, pz, qx, qy, qz);
            ux = qx - px; uy = qy - py; uz = qz - pz;
            fl = 3u;
        } else {                                // the heightmap rays: the appended (0,0,-1) point, camera.py:179-181,202-204, float64
            PREP_POSE_F64;
            const double xn = 0.0, yn = 0.0, zn = -1.0;
            const double A = yn * dcx + zn * dsx, C = zn * dcx - yn * dsx, B = xn * dcy - dsy * C;
            ux = (float)((X + dsz * A + dcz * B) - X);
            uy = (float)((Y + dcz * A - dsz * B) - Y);
            uz = (float)((Z + xn * dsy + dcy * C) - Z);
            fl = 2u;
        }
        if (a.precision >= 1) { ux = (float)(_Float16)ux; uy = (float)(_Float16)uy; uz = (float)(_Float16)uz; }      // rover_dir.type(float16): camera.py:212, rock_detect.py:319,371
        float dx, dy, dz;
        if (a.precision == 2) {         // F.normalize on a float16 tensor: f32-accumulated norm rounded to fp16, fp16 division
            float nrm = (float)(_Float16)sqrtf(ux * ux + uy * uy + uz * uz);     // (eps 1e-12 is 0 in fp16)
            dx = -(float)(_Float16)(ux / nrm); dy = -(float)(_Float16)(uy / nrm); dz = -(float)(_Float16)(uz / nrm);
        } else {
            neg_normalize(ux, uy, uz, dx, dy, dz);
        }
        const uint32_t qq = ray_cone_bound(dx, dy, dz, a.precision);
        s_dir[w][lane] = make_float4(dx, dy, dz, __uint_as_float(fl | (qq << 16)));
    }
    // The group's 64 x 8 records and bin keys, env-major through LDS (rows of 17 float4 / 9 dwords: a wave's 64 rows spread over the banks):
    // a store instruction then writes four envs' 256-byte groups — whole lines — instead of 64 records 2 KB apart.
    __shared__ float4 s_t[64 * (2 * PREP_SLOTS + 1)];
    __shared__ uint32_t s_b[64 * (PREP_SLOTS + 1)];
    float4* const rays4 = reinterpret_cast<float4*>(a.rays);
    const Trig6& t_ = t;
    const float px_ = px, py_ = py, pz_ = pz;
    for (uint32_t g = g0; g < g1; ++g) {
        const uint32_t slot0 = g * PREP_SLOTS, slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(slot0 + w));
        RayRec rec;
        rec.sx = rec.sy = rec.sz = 0.0f; rec.cell = 0u; rec.dx = rec.dy = 0.0f; rec.dz = 1.0f; rec.flags = 0u;
        uint32_t bin = 0xffffffffu;
#ifdef ROVER_DIAG_SKIP_LO       // diagnostic builds only (wrong results, right timing): ray slots [LO, HI) are left out of the cast
        const bool real = live && slot < n_real && !(slot >= (ROVER_DIAG_SKIP_LO) && slot < (ROVER_DIAG_SKIP_HI));
#else
        const bool real = live && slot < n_real;
#endif
        uint32_t kind = 0u;
        if (real) {
            float sx, sy, sz;                  // origin
            const bool rk = slot < 26u;         // the ray's map: rocks, or the terrain
            const float m_shift_x = rk ? a.rocks.shift_x : a.terrain.shift_x, m_shift_y = rk ? a.rocks.shift_y : a.terrain.shift_y;
            const float m_cell = rk ? a.rocks.cell : a.terrain.cell, m_inv_cell = rk ? a.rocks.inv_cell : a.terrain.inv_cell;
            const int32_t m_X = rk ? a.rocks.X : a.terrain.X, m_Y = rk ? a.rocks.Y : a.terrain.Y;
            if (slot < 24u) {                   // rock_detect.py:160-319
                const uint32_t wh = slot >> 2, r = slot & 3u;
                const float2 d0 = s_wheel[wh][0][lane], d1 = s_wheel[wh][1][lane], d2 = s_wheel[wh][2][lane];
                wheel_chain(c_wheel_ray[r][0], c_wheel_ray[r][1], c_wheel_ray[r][2], c_wp0[wh], c_wp1[wh], d0.x, d0.y, d1.x, d1.y,
                            d2.x, d2.y, t, px, py, pz, sx, sy, sz);
                kind = wh;
            } else if (slot < 26u) {            // rock_detect.py:321-371
                const uint32_t r = slot - 24u;
                body_xf(c_body_pt[r][0], c_body_pt[r][1], c_body_pt[r][2], t, px, py, pz, sx, sy, sz);
                kind = 6u;
            } else {                            // camera.py:165-212, float64 like the distribution tensor
                Trig6 t = t_;                   // (opaque copies: hoisted out of the loop the widened pose is those eighteen registers)
                float px = px_, py = py_, pz = pz_;
                asm volatile("" : "+v"(t.sx), "+v"(t.cx), "+v"(t.sy), "+v"(t.cy), "+v"(t.sz), "+v"(t.cz), "+v"(px), "+v"(py), "+v"(pz));
                PREP_POSE_F64;
                double x, y, z;
                dist_of(slot, x, y, z);
                const double A = y * dcx + z * dsx, C = z * dcx - y * dsx, B = x * dcy - dsy * C;
                sx = (float)(X + dsz * A + dcz * B);
                sy = (float)(Y + dcz * A - dsz * B);
                sz = (float)(Z + x * dsy + dcy * C);
                kind = 7u;
            }
            if (a.precision >= 1) { sx = (float)(_Float16)sx; sy = (float)(_Float16)sy; sz = (float)(_Float16)sz; }      // sources.type(float16): camera.py:212
            rec.sx = sx; rec.sy = sy; rec.sz = sz;
            uint32_t ix = cell_coord(sx, m_shift_x, m_cell, m_inv_cell, a.cell_rcp, m_X);
            uint32_t iy = cell_coord(sy, m_shift_y, m_cell, m_inv_cell, a.cell_rcp, m_X);
            if (iy > (uint32_t)(m_Y - 1)) iy = (uint32_t)(m_Y - 1);        // memory safety only
            rec.cell = ix * (uint32_t)m_Y + iy;
            bin = (rk ? a.rocks_bin_offset : 0u) + rec.cell;
            if (a.hist) atomicAdd(&s_hist[bin >> a.hist_low_bits], 1u);
        }
        // the direction records are written (first group: the waves that did not build one have worked on their origins meanwhile);
        // the previous group's records have left s_t / s_b
        __syncthreads();
        if (real) {
            const float4 dr = s_dir[kind][lane];
            rec.dx = dr.x; rec.dy = dr.y; rec.dz = dr.z; rec.flags = __float_as_uint(dr.w);
        }
        s_t[lane * (2 * PREP_SLOTS + 1) + 2u * w] = make_float4(rec.sx, rec.sy, rec.sz, __uint_as_float(rec.cell));
        s_t[lane * (2 * PREP_SLOTS + 1) + 2u * w + 1u] = make_float4(rec.dx, rec.dy, rec.dz, __uint_as_float(rec.flags));
        s_b[lane * (PREP_SLOTS + 1) + w] = bin;                  // key of the bucket sort; 0xffffffff for padding slots
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < 2u; ++r) {
            const uint32_t idx = threadIdx.x + r * 64u * PREP_SLOTS, el = idx / (2u * PREP_SLOTS), q = idx % (2u * PREP_SLOTS);
            if (e0 + el < a.E) rays4[((size_t)(e0 + el) * a.R8 + slot0) * 2u + q] = s_t[el * (2 * PREP_SLOTS + 1) + q];
        }
        {       // (four groups' keys gathered in LDS and written as whole 128-byte lines: 45 KB of LDS, three blocks per CU, +2.6 us)
            const uint32_t el = threadIdx.x / PREP_SLOTS, q = threadIdx.x % PREP_SLOTS;
            if (a.bin_out && e0 + el < a.E) a.bin_out[(size_t)(e0 + el) * a.R8 + slot0 + q] = s_b[el * (PREP_SLOTS + 1) + q];
        }
    }
#undef PREP_POSE_F64
    if (a.hist) {
        __syncthreads();
        uint32_t* const row = a.hist + (size_t)(blockIdx.x / a.hist_blocks_per_tile) * a.hist_buckets;
        for (uint32_t i = threadIdx.x; i < a.hist_buckets; i += 64u * PREP_SLOTS) {
            const uint32_t v = s_hist[i];
            if (v) atomicAdd(row + i, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// ray cast — the roofline kernel.  32 lanes per ray; a lane owns 8 consecutive triangles of the cell per
// pass: nine 16-byte loads (one per vertex component) straight to VGPRs, fp16 -> f32, Möller–Trumbore with
// the reference's padded barycentric test, running min; 5-step shuffle min over the 32 lanes.
// Algorithmic traffic: 18 B per (ray, triangle).
// ---------------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float tri_test(float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                                          float ax, float ay, float az, float sx, float sy, float sz,
                                          float dx, float dy, float dz) {
    // ray_casting.py:34-59; a = v2
    float bx = v1x - ax, by = v1y - ay, bz = v1z - az;
    float cx = v0x - ax, cy = v0y - ay, cz = v0z - az;
    float gx = sx - ax, gy = sy - ay, gz = sz - az;
    float bcx = by * cz - bz * cy, bcy = bz * cx - bx * cz, bcz = bx * cy - by * cx;
    float det = bcx * dx + bcy * dy + bcz * dz;
    float gcx = gy * cz - gz * cy, gcy = gz * cx - gx * cz, gcz = gx * cy - gy * cx;
    float n = (gcx * dx + gcy * dy + gcz * dz) / det;
    float bgx = by * gz - bz * gy, bgy = bz * gx - bx * gz, bgz = bx * gy - by * gx;
    float m = (bgx * dx + bgy * dy + bgz * dz) / det;
    float k = (bcx * gx + bcy * gy + bcz * gz) / det;
    bool ok = (n >= RAY_NEG_EPS) && (m >= RAY_NEG_EPS) && (n + m <= RAY_ONE_EPS)
              && (det != RAY_NEG_EPS) && (det != RAY_ONE_EPS);       // :46,:51,:56 guards
    return ok ? k : RAY_MISS;
}

__global__ void __launch_bounds__(256) raycast_kernel(const RayRec* __restrict__ rays, uint32_t n_rays,
                                                      const _Float16* __restrict__ tab0, const _Float16* __restrict__ tab1,
                                                      uint32_t kp0, uint32_t kp1, float* __restrict__ out) {
    const uint32_t ray = (blockIdx.x * 256u + threadIdx.x) >> 5;
    const uint32_t l = threadIdx.x & 31u;
    float best = __builtin_inff();
    const bool active = ray < n_rays;
    if (active) {
        const float4* rp = reinterpret_cast<const float4*>(rays + ray);
        const float4 ra = rp[0], rb = rp[1];
        const uint32_t cell = __float_as_uint(ra.w), flags = __float_as_uint(rb.w);
        if (flags & 2u) {
            const uint32_t kp = (flags & 1u) ? kp1 : kp0;
            const _Float16* base = ((flags & 1u) ? tab1 : tab0) + (size_t)cell * 9u * kp;
            for (uint32_t c = l * 8u; c < kp; c += 256u) {
                half8 v[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = *reinterpret_cast<const half8*>(base + (size_t)q * kp + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float r = tri_test((float)v[0][j], (float)v[1][j], (float)v[2][j], (float)v[3][j], (float)v[4][j],
                                       (float)v[5][j], (float)v[6][j], (float)v[7][j], (float)v[8][j],
                                       ra.x, ra.y, ra.z, rb.x, rb.y, rb.z);
                    best = (r < best) ? r : best;
                }
            }
        }
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        float o = __shfl_xor(best, off, 32);
        best = (o < best) ? o : best;
    }
    if (active && l == 0u) out[ray] = best;
}

// ---------------------------------------------------------------------------------------------------
// obs assembly: one thread per obs element -> coalesced row writes (rover.py:320-325)
// ---------------------------------------------------------------------------------------------------
#define OBS_ILP 4
__device__ __forceinline__ void assemble_obs_block(const ObsArgs& a, uint32_t bid, uint32_t n_blocks) {
    // One thread = OBS_ILP obs elements, a grid apart.  All loads come first and from addresses that are valid for every lane: a
    // wave holds the four proprioceptive columns of one or two envs next to heightmap columns, and with the loads inside the
    // five branches it waited for memory once per branch (six round trips in a row, 3.9 us per wave: 21 us for 21 MB).  Now the
    // chain is index -> distance with everything else beside it, four elements in flight per lane.
    const uint64_t total = (uint64_t)a.E * a.W, stride = (uint64_t)n_blocks * 256u;
    const uint64_t i0 = (uint64_t)bid * 256u + threadIdx.x;
    const bool small = total <= 0xffffffffull;      // (uniform) the usual case: no 64-bit division chain in front of the loads
    uint32_t e[OBS_ILP], col[OBS_ILP];
    int32_t p[OBS_ILP];
    bool on[OBS_ILP];
#pragma unroll
    for (int k = 0; k < OBS_ILP; ++k) {
        const uint64_t i = i0 + k * stride;
        on[k] = i < total;
        const uint64_t ic = on[k] ? i : 0;
        if (small) { e[k] = a.w_div.div((uint32_t)ic); col[k] = (uint32_t)ic - e[k] * a.W; }
        else { e[k] = (uint32_t)(ic / a.W); col[k] = (uint32_t)(ic % a.W); }
        p[k] = a.obs_idx[col[k] >= 4u ? col[k] - 4u : 0u];       // sparse then dense, heightmap_distribution.py:126-133
    }
    float sv[OBS_ILP], tx[OBS_ILP], ty[OBS_ILP], dv[OBS_ILP];
#pragma unroll
    for (int k = 0; k < OBS_ILP; ++k) {
        const uint64_t e3 = 3ull * e[k];
        const float* one = col[k] == 1u ? a.heading + e[k] : (col[k] == 2u ? a.lin_hist + e3 : a.ang_hist + e3);
        sv[k] = *one;
        tx[k] = a.target[e3] - a.pos[e3]; ty[k] = a.target[e3 + 1] - a.pos[e3 + 1];
        dv[k] = a.dist[(uint64_t)e[k] * a.R8 + 26u + (uint32_t)p[k]];
    }
#pragma unroll
    for (int k = 0; k < OBS_ILP; ++k) {
        float v;
        if (col[k] >= 4u) {
            v = dv[k] / 2.0f;
            if (a.fp16_div) v = (float)(_Float16)v;            // as shipped: `sparse / 2` is an fp16 division (rover.py:324-325)
        } else if (col[k] == 0u) {
            v = sqrtf(tx[k] * tx[k] + ty[k] * ty[k]) / 9.0f;   // :320
        } else if (col[k] == 1u) {
            v = sv[k] / 3.14159265358979323846f;               // :321
        } else {
            v = sv[k];                                         // :322 lin_hist, :323 ang_hist
        }
        if (on[k]) a.obs[(uint64_t)e[k] * a.obs_stride + col[k]] = v;
    }
}
__global__ void __launch_bounds__(256) assemble_obs_kernel(ObsArgs a) { assemble_obs_block(a, blockIdx.x, gridDim.x); }

// optional intermediates: the per-ray distances by kind, and the other two return values of Camera.get_depths (camera.py:145) —
// the ray origins (camera.py:212) and the "intersection points" sources - d * k (ray_casting.py:63: d = -normalize(direction), the
// ray record's direction; k = the ray's distance, 11.0 for a miss).  As shipped (precision 2) the product and the difference are
// fp16 operations on fp16 values: rounded once each (the f32 product of two fp16 values is exact, and rounding an f32 difference of
// fp16 values to fp16 equals the fp16 difference: 24 >= 2 * 11 + 2 bits).
__global__ void __launch_bounds__(256) export_dist_kernel(const float* __restrict__ dist, const RayRec* __restrict__ rays, uint32_t E,
                                                          uint32_t R8, uint32_t P, int precision, float* __restrict__ ray_dist,
                                                          float* __restrict__ wheel, float* __restrict__ body,
                                                          float* __restrict__ ray_src, float* __restrict__ hit_pt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 26u + P;
    if (i >= (uint64_t)E * n) return;
    uint32_t e = (uint32_t)(i / n), s = (uint32_t)(i % n);
    float v = dist[(uint64_t)e * R8 + s];
    if (s < 24u) { if (wheel) wheel[24ull * e + s] = v; }
    else if (s < 26u) { if (body) body[2ull * e + (s - 24u)] = v; }
    else {
        const uint64_t o = (uint64_t)e * P + (s - 26u);
        if (ray_dist) ray_dist[o] = v;
        if (ray_src || hit_pt) {
            const RayRec r = rays[(uint64_t)e * R8 + s];
            if (ray_src) { ray_src[3 * o] = r.sx; ray_src[3 * o + 1] = r.sy; ray_src[3 * o + 2] = r.sz; }
            if (hit_pt) {
                float tx = r.dx * v, ty = r.dy * v, tz = r.dz * v;
                if (precision == 2) { tx = (float)(_Float16)tx; ty = (float)(_Float16)ty; tz = (float)(_Float16)tz; }
                float hx = r.sx - tx, hy = r.sy - ty, hz = r.sz - tz;
                if (precision == 2) { hx = (float)(_Float16)hx; hy = (float)(_Float16)hy; hz = (float)(_Float16)hz; }
                hit_pt[3 * o] = hx; hit_pt[3 * o + 1] = hy; hit_pt[3 * o + 2] = hz;
            }
        }
    }
}

// The ray records of the last cast in slot order (24 wheel, 2 body, P heightmap rays per env): origin, the record's direction
// (-normalize(direction), ray_casting.py:31), cell id, distance — what a test feeds to the oracle's per-ray arithmetic.
__global__ void __launch_bounds__(256) export_rays_kernel(const RayRec* __restrict__ rays, const float* __restrict__ dist, uint32_t E, uint32_t R8,
                                                          uint32_t n_real, float* __restrict__ src, float* __restrict__ dir,
                                                          int32_t* __restrict__ cell, float* __restrict__ out_dist) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)E * n_real) return;
    const uint32_t e = (uint32_t)(i / n_real), sl = (uint32_t)(i % n_real);
    const RayRec r = rays[(uint64_t)e * R8 + sl];
    if (src) { src[3 * i] = r.sx; src[3 * i + 1] = r.sy; src[3 * i + 2] = r.sz; }
    if (dir) { dir[3 * i] = r.dx; dir[3 * i + 1] = r.dy; dir[3 * i + 2] = r.dz; }
    if (cell) cell[i] = (int32_t)r.cell;
    if (out_dist) out_dist[i] = dist[(uint64_t)e * R8 + sl];
}

// Caller-supplied rays into the step's workspace (rover_cast_rays): origin and record direction as given, cell id, map flag, cone bound
// and bin key as prep_rays_kernel derives them from an origin / a direction (camera.py:233-264 for the cell).
__global__ void __launch_bounds__(256) import_rays_kernel(const float* __restrict__ src, const float* __restrict__ dir, uint32_t E, uint32_t R8,
                                                          uint32_t n_real, KnnDev terrain, KnnDev rocks, uint32_t rocks_bin_offset,
                                                          int precision, int cell_rcp, RayRec* __restrict__ rays, uint32_t* __restrict__ bin_out,
                                                          uint32_t* __restrict__ not_unit) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)E * R8) return;
    const uint32_t e = (uint32_t)(i / R8), sl = (uint32_t)(i % R8);
    RayRec rec;
    rec.sx = rec.sy = rec.sz = 0.0f; rec.cell = 0u; rec.dx = rec.dy = 0.0f; rec.dz = 1.0f; rec.flags = 0u;
    uint32_t bin = 0xffffffffu;
    if (sl < n_real) {
        const uint64_t o = 3ull * ((uint64_t)e * n_real + sl);
        const bool rk = sl < 26u;
        const KnnDev& m = rk ? rocks : terrain;
        rec.sx = src[o]; rec.sy = src[o + 1]; rec.sz = src[o + 2];
        rec.dx = dir[o]; rec.dy = dir[o + 1]; rec.dz = dir[o + 2];
        // the rejection proofs of the culled / staged ray cast take |d|^2 <= 1.00001 (f32; as shipped, normalised in fp16: within 4e-3 of 1),
        // what -normalize() gives: a finite direction of another length is counted, rover_cast_rays refuses the call (NaN compares false
        // in every test: such a ray keeps all its candidates by itself)
        if (not_unit) {
            const float dd = rec.dx * rec.dx + rec.dy * rec.dy + rec.dz * rec.dz;
            if (fabsf(dd - 1.0f) > (precision == 2 ? 4.0e-3f : 1.0e-5f)) atomicAdd(not_unit, 1u);
        }
        const uint32_t ix = cell_coord(rec.sx, m.shift_x, m.cell, m.inv_cell, cell_rcp, m.X);
        uint32_t iy = cell_coord(rec.sy, m.shift_y, m.cell, m.inv_cell, cell_rcp, m.X);
        if (iy > (uint32_t)(m.Y - 1)) iy = (uint32_t)(m.Y - 1);
        rec.cell = ix * (uint32_t)m.Y + iy;
        rec.flags = (rk ? 3u : 2u) | (ray_cone_bound(rec.dx, rec.dy, rec.dz, precision) << 16);
        bin = (rk ? rocks_bin_offset : 0u) + rec.cell;
    }
    rays[i] = rec;
    if (bin_out) bin_out[i] = bin;
}

// ---------------------------------------------------------------------------------------------------
// collision mask + reward + extras + done: one thread per env (rover.py:663-668, 460-531, 610-647)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_count_flags(bool flag, uint32_t* __restrict__ block_cnt, uint32_t bid) {
    __shared__ uint32_t wcnt[4];
    unsigned long long ballot = __ballot(flag);
    if ((threadIdx.x & 63u) == 0u) wcnt[threadIdx.x >> 6] = __popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[bid] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

__device__ __forceinline__ bool collides_grid(const StoneGridDev& g, const float* __restrict__ info7, float x, float y, float thr);

__device__ __forceinline__ void metrics_done_env(const MetricsArgs& a, uint32_t e, bool& done_flag) {
    // Every load of the env first, every store after: the compiler may not move a load above a store it cannot prove
    // disjoint, so loads placed where they are used made the thread wait for memory five or six times in a row (one env per
    // thread: 1 024 waves, nothing else to run meanwhile).
    const bool need_progress = a.do_increment | a.do_metrics | a.do_done;
    int64_t progress = need_progress ? a.progress[e] : 0;
    const bool cast = a.do_collision && a.curriculum_level >= 2;
    float4 dw[6];
    float2 db = make_float2(0.0f, 0.0f);
    if (cast) {                                                             // rows are 32-byte aligned (R8 is a multiple of 8)
        const float4* d4 = reinterpret_cast<const float4*>(a.dist + (uint64_t)e * a.R8);
#pragma unroll
        for (int r = 0; r < 6; ++r) dw[r] = d4[r];
        db = *reinterpret_cast<const float2*>(a.dist + (uint64_t)e * a.R8 + 24u);
    }
    const float px = a.pos[3ull * e], py = a.pos[3ull * e + 1];
    const float tx = a.target[3ull * e] - px, ty = a.target[3ull * e + 1] - py;
    int64_t coll = a.do_collision ? 0 : a.rock_collision[e];
    float hd = 0.0f, j0 = 0.0f, j1 = 0.0f, j2 = 0.0f, lin = 0.0f, lin_prev = 0.0f, ang = 0.0f, ang_prev = 0.0f;
    if (a.do_metrics) {
        hd = a.heading[e];
        const float* jn = a.joints + 13ull * e;
        j0 = jn[0]; j1 = jn[1]; j2 = jn[2];
        lin = a.lin_hist[3ull * e]; lin_prev = a.lin_hist[3ull * e + 1];
        ang = a.ang_hist[3ull * e]; ang_prev = a.ang_hist[3ull * e + 1];
    }
    float ep0 = 0.0f, ep1 = 0.0f;
    if (a.do_done) { ep0 = a.euler_pre[3ull * e]; ep1 = a.euler_pre[3ull * e + 1]; }
    // additional output (not in the reference's step): stone_info occupancy mask at the rover's position,
    // the clearance test of rover.py:536-539 through the stone-occupancy grid (a chain of dependent loads: started here)
    const bool stone = (a.do_collision && a.stone_collision) ? collides_grid(a.sgrid, a.info7, px, py, a.stone_margin) : false;

    if (a.do_increment) { progress += 1; a.progress[e] = progress; }       // rl_task.py:250
    if (a.do_collision) {                                                   // check_collision, rover.py:663-668
        if (cast) {
            float mw = dw[0].x;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                if (r) mw = (dw[r].x < mw) ? dw[r].x : mw;
                mw = (dw[r].y < mw) ? dw[r].y : mw;
                mw = (dw[r].z < mw) ? dw[r].z : mw;
                mw = (dw[r].w < mw) ? dw[r].w : mw;
            }
            float mb = (db.y < db.x) ? db.y : db.x;
            coll = (fabsf(mw) < a.wheel_thr) ? 1 : 0;            // 0.8 / 0.45, as fp16 values in the as-shipped mode
            if (fabsf(mb) < a.body_thr) coll = 1;
        }
        a.rock_collision[e] = coll;
        if (a.stone_collision) a.stone_collision[e] = stone ? 1 : 0;
    }
    float td = sqrtf(tx * tx + ty * ty);                                    // :482 / :617
    if (a.do_metrics) {
        float heading_pen = ((lin < 0.0f) ? -1.0f : 0.0f) * a.heading_contraint_reward;           // :486
        float boogie = (fabsf(j0) + fabsf(j1) + fabsf(j2)) * a.boogie_contraint_reward;       // :492
        float goal_pen = (fabsf(hd) > 2.0f) ? -fabsf(hd * 0.3f * a.goal_angle_reward) : 0.0f;      // :495
        float dl = fabsf(lin * 3.0f - 3.0f * lin_prev), da = fabsf(ang * 3.0f - 3.0f * ang_prev);
        float p1 = (dl > 0.05f) ? dl * dl : 0.0f;                                                  // :498
        float p2 = (da > 0.05f) ? da * da : 0.0f;                                                  // :499
        float motion = (p1 * p1) * a.motion_contraint_reward;                                      // :500
        motion = motion + (p2 * p2) * a.motion_contraint_reward;                                   // :502
        float pos_rew = (1.0f / (1.0f + ((float)(0.33 * 0.33) * td) * td)) * a.pos_reward;         // :505
        if (td <= 0.18f) pos_rew = 1.03f * (float)((int64_t)a.max_episode_length - progress);      // :506
        float reward = pos_rew + heading_pen + motion + goal_pen;                                  // :512
        int64_t tracker = 0;
        if (a.curriculum_level >= 2 && coll == 1) { tracker = a.num_envs_global; reward = reward - 300.0f; }  // :517-519
        reward = reward / 3000.0f;                                                                 // :522
        a.rew[e] = reward;
        if (a.ex_pos_reward) a.ex_pos_reward[e] = pos_rew;
        if (a.ex_collision) a.ex_collision[e] = tracker;
        if (a.ex_upright) a.ex_upright[e] = boogie;
        if (a.ex_heading) a.ex_heading[e] = heading_pen;
        if (a.ex_motion) a.ex_motion[e] = motion;
        if (a.ex_goal_angle) a.ex_goal_angle[e] = goal_pen;
        if (a.ex_lin) a.ex_lin[e] = lin;
        if (a.ex_ang) a.ex_ang[e] = ang;
    }
    if (a.do_done) {                                                        // is_done, rover.py:610-647
        const float tilt = (float)(0.78 * 1.5);
        int64_t reset = (progress >= (int64_t)a.max_episode_length) ? 1 : 0;
        if (fabsf(ep0) >= tilt) reset = 1;
        if (fabsf(ep1) >= tilt) reset = 1;
        if (td >= 11.0f) reset = 1;
        if (td <= 0.18f) reset = 1;
        if (a.curriculum_level >= 2 && coll == 1) reset = 1;
        a.reset[e] = reset;
        if (a.done_u8) a.done_u8[e] = (uint8_t)reset;
        done_flag = reset != 0;
    }
}

__device__ __forceinline__ void metrics_done_block(const MetricsArgs& a, uint32_t bid) {
    uint32_t e = bid * 256u + threadIdx.x;
    bool done_flag = false;
    if (e < a.E) metrics_done_env(a, e, done_flag);
    if (a.block_cnt) block_count_flags(done_flag, a.block_cnt, bid);      // per-block done count for the compaction
}
__global__ void __launch_bounds__(256) metrics_done_kernel(MetricsArgs a) { metrics_done_block(a, blockIdx.x); }

// rover_step: both consumers of the ray distances in ONE launch — the first n_met workgroups are metrics_done_kernel's (one env per
// thread, a long chain of dependent loads: they start first), the rest assemble_obs_kernel's.  Neither reads what the other
// writes; side by side the 9 us of the metrics pass hide behind the obs pass instead of following it.
__global__ void __launch_bounds__(256) obs_metrics_kernel(ObsArgs o, MetricsArgs m, uint32_t n_met) {
    if (blockIdx.x < n_met) metrics_done_block(m, blockIdx.x);
    else assemble_obs_block(o, blockIdx.x - n_met, gridDim.x - n_met);
}

// ---------------------------------------------------------------------------------------------------
// done compaction: nonzero(reset_buf) ascending, no host sync.  Two passes over 256-env blocks:
//   count  (fused into metrics_done_kernel, or compact_count_kernel for the stand-alone call): per-block popcount
//   write  each block sums the counts of the blocks before it, ranks its own flags with ballot + popcount
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) compact_count_kernel(const int64_t* __restrict__ reset, uint32_t n,
                                                            uint32_t* __restrict__ block_cnt) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    block_count_flags((i < n) && (reset[i] != 0), block_cnt, blockIdx.x);
}

__global__ void __launch_bounds__(256) compact_write_kernel(const int64_t* __restrict__ reset, uint32_t n, int64_t offset,
                                                            const uint32_t* __restrict__ block_cnt, int64_t* __restrict__ ids,
                                                            int32_t* __restrict__ count) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t wcnt[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    // exclusive prefix of the preceding blocks' counts (<= a few thousand values: one strided pass)
    uint32_t part = 0;
    for (uint32_t b = tid; b < blockIdx.x; b += 256u) part += block_cnt[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    uint32_t i = blockIdx.x * 256u + tid;
    bool flag = (i < n) && (reset[i] != 0);
    unsigned long long ballot = __ballot(flag);
    if (lane == 0u) { wsum[w] = part; wcnt[w] = __popcll(ballot); }
    __syncthreads();
    uint32_t base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint32_t woff = 0;
#pragma unroll
    for (uint32_t v = 0; v < 4u; ++v) woff += (v < w) ? wcnt[v] : 0u;
    if (flag) ids[base + woff + __popcll(ballot & ((1ull << lane) - 1ull))] = offset + (int64_t)i;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *count = (int32_t)(base + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
}

__global__ void __launch_bounds__(256) quat_to_euler_kernel(const float* __restrict__ q, float* __restrict__ eul, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r, p, y;
    quat_to_euler(q + 4ull * i, r, p, y);
    eul[3ull * i] = r; eul[3ull * i + 1] = p; eul[3ull * i + 2] = y;
}

// ---------------------------------------------------------------------------------------------------
// reset path.  Stones are staged through LDS in tiles of STONE_TILE (x, y, r) triples.
// ---------------------------------------------------------------------------------------------------
#define STONE_TILE 1024

__device__ __forceinline__ float clearance_tiles(const float* __restrict__ info7, uint32_t S, float x, float y, bool live,
                                                 float* sx, float* sy, float* sr) {
    float best = __builtin_inff();
    for (uint32_t s0 = 0; s0 < S; s0 += STONE_TILE) {
        uint32_t cnt = min((uint32_t)STONE_TILE, S - s0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float* r = info7 + 7ull * (s0 + i);
            sx[i] = r[0]; sy[i] = r[1]; sr[i] = r[6];
        }
        __syncthreads();
        if (live) {
            for (uint32_t i = 0; i < cnt; ++i) {
                float dx = x - sx[i], dy = y - sy[i];
                float d = sqrtf(dx * dx + dy * dy) - sr[i];                 // rover.py:536-537 / :655-656
                best = (d < best) ? d : best;
            }
        }
    }
    return best;
}

// Decision form of the same test, min_s(|p - stone_s| - r_s) <= thr, through the stone-occupancy grid built at
// rover_set_stones: a grid cell lists every stone whose disc inflated by the largest threshold used (1.4 m, plus a
// margin far above f32 rounding) touches the cell, so stones outside the list cannot satisfy the inequality and the
// per-stone arithmetic on the listed ones is the reference's own (sqrt of the f32 sum of squares, minus r).
__device__ __forceinline__ bool collides_grid(const StoneGridDev& g, const float* __restrict__ info7, float x, float y, float thr) {
    float fx = (x - g.x0) * g.inv_cell, fy = (y - g.y0) * g.inv_cell;
    if (!(fx >= 0.0f) || !(fy >= 0.0f) || !(fx < (float)g.nx) || !(fy < (float)g.ny)) return false;   // also NaN
    uint32_t c = (uint32_t)fx * (uint32_t)g.ny + (uint32_t)fy;
    uint32_t k0 = g.cell_start[c], k1 = g.cell_start[c + 1];
    (void)info7;
    for (uint32_t k = k0; k < k1; k += 4u) {                                // 4 independent 16-B loads in flight per trip
        bool hit = false;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const float4 r = g.stone_xyr[min(k + j, k1 - 1u)];                 // past the end: the last entry again
            float dx = x - r.x, dy = y - r.y;
            float d = sqrtf(dx * dx + dy * dy) - r.z;                       // rover.py:536-537 / :655-656
            hit |= d <= thr;
        }
        if (hit) return true;
    }
    return false;
}

__global__ void __launch_bounds__(256) clearance_kernel(const float* __restrict__ info7, uint32_t S, const float* __restrict__ xy,
                                                        uint32_t n, float* __restrict__ out) {
    __shared__ float sx[STONE_TILE], sy[STONE_TILE], sr[STONE_TILE];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < n;
    float x = live ? xy[2ull * i] : 0.0f, y = live ? xy[2ull * i + 1] : 0.0f;
    float c = clearance_tiles(info7, S, x, y, live, sx, sy, sr);
    if (live) out[i] = c;
}

// avoid_pos_rock_collision rover.py:649-661: per env, x += 0.05 while the stone clearance is <= 1.4.  (The reference
// iterates over all envs until nothing moves; per env that is this loop.)
__global__ void __launch_bounds__(256) shift_spawns_kernel(StoneGridDev g, const float* __restrict__ info7, float* __restrict__ pos3,
                                                           uint32_t n, int32_t max_iter) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = pos3[3ull * i], y = pos3[3ull * i + 1];
    for (int32_t it = 0; it < max_iter && collides_grid(g, info7, x, y, 1.4f); ++it) x = x + 0.05f;   // :660
    pos3[3ull * i] = x;
}

// get_pos_height rover.py:588-608
__global__ void __launch_bounds__(256) sample_height_kernel(HeightDev h, const float* __restrict__ xy, uint32_t n,
                                                            float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ix = cell_coord(xy[2ull * i], h.shift_x, h.hscale, h.inv_hscale, h.rcp, h.N0);
    uint32_t iy = cell_coord(xy[2ull * i + 1], h.shift_y, h.hscale, h.inv_hscale, h.rcp, h.N0);
    if (iy > (uint32_t)(h.N1 - 1)) iy = (uint32_t)(h.N1 - 1);
    out[i] = h.hm[(uint64_t)ix * h.N1 + iy] * h.vscale;
}

// Philox4x32-10 (Salmon et al. 2011), used when the caller supplies no uniforms
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t draw, uint32_t idx) {
    uint32_t c0 = idx, c1 = draw, c2 = 0u, c3 = 0u, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);                        // [0,1) with 24 bits, like torch.rand f32
}

// generate_goals rover.py:544-549 (+ random_goals :554-564, check_goal_collision :533-542, goal z :581-583).
//
// The reference loops globally: draw goals for every listed env, test them against the stone list, and — the
// env_ids = mask*env_ids aliasing of :540 — turn the id of every ACCEPTED entry into 0, so accepted entries keep
// re-randomising env 0's goal and the loop also waits for env 0's goal to be clear.  Unrolled per entry this is:
//   * entry i keeps its own first clear draw u[t_i][i] (later iterations no longer write its env);
//   * the loop ends at the first iteration T >= max_i t_i at which env 0's goal — written by the LAST zero-id entry
//     of that iteration ("last writer wins", what a sequential index_put gives) — is clear, or at max_i t_i when no
//     zero-id entry exists yet.
// goals_draw_kernel does the per-entry part in parallel over all workgroups (stones staged through LDS tiles);
// goals_env0_kernel replays env 0's part on one workgroup.  Same draws, same results as the sequential loop
// (oracle_generate_goals); n may come from device memory so no host sync is needed.
__device__ __forceinline__ void goal_from_draw(float u, float radius, const float* __restrict__ initial_pos3, int64_t id,
                                               float& x, float& y) {
    float alpha = (float)(2 * 3.14159265358979323846) * u;                  // :556
    float gx = radius * cosf(alpha) + 0.0f, gy = radius * sinf(alpha) + 0.0f;   // :560-561
    x = gx + initial_pos3[3ull * id];                                       // :563-564
    y = gy + initial_pos3[3ull * id + 1];
}

__device__ __forceinline__ float height_at(const HeightDev& h, float x, float y) {
    uint32_t ix = cell_coord(x, h.shift_x, h.hscale, h.inv_hscale, h.rcp, h.N0);
    uint32_t iy = cell_coord(y, h.shift_y, h.hscale, h.inv_hscale, h.rcp, h.N0);
    if (iy > (uint32_t)(h.N1 - 1)) iy = (uint32_t)(h.N1 - 1);
    return h.hm[(uint64_t)ix * h.N1 + iy] * h.vscale;
}

// GOAL_LANES lanes share one entry and test GOAL_LANES consecutive draws at once (the first clear one wins, exactly what the
// sequential loop finds): with dense stones most draws collide and the per-entry chain of draw -> grid lookup was the cost.
#define GOAL_LANES 8u

__global__ void __launch_bounds__(256) goals_draw_kernel(GoalArgs a) {
    const uint32_t n = a.n_dev ? (uint32_t)max(*a.n_dev, 0) : a.n_host;
    const uint32_t gt = blockIdx.x * 256u + threadIdx.x;
    const uint32_t i = gt / GOAL_LANES, j = gt % GOAL_LANES;
    const uint32_t shift = (threadIdx.x & 63u) & ~(GOAL_LANES - 1u);             // first lane of this entry's group
    const bool live = i < n;
    const int64_t id = live ? a.env_ids[i] - a.id_offset : 0;
    if (live && id == 0 && j == 0) a.t_acc[i] = -1;                            // zero-id entry from the start
    bool active = live && id != 0;
    // whole waves iterate together (ballot); finished / idle groups just stop contributing
    for (int32_t t0 = 0; t0 < a.max_draws; t0 += (int32_t)GOAL_LANES) {
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        const int32_t t = t0 + (int32_t)j;
        bool clear = false;
        float x = 0.0f, y = 0.0f;
        if (active && t < a.max_draws) {
            // library RNG: keyed by the entry's GLOBAL env id, so that the shards of a multi-GPU run draw different streams
            float u = a.draws ? a.draws[(uint64_t)t * n + i] : philox_uniform(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), (uint32_t)t, (uint32_t)a.env_ids[i]);
            goal_from_draw(u, a.radius, a.initial_pos3, id, x, y);
            clear = !collides_grid(a.grid, a.info7, x, y, 1.0f);               // :539: redraw while nearest_rock <= 1.0
        }
        const uint32_t m = (uint32_t)(__builtin_amdgcn_ballot_w64(clear) >> shift) & ((1u << GOAL_LANES) - 1u);
        if (active) {
            const bool last_trip = t0 + (int32_t)GOAL_LANES >= a.max_draws;
            int32_t winner = -1;                                               // lane of the group whose draw is kept
            if (m) winner = (int32_t)__builtin_ctz(m);
            else if (last_trip) winner = a.max_draws - 1 - t0;                 // never clear: the last draw stays (t_acc = max)
            if (winner >= 0) {
                if ((int32_t)j == winner) {
                    a.t_acc[i] = m ? t : a.max_draws;
                    a.target3[3ull * id] = x; a.target3[3ull * id + 1] = y;
                    a.target3[3ull * id + 2] = height_at(a.h, x, y);            // set_targets :581-583
                }
                active = false;
            }
        }
    }
}

__global__ void __launch_bounds__(1024) goals_env0_kernel(GoalArgs a) {
    __shared__ int s_tmax, s_lz, s_has_zero, s_fail;
    const uint32_t tid = threadIdx.x;
    const uint32_t n = a.n_dev ? (uint32_t)max(*a.n_dev, 0) : a.n_host;
    if (n == 0) { if (tid == 0 && a.n_draws_used) *a.n_draws_used = 0; return; }
    if (tid == 0) { s_tmax = -1; s_lz = -1; s_has_zero = 0; s_fail = 0; }
    __syncthreads();
    // per-thread partial results, one wave reduction, one LDS atomic per wave (n can be all 65 536 envs)
    int my_tmax = -1;
    bool my_zero = false, my_fail = false;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        int32_t t = a.t_acc[i];
        if (t >= a.max_draws) my_fail = true;
        else if (t < 0) my_zero = true;
        else my_tmax = max(my_tmax, t);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_tmax = max(my_tmax, __shfl_xor(my_tmax, o));
    const bool w_zero = __builtin_amdgcn_ballot_w64(my_zero) != 0, w_fail = __builtin_amdgcn_ballot_w64(my_fail) != 0;
    if ((tid & 63u) == 0u) {
        if (my_tmax >= 0) atomicMax(&s_tmax, my_tmax);
        if (w_zero) s_has_zero = 1;
        if (w_fail) s_fail = 1;
    }
    __syncthreads();
    if (s_fail) { if (tid == 0 && a.n_draws_used) *a.n_draws_used = -1; return; }
    const int tmax = s_tmax;                          // -1 when every entry was a zero-id entry
    // zero-id set before the draw of iteration max(tmax, 0): original zeros and entries accepted earlier
    int my_lz = -1;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        int32_t t = a.t_acc[i];
        if (t < 0 || t < tmax) my_lz = max(my_lz, (int)i);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_lz = max(my_lz, __shfl_xor(my_lz, o));
    if ((tid & 63u) == 0u && my_lz >= 0) atomicMax(&s_lz, my_lz);
    __syncthreads();
    if (tid >= 64u) return;
    // env 0's own sequence, 64 iterations per trip on one wave: iteration t0 is written by entry lz, every later one by
    // entry n - 1 (from iteration tmax + 1 on every entry has id 0); the first clear draw ends the loop.
    const int32_t t0 = max(tmax, 0);
    const int lz0 = s_lz;
    int32_t used = -1;
    float x = 0.0f, y = 0.0f;
    bool wrote = false;
    if (lz0 < 0) {
        used = t0 + 1;                                // nobody writes env 0 in this iteration: bad == 0
    } else {
        for (int32_t base = t0; base < a.max_draws; base += 64) {
            const int32_t t = base + (int32_t)tid;
            const uint32_t src = (t == t0) ? (uint32_t)lz0 : n - 1u;
            bool clear = false;
            float cx = 0.0f, cy = 0.0f;
            if (t < a.max_draws) {
                float u = a.draws ? a.draws[(uint64_t)t * n + src] : philox_uniform(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), (uint32_t)t, (uint32_t)a.env_ids[src]);
                goal_from_draw(u, a.radius, a.initial_pos3, 0, cx, cy);
                clear = !collides_grid(a.grid, a.info7, cx, cy, 1.0f);
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(clear);
            const bool last_trip = base + 64 >= a.max_draws;
            int32_t winner = -1;
            if (m) winner = (int32_t)__builtin_ctzll(m);
            else if (last_trip) winner = a.max_draws - 1 - base;
            if (winner >= 0) {
                x = __shfl(cx, winner); y = __shfl(cy, winner);
                wrote = true;
                if (m) used = base + winner + 1;
                break;
            }
        }
    }
    if (tid != 0) return;
    if (wrote) {
        a.target3[0] = x; a.target3[1] = y;
        if (s_has_zero) a.target3[2] = height_at(a.h, x, y);               // env 0 was listed: set_targets :581-583
    }
    if (a.n_draws_used) *a.n_draws_used = used;
}

// reset_idx rover.py:416-453 for the compacted reset ids, count read from device memory (no host sync).
// Orientation: the reference feeds scipy's (x,y,z,w) of a rotation about x to Isaac as (w,x,y,z) (:429-431,:449),
// i.e. w = sin(d/2), z = cos(d/2) with d = randint(0, 360) degrees.  yaw_deg (optional) replaces the draw.
__global__ void __launch_bounds__(256) reset_envs_kernel(ResetArgs a) {
    const uint32_t n = a.n_dev ? (uint32_t)max(*a.n_dev, 0) : a.n_host;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t id = a.ids[i] - a.id_offset;
    float deg = a.yaw_deg ? (float)a.yaw_deg[i]
                          : floorf(philox_uniform((a.seed + (a.seed_dev ? *a.seed_dev : 0ull)) ^ 0x9E3779B97F4A7C15ull, 0u, (uint32_t)a.ids[i]) * 361.0f);   // global id
    float half = (deg * (3.14159265358979323846f / 180.0f)) / 2.0f;
    a.quat4[4ull * id] = sinf(half); a.quat4[4ull * id + 1] = 0.0f; a.quat4[4ull * id + 2] = 0.0f; a.quat4[4ull * id + 3] = cosf(half);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = a.initial_pos3[3ull * id + c];
        a.pos3[3ull * id + c] = v;
        if (a.base_pos3) a.base_pos3[3ull * id + c] = v;
    }
    for (int j = 0; j < 13; ++j) {
        if (a.joint_pos13) a.joint_pos13[13ull * id + j] = 0.0f;
        if (a.joint_vel13) a.joint_vel13[13ull * id + j] = 0.0f;
    }
    a.reset[id] = 0;                                                        // :452-453
    a.progress[id] = 0;
}

// pre_physics_step rover.py:338-414 without the reset branch: pre-physics euler (:343), Memory.input_state x2
// (:379-380), Ackermann (:391) and the scatter into the joint-target layout (:400-414, rover_view.py:45-46).
__global__ void __launch_bounds__(256) pre_physics_kernel(PrePhysicsArgs a) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.E) return;
    if (a.euler_pre) {
        float r, p, y;
        quat_to_euler(a.quat + 4ull * e, r, p, y);
        a.euler_pre[3ull * e] = r; a.euler_pre[3ull * e + 1] = p; a.euler_pre[3ull * e + 2] = y;
    }
    float lin = a.actions[2ull * e], ang = a.actions[2ull * e + 1];
    float l0 = a.lin_hist[3ull * e], l1 = a.lin_hist[3ull * e + 1];
    a.lin_hist[3ull * e] = lin; a.lin_hist[3ull * e + 1] = l0; a.lin_hist[3ull * e + 2] = l1;
    float g0 = a.ang_hist[3ull * e], g1 = a.ang_hist[3ull * e + 1];
    a.ang_hist[3ull * e] = ang; a.ang_hist[3ull * e + 1] = g0; a.ang_hist[3ull * e + 2] = g1;
    if (a.actions_nn) {                         // rover.py:366: cat((actions[:, :, None], actions_nn), 2)[:, :, 0:3]
        float* n = a.actions_nn + 6ull * e;
        const float n0 = n[0], n1 = n[1], n3 = n[3], n4 = n[4];
        n[0] = lin; n[1] = n0; n[2] = n1; n[3] = ang; n[4] = n3; n[5] = n4;
    }
    if (!a.pos_targets13 && !a.vel_targets13) return;
    // Ackermann, tasks/utils/kinematics.py:13-67 (same operation order as ackermann_kernel)
    const float wl[6][2] = {{-0.385, 0.438}, {0.385, 0.438}, {-0.447, 0.0}, {0.447, 0.0}, {-0.385, -0.411}, {0.385, -0.411}};
    const float side[6] = {-1.0f, 1.0f, -1.0f, 1.0f, -1.0f, 1.0f};
    float Px = copysignf(lin / ang, -ang);
    Px = (fabsf(Px) > 0.45f) ? Px : 0.0f;
    lin = (Px != 0.0f) ? lin : 0.0f;
    float steer[6], vel[6];
#pragma unroll
    for (int w = 0; w < 6; ++w) {
        float dx = Px - wl[w][0], dy = 0.0f - wl[w][1];
        float dist = sqrtf(dx * dx + dy * dy);
        float av = (lin != 0.0f) ? copysignf(ang, lin) : ang * side[w];
        float mv = dist * av;
        if (dist > 1000.0f) mv = lin;
        vel[w] = mv / 0.2f;
        float sa = atan2f(wl[w][1], wl[w][0] - Px);
        if (sa < (float)(-3.14 / 2)) sa = sa + 3.14159265358979323846f;
        if (sa > (float)(3.14 / 2)) sa = sa - 3.14159265358979323846f;
        steer[w] = sa;
    }
    if (a.pos_targets13) {                      // positions [FR, RR, FL, RL] = steer[1,5,0,4] -> joints [6,8,4,7]
        float* p = a.pos_targets13 + 13ull * e;
        p[6] = steer[1]; p[8] = steer[5]; p[4] = steer[0]; p[7] = steer[4];
    }
    if (a.vel_targets13) {                      // velocities [FR,CR,RR,FL,CL,RL] = vel[1,3,5,0,2,4] -> joints [10,5,12,9,3,11]
        float* v = a.vel_targets13 + 13ull * e;
        v[10] = vel[1]; v[5] = vel[3]; v[12] = vel[5]; v[9] = vel[0]; v[3] = vel[2]; v[11] = vel[4];
    }
}

// Ackermann, tasks/utils/kinematics.py:13-67
__global__ void __launch_bounds__(256) ackermann_kernel(const float* __restrict__ lin_in, const float* __restrict__ ang_in,
                                                        uint32_t n, float* __restrict__ steer, float* __restrict__ vel) {
    const float wl[6][2] = {{-0.385, 0.438}, {0.385, 0.438}, {-0.447, 0.0}, {0.447, 0.0}, {-0.385, -0.411}, {0.385, -0.411}};
    const float side[6] = {-1.0f, 1.0f, -1.0f, 1.0f, -1.0f, 1.0f};
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lin = lin_in[i], ang = ang_in[i];
    float Px = copysignf(lin / ang, -ang);
    Px = (fabsf(Px) > 0.45f) ? Px : 0.0f;
    lin = (Px != 0.0f) ? lin : 0.0f;
#pragma unroll
    for (int w = 0; w < 6; ++w) {
        float dx = Px - wl[w][0], dy = 0.0f - wl[w][1];
        float dist = sqrtf(dx * dx + dy * dy);
        float av = (lin != 0.0f) ? copysignf(ang, lin) : ang * side[w];
        float mv = dist * av;
        if (dist > 1000.0f) mv = lin;
        vel[6ull * i + w] = mv / 0.2f;
        float sa = atan2f(wl[w][1], wl[w][0] - Px);
        if (sa < (float)(-3.14 / 2)) sa = sa + 3.14159265358979323846f;
        if (sa > (float)(3.14 / 2)) sa = sa - 3.14159265358979323846f;
        steer[6ull * i + w] = sa;
    }
}

// ---------------------------------------------------------------------------------------------------
// ray binning: sort of the step's rays by (map, cell).  At 65 536 envs a bin holds ~4-5 rays per step; sorted, the
// 3.6 KB cell block is fetched from HBM once per bin and served from registers to its rays.
// Results do not depend on the order inside a bin, so the LDS atomics do not make the step non-deterministic.
// The generic 3-level scan below now only serves the KNN map builder; the binning has its own row scan.
// ---------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

// exclusive scan of one value per thread across a workgroup of NW waves; returns the exclusive prefix, total via ref
template <int NW>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /*[NW]*/, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    if (lane == 63u) lds[w] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { uint32_t c = lds[i]; wbase += ((uint32_t)i < w) ? c : 0u; tot += c; }
    __syncthreads();
    total = tot;
    return wbase + inc - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_block_sums_kernel(const uint32_t* __restrict__ cnt, uint32_t n,
                                                                     uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS, sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) sum += (base + i < n) ? cnt[base + i] : 0u;
    uint32_t total;
    (void)block_exclusive_scan<SCAN_BLOCK / 64>(sum, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one workgroup: exclusive scan in place of up to 1024*SCAN_ITEMS block sums
__global__ void __launch_bounds__(1024) scan_top_kernel(uint32_t* __restrict__ block_sums, uint32_t nb) {
    __shared__ uint32_t lds[16];
    uint32_t base = threadIdx.x * SCAN_ITEMS, v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < nb) ? block_sums[base + i] : 0u; sum += v[i]; }
    uint32_t total, run = block_exclusive_scan<16>(sum, lds, total);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < nb) block_sums[base + i] = run; run += v[i]; }
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(uint32_t* __restrict__ cnt, uint32_t n,
                                                                const uint32_t* __restrict__ block_offsets) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS, v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? cnt[base + i] : 0u; sum += v[i]; }
    uint32_t total, run = block_exclusive_scan<SCAN_BLOCK / 64>(sum, lds, total) + block_offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) cnt[base + i] = run; run += v[i]; }
}

// ---- bucket sort of the ray slots by bin = (map, cell), without returning global atomics ---------------------------
// (the only global atomics are the few per (tile, bucket) of the histogram pass fused into prep_rays_kernel: bin_hist_fused)
//   bucket_hist_kernel     per 4096-slot block: LDS histogram over the coarse buckets (bin >> low_bits) -> counts[block][bucket]
//                          (or prep_rays_kernel itself: bin_hist_fused)
//   bucket_rowscan_kernel  one workgroup per bucket: exclusive prefix along its row of block counts + the bucket total
//                          (the scan over the bucket totals is redone by every scatter block in LDS: <= 4096 values)
//   bucket_scatter_kernel  same blocks: (bin, slot) pairs to their bucket range, position from an LDS cursor per bucket
//   bucket_sort_kernel     one workgroup per bucket: LDS counting sort on the low bits -> sorted slot ids
// The order of equal bins is whatever the LDS atomics produce; the ray cast does not depend on it.
#define BKT_ITEMS 16
#define BKT_TILE (256 * BKT_ITEMS)
#define BKT_MAX 4096            // max coarse buckets, and max 2^low_bits
#define BKT_STAGE 8192u         // entries of a bucket that bucket_sort_kernel orders in LDS before writing them out

// NT threads take a tile of NT x 16 slots: 256 (4 096 slots), or 1 024 for big ray sets (launch_bin_rays)
template <int NT>
__global__ void __launch_bounds__(NT) bucket_hist_kernel(const uint32_t* __restrict__ bins, uint32_t n_slots, uint32_t low_bits,
                                                          uint32_t n_buckets, uint32_t n_blocks, uint32_t* __restrict__ counts) {
    __shared__ uint32_t h[BKT_MAX];
    for (uint32_t i = threadIdx.x; i < n_buckets; i += NT) h[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (NT * BKT_ITEMS);
    uint32_t bv[BKT_ITEMS];                                   // all 16 loads in flight before the first atomic (the kernel is latency-bound)
#pragma unroll
    for (uint32_t it = 0; it < BKT_ITEMS; ++it) {
        const uint32_t i = base + it * NT + threadIdx.x;
        bv[it] = i < n_slots ? bins[i] : 0xffffffffu;
    }
#pragma unroll
    for (uint32_t it = 0; it < BKT_ITEMS; ++it)
        if (bv[it] != 0xffffffffu) atomicAdd(&h[bv[it] >> low_bits], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_buckets; i += NT) counts[(size_t)blockIdx.x * n_buckets + i] = h[i];   // [block][bucket]: coalesced
}

// counts[bucket][block] -> exclusive prefix inside every bucket row (one workgroup per bucket) + the row total
// counts [block][bucket] -> exclusive prefix over the BLOCKS, in place, and the bucket totals.  One workgroup per 16 buckets, 16 waves;
// a wave takes 64 consecutive blocks of a 1 024-block tile: lane (kb = lane & 15, ph = lane >> 4) holds bucket kb of blocks b0 + 4 j + ph,
// j < 16, in registers — one round of loads (four 64-byte rows per instruction), a register / cross-lane scan, one round of stores.
// (The table was [bucket][block] with one workgroup scanning a row: the histogram's 720 k stores and the scatter's 720 k loads were
// 4 bytes at a 4 KB stride; as [block][bucket] with one lane per bucket and 64 buckets per workgroup only 11 CUs worked: 10.6 us.)
#define RSCAN_WAVES 16
__global__ void __launch_bounds__(64 * RSCAN_WAVES) bucket_rowscan_kernel(uint32_t* __restrict__ counts, uint32_t n_blocks, uint32_t n_buckets,
                                                                          uint32_t* __restrict__ bucket_tot) {
    __shared__ uint32_t part[RSCAN_WAVES][16];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, kb = lane & 15u, ph = lane >> 4;
    const uint32_t k = blockIdx.x * 16u + kb;
    const bool kin = k < n_buckets;
    uint32_t carry = 0;                                              // rays of this bucket in the tiles before
    for (uint32_t tile0 = 0; tile0 < n_blocks; tile0 += 64u * RSCAN_WAVES) {
        const uint32_t b0 = tile0 + w * 64u + ph;                    // this lane's blocks: b0 + 4 j
        uint32_t* __restrict__ p = counts + (size_t)b0 * n_buckets + (kin ? k : 0u);
        uint32_t v[16], ex[16], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j) v[j] = (kin && b0 + 4u * j < n_blocks) ? p[(size_t)(4u * j) * n_buckets] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j) {
            // blocks 4 j .. 4 j + 3 sit in the four lane groups: inclusive scan over ph, then this lane's exclusive share
            uint32_t inc = v[j];
            uint32_t o = (uint32_t)__shfl_up((int)inc, 16, 64); if (ph >= 1u) inc += o;
            o = (uint32_t)__shfl_up((int)inc, 32, 64);          if (ph >= 2u) inc += o;
            const uint32_t four = (uint32_t)__shfl((int)inc, 48 + (int)kb, 64);       // all four blocks of this j
            ex[j] = sum + inc - v[j];
            sum += four;
        }
        if (ph == 0u) part[w][kb] = sum;
        __syncthreads();
        uint32_t run = carry, total = 0;
#pragma unroll
        for (uint32_t i = 0; i < RSCAN_WAVES; ++i) { const uint32_t c = part[i][kb]; run += i < w ? c : 0u; total += c; }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j)
            if (kin && b0 + 4u * j < n_blocks) p[(size_t)(4u * j) * n_buckets] = run + ex[j];
        carry += total;
    }
    if (w == 0 && ph == 0u && kin) bucket_tot[k] = carry;
}

// PACKED: an entry is one dword, low bin bits << (32 - low_bits) | slot — all bucket_sort_kernel reads of a pair — when the slot
// ids leave room for them (n_slots <= 2^(32 - low_bits): 65 536 envs x 64 slots with 1 024 bins per bucket just fit); half the
// scatter's writes and half the sort's reads.  Otherwise (bin, slot) as two dwords.
template <bool PACKED> struct BktEntry;
template <> struct BktEntry<false> {
    typedef uint2 T;
    static __device__ __forceinline__ T make(uint32_t bin, uint32_t slot, uint32_t) { return make_uint2(bin, slot); }
    static __device__ __forceinline__ T none() { return make_uint2(0u, 0xffffffffu); }
    static __device__ __forceinline__ uint32_t lo(T v, uint32_t low_bits) { return v.x & ((1u << low_bits) - 1u); }
    static __device__ __forceinline__ uint32_t slot(T v, uint32_t) { return v.y; }
};
template <> struct BktEntry<true> {
    typedef uint32_t T;
    static __device__ __forceinline__ T make(uint32_t bin, uint32_t slot, uint32_t low_bits) { return (bin << (32u - low_bits)) | slot; }
    static __device__ __forceinline__ T none() { return 0xffffffffu; }
    static __device__ __forceinline__ uint32_t lo(T v, uint32_t low_bits) { return v >> (32u - low_bits); }
    static __device__ __forceinline__ uint32_t slot(T v, uint32_t low_bits) { return v & (0xffffffffu >> low_bits); }
};

// Every block turns the bucket totals into bucket start offsets in LDS (n_buckets <= 4096: 16 values per thread); block 0 also
// publishes them (plus the grand total) for bucket_sort_kernel.  The block's 4 096 entries are first ordered by bucket in LDS
// (rank from one LDS atomic per entry, local start from a scan of the block's bucket counts) next to their global positions, then
// written out in that order: neighbouring lanes then write neighbouring entries of a bucket's run (~6 per bucket and block) and the
// run leaves as one or two write transactions instead of one per entry.
template <bool PACKED, int NT>
__global__ void __launch_bounds__(NT) bucket_scatter_kernel(const uint32_t* __restrict__ bins, uint32_t n_slots, uint32_t low_bits,
                                                             uint32_t n_buckets, uint32_t n_blocks, const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ bucket_tot, uint32_t* __restrict__ bucket_base,
                                                             uint2* __restrict__ pairs2) {
    typedef BktEntry<PACKED> En;
    typename En::T* __restrict__ pairs = reinterpret_cast<typename En::T*>(pairs2);
    extern __shared__ __attribute__((aligned(8))) uint32_t bkt_lds[];        // sized by the launch: 2 n_buckets + (NT * BKT_ITEMS) (1 + dwords per entry) dwords
    uint32_t* const cur = bkt_lds;               // global position of this block's first entry in each bucket
    uint32_t* const cnt = cur + n_buckets;       // this block's entries per bucket -> their local start
    uint32_t* const s_dst = cnt + n_buckets;
    typename En::T* const s_val = reinterpret_cast<typename En::T*>(s_dst + (NT * BKT_ITEMS));      // (an even dword offset from an 8-byte aligned base: two-dword entries stay 8-byte aligned)
    __shared__ uint32_t wl[NT / 64];
    const uint32_t per = (n_buckets + NT - 1u) / NT, first = threadIdx.x * per;
    const uint32_t base = blockIdx.x * (NT * BKT_ITEMS);
    uint32_t bv[BKT_ITEMS], rk[BKT_ITEMS];                    // the block's 16 loads per thread go out first: their latency passes
#pragma unroll                                               // under the bucket-offset scan below
    for (uint32_t it = 0; it < BKT_ITEMS; ++it) {
        const uint32_t i = base + it * NT + threadIdx.x;
        bv[it] = i < n_slots ? bins[i] : 0xffffffffu;
    }
    {
        uint32_t sum = 0;
        for (uint32_t j = 0; j < per; ++j) sum += (first + j < n_buckets) ? bucket_tot[first + j] : 0u;
        uint32_t total, run = block_exclusive_scan<NT / 64>(sum, wl, total);
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t i = first + j;
            if (i < n_buckets) {
                cur[i] = run + offsets[(size_t)blockIdx.x * n_buckets + i];
                cnt[i] = 0u;
                if (blockIdx.x == 0) bucket_base[i] = run;
                run += bucket_tot[i];
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) bucket_base[n_buckets] = total;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t it = 0; it < BKT_ITEMS; ++it) rk[it] = bv[it] != 0xffffffffu ? atomicAdd(&cnt[bv[it] >> low_bits], 1u) : 0u;
    __syncthreads();
    uint32_t n_here;
    {
        uint32_t sum = 0;
        for (uint32_t j = 0; j < per; ++j) sum += (first + j < n_buckets) ? cnt[first + j] : 0u;
        uint32_t run = block_exclusive_scan<NT / 64>(sum, wl, n_here);
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t i = first + j;
            if (i < n_buckets) { const uint32_t c = cnt[i]; cnt[i] = run; run += c; }
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t it = 0; it < BKT_ITEMS; ++it) {
        if (bv[it] != 0xffffffffu) {
            const uint32_t b = bv[it] >> low_bits, l = cnt[b] + rk[it];
            s_dst[l] = cur[b] + rk[it];
            s_val[l] = En::make(bv[it], base + it * NT + threadIdx.x, low_bits);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_here; k += NT) pairs[s_dst[k]] = s_val[k];
}

template <bool PACKED, int NT, int IPT>      // IPT: items a thread may hold (buckets of up to NT x IPT entries take the one-pass path)
__global__ void __launch_bounds__(NT) bucket_sort_kernel(const uint2* __restrict__ pairs2, const uint32_t* __restrict__ bucket_base,
                                                          uint32_t low_bits, uint32_t* __restrict__ sorted, uint32_t* __restrict__ zero,
                                                          uint32_t n_zero) {
    typedef BktEntry<PACKED> En;
    const typename En::T* __restrict__ pairs = reinterpret_cast<const typename En::T*>(pairs2);
    extern __shared__ uint32_t bkt_lds[];        // sized by the launch: 2^low_bits + BKT_STAGE dwords
    __shared__ uint32_t wl[NT / 64];
    const uint32_t b = blockIdx.x, tid = threadIdx.x, nl = 1u << low_bits;
    if (zero) {      // the count table is spent (the scatter has run): cleared for the next step's prep_rays_kernel, a share per block
        const uint32_t per = (n_zero + gridDim.x - 1u) / gridDim.x, z1 = min(n_zero, (b + 1u) * per);
        for (uint32_t i = b * per + tid; i < z1; i += NT) zero[i] = 0u;
    }
    uint32_t* const h = bkt_lds;
    uint32_t* const stage = bkt_lds + nl;
    const uint32_t s0 = bucket_base[b], s1 = bucket_base[b + 1u];
    if (s1 <= s0) return;
    for (uint32_t i = tid; i < nl; i += NT) h[i] = 0;
    __syncthreads();
    if (s1 - s0 <= (uint32_t)(NT * IPT)) {
        // The usual bucket (6-12 k rays): ONE pass of LDS atomics.  A thread keeps its <= 64 items in registers with the rank the
        // histogram atomic returned (rank inside the bin), so after the scan the position is start[bin] + rank — no second read of
        // the pairs and no second round of atomics.
        uint32_t slot[IPT], br[IPT];                                 // br = low bin bits | rank << 12
#pragma unroll
        for (int g = 0; g < IPT / 8; ++g) {
            typename En::T pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const uint32_t k = s0 + tid + (uint32_t)(8 * g + j) * NT; pv[j] = k < s1 ? pairs[k] : En::none(); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = s0 + tid + (uint32_t)(8 * g + j) * NT, lo = En::lo(pv[j], low_bits);
                slot[8 * g + j] = En::slot(pv[j], low_bits);
                br[8 * g + j] = k < s1 ? (lo | (atomicAdd(&h[lo], 1u) << 12)) : 0u;
            }
        }
        __syncthreads();
        const uint32_t per = (nl + NT - 1u) / NT, first = tid * per;
        uint32_t sum = 0;
        for (uint32_t j = 0; j < per; ++j) sum += first + j < nl ? h[first + j] : 0u;
        uint32_t total, run = block_exclusive_scan<NT / 64>(sum, wl, total);
        for (uint32_t j = 0; j < per; ++j) if (first + j < nl) { const uint32_t c = h[first + j]; h[first + j] = run; run += c; }
        __syncthreads();
        // through LDS: a wave's 64 positions are anywhere in the bucket's 24-48 KB of output, 64 partial-line writes per store
        // instruction (4.1 M L2 write transactions a launch); staged, the bucket leaves in whole lines
        if (s1 - s0 <= BKT_STAGE) {
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const uint32_t k = s0 + tid + (uint32_t)i * NT;
                if (k < s1) stage[h[br[i] & 0xfffu] + (br[i] >> 12)] = slot[i];
            }
            __syncthreads();
            for (uint32_t k = tid; k < s1 - s0; k += NT) sorted[s0 + k] = stage[k];
            return;
        }
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t k = s0 + tid + (uint32_t)i * NT;
            if (k < s1) sorted[s0 + h[br[i] & 0xfffu] + (br[i] >> 12)] = slot[i];
        }
        return;
    }
    // eight loads in flight per thread before their atomics: the kernel waits on memory 93 % of the time otherwise
    for (uint32_t k0 = s0 + tid; k0 < s1; k0 += 8u * NT) {
        typename En::T bx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint32_t k = k0 + (uint32_t)j * NT; bx[j] = k < s1 ? pairs[k] : En::none(); }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + (uint32_t)j * NT < s1) atomicAdd(&h[En::lo(bx[j], low_bits)], 1u);
    }
    __syncthreads();
    // exclusive scan of h[0..nl): each thread owns ceil(nl / NT) consecutive entries
    const uint32_t per = (nl + NT - 1u) / NT, first = tid * per;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; ++j) sum += first + j < nl ? h[first + j] : 0u;
    uint32_t total, run = block_exclusive_scan<NT / 64>(sum, wl, total);
    for (uint32_t j = 0; j < per; ++j) if (first + j < nl) { const uint32_t c = h[first + j]; h[first + j] = run; run += c; }
    __syncthreads();
    for (uint32_t k0 = s0 + tid; k0 < s1; k0 += 8u * NT) {
        typename En::T pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint32_t k = k0 + (uint32_t)j * NT; pv[j] = k < s1 ? pairs[k] : En::none(); }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + (uint32_t)j * NT < s1) sorted[s0 + atomicAdd(&h[En::lo(pv[j], low_bits)], 1u)] = En::slot(pv[j], low_bits);
    }
}

// ---------------------------------------------------------------------------------------------------
// ray cast over the binned rays.  One wave walks RUN consecutive sorted rays.  When the cell changes the
// wave loads the cell block (4 triangles per lane, 9 x 8-byte loads) and sets up a, b = v1-a, c = v0-a, b x c
// in registers; every further ray of the same cell only pays the ray-dependent part of ray_casting.py:37-59.
// The three quotients share one reciprocal refinement: the exact instruction sequence of the IEEE f32
// division expansion without its range scaling (den and quotients here are far from the f32 range ends),
// so results stay bit-identical to n = N/det, m = M/det, k = K/det.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) raycast_binned_kernel(const RayRec* __restrict__ rays, const uint32_t* __restrict__ sorted,
                                                             uint32_t n_sorted, const _Float16* __restrict__ tab0,
                                                             const _Float16* __restrict__ tab1, uint32_t kp0, uint32_t kp1,
                                                             uint32_t run, uint32_t n_blocks, uint32_t nb8,
                                                             uint32_t pre_terrain, uint32_t pre_rocks, float* __restrict__ out) {
    // XCD-aware order: blocks b, b+8, b+16.. run on one XCD (round-robin dispatch) -> give each XCD one
    // contiguous eighth of the sorted rays so a cell's rays meet in one L2.  Speed only, never correctness.
    const uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    if (lb >= n_blocks) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t i = wave * run;
    if (i >= n_sorted) return;
    const uint32_t i_end = min(i + run, n_sorted);
    uint32_t cur_cell = 0xffffffffu, cur_map = 0xffffffffu;
    CellRegs<2> t;                    // per lane: 4 triangles as 2 packed pairs
    t.poison();
    uint64_t vmask[2][2] = {{0, 0}, {0, 0}};
    for (; i < i_end; ++i) {
        const uint32_t gid = __builtin_amdgcn_readfirstlane(sorted[i]);
        const float4* rp = reinterpret_cast<const float4*>(rays + gid);
        const float4 ra = rp[0], rb = rp[1];
        const uint32_t cell = __builtin_amdgcn_readfirstlane(__float_as_uint(ra.w));
        const uint32_t map = __builtin_amdgcn_readfirstlane(__float_as_uint(rb.w)) & 1u;
        if (cell != cur_cell || map != cur_map) {
            if (map != cur_map && cur_map != 0xffffffffu && kp0 != kp1) t.poison();   // lanes past the new map's K
            cur_cell = cell; cur_map = map;
            const uint32_t kp = map ? kp1 : kp0;
            const _Float16* base = (map ? tab1 : tab0) + (size_t)cell * 9u * kp + lane * 4u;
            if (lane * 4u < kp) {
                half4 v[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = *reinterpret_cast<const half4*>(base + (size_t)q * kp);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int t0 = 2 * p, t1 = 2 * p + 1;
                    f2 w[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) w[q] = f2{(float)v[q][t0], (float)v[q][t1]};
                    set_pair(t, p, w);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {                     // real (non-NaN) triangles per pair element, all lanes vote
                vmask[p][0] = __builtin_amdgcn_ballot_w64(t.ax[p].x == t.ax[p].x);
                vmask[p][1] = __builtin_amdgcn_ballot_w64(t.ax[p].y == t.ax[p].y);
            }
        }
        const f2 sx = {ra.x, ra.x}, sy = {ra.y, ra.y}, sz = {ra.z, ra.z};
        const f2 dx = {rb.x, rb.x}, dy = {rb.y, rb.y}, dz = {rb.z, rb.z};
        float best = cast_pairs<2>(t, sx, sy, sz, dx, dy, dz, vmask, map ? pre_rocks : pre_terrain);
        best = wave_min_to_lane63(best);
        if (lane == 63u) out[gid] = best;
    }
}

// ---------------------------------------------------------------------------------------------------
// KNN map builder ("next" row f-3): for every map cell the K triangles whose centroid is nearest in xy
// (tasks/utils/rover_utils.py:52-118, which ranks ALL centroids per cell with a python double loop of topk).
// Here: centroids are bucketed on a uniform grid; one workgroup per map cell gathers buckets ring by ring into
// LDS as 64-bit keys (f32 squared distance bits << 32 | triangle id), bitonic-sorts them and stops as soon as the
// K-th distance is inside the searched radius.  Ranking: exact f32 squared distance, ties by triangle id (the
// reference ranks fp16-rounded distances with an unspecified tie order, so it cannot be matched bit for bit).
// ---------------------------------------------------------------------------------------------------
#define KNN_CAP 8192

// ref = 1: the reference's own arithmetic — numpy float64 centroid of the (float64) open3d vertices (rover_utils.py:68-70),
// then torch.tensor(..., dtype=float16) (:72), which converts double -> float -> half; kept here as the half's float value.
__global__ void __launch_bounds__(256) knn_centroid_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tris,
                                                           uint32_t T, uint32_t V, int ref, float* __restrict__ cx, float* __restrict__ cy) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    uint32_t a = (uint32_t)tris[3ull * t], b = (uint32_t)tris[3ull * t + 1], c = (uint32_t)tris[3ull * t + 2];
    a = a < V ? a : 0u; b = b < V ? b : 0u; c = c < V ? c : 0u;
    if (ref) {
        double x = (((double)verts[3ull * a] + (double)verts[3ull * b]) + (double)verts[3ull * c]) / 3.0;
        double y = (((double)verts[3ull * a + 1] + (double)verts[3ull * b + 1]) + (double)verts[3ull * c + 1]) / 3.0;
        cx[t] = (float)(_Float16)(float)x;
        cy[t] = (float)(_Float16)(float)y;
        return;
    }
    cx[t] = (verts[3ull * a] + verts[3ull * b] + verts[3ull * c]) / 3.0f;             // rover_utils.py:68-70
    cy[t] = (verts[3ull * a + 1] + verts[3ull * b + 1] + verts[3ull * c + 1]) / 3.0f;
}

__device__ __forceinline__ uint32_t knn_bucket(float v, float origin, float inv_g, uint32_t n) {
    float f = (v - origin) * inv_g;
    if (!(f > 0.0f)) return 0u;
    uint32_t b = (uint32_t)f;
    return b < n ? b : n - 1u;
}

// pass 1 (count = 1): histogram of bucket ids; pass 2 (count = 0): fill ids at cursor positions
__global__ void __launch_bounds__(256) knn_bucket_kernel(const float* __restrict__ cx, const float* __restrict__ cy, uint32_t T,
                                                         float ox, float oy, float inv_g, uint32_t nbx, uint32_t nby,
                                                         uint32_t* __restrict__ cursor, uint32_t* __restrict__ items, int count) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    uint32_t b = knn_bucket(cx[t], ox, inv_g, nbx) * nby + knn_bucket(cy[t], oy, inv_g, nby);
    uint32_t pos = atomicAdd(cursor + b, 1u);
    if (!count) items[pos] = t;
}

__global__ void __launch_bounds__(256) knn_select_kernel(const float* __restrict__ cx, const float* __restrict__ cy,
                                                         const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ items,
                                                         float ox, float oy, float g, uint32_t nbx, uint32_t nby, uint32_t X,
                                                         uint32_t Y, float res, uint32_t K, const float* __restrict__ cell_x,
                                                         const float* __restrict__ cell_y, int32_t* __restrict__ out,
                                                         int32_t* __restrict__ overflow) {
    __shared__ unsigned long long keys[KNN_CAP];
    __shared__ uint32_t count;
    const uint32_t cell = blockIdx.x, x = cell / Y, y = cell % Y, tid = threadIdx.x;
    // rover_utils.py:75-81: cell (x, y) sits at (x res, y res).  Reference ranking (cell_x != NULL): the fp16 coordinate tables
    // of its torch.arange(..., dtype=float16), and distances as its fp16 tensors give them (:99-102): fp16(c - p) per axis,
    // norm = sqrt of the f32 sum of squares, rounded to fp16; the key is that fp16 value, ties by triangle id.
    const bool ref = cell_x != nullptr;
    const float px = ref ? cell_x[x] : (float)x * res, py = ref ? cell_y[y] : (float)y * res;
    const float inv_g = 1.0f / g;
    const int bx = (int)knn_bucket(px, ox, inv_g, nbx), by = (int)knn_bucket(py, oy, inv_g, nby);
    const int rmax = (int)max(nbx, nby);
    if (tid == 0) count = 0;
    __syncthreads();
    for (int r = 0; r <= rmax; ++r) {
        // append the buckets at Chebyshev distance r (the square ring), one bucket per thread at a time
        const int side = 2 * r + 1, nring = r == 0 ? 1 : 8 * r;
        for (int q = (int)tid; q < nring; q += 256) {
            int i, j;
            if (r == 0) { i = bx; j = by; }
            else if (q < side) { i = bx - r; j = by - r + q; }                         // left column
            else if (q < 2 * side) { i = bx + r; j = by - r + (q - side); }            // right column
            else if (q < 2 * side + (side - 2)) { i = bx - r + 1 + (q - 2 * side); j = by - r; }     // bottom row
            else { i = bx - r + 1 + (q - 2 * side - (side - 2)); j = by + r; }                     // top row
            if (i < 0 || j < 0 || i >= (int)nbx || j >= (int)nby) continue;
            const uint32_t b = (uint32_t)i * nby + (uint32_t)j, s0 = bucket_start[b], s1 = bucket_start[b + 1];
            if (s1 == s0) continue;
            uint32_t base = atomicAdd(&count, s1 - s0);
            for (uint32_t k = s0; k < s1; ++k, ++base) {
                if (base >= KNN_CAP) break;
                const uint32_t t = items[k];
                float dx = cx[t] - px, dy = cy[t] - py;
                if (ref) { dx = (float)(_Float16)dx; dy = (float)(_Float16)dy; }
                const float d2 = dx * dx + dy * dy;
                const float key = ref ? (float)(_Float16)sqrtf(d2) : d2;        // non-negative floats order like their bits
                keys[base] = ((unsigned long long)__float_as_uint(key) << 32) | t;
            }
        }
        __syncthreads();
        const uint32_t n = count;
        __syncthreads();                 // every wave has read n before any wave appends the next ring to `count`
        if (n > KNN_CAP) { if (tid == 0) *overflow = 1; return; }
        const bool covers_all = (bx - r <= 0) && (by - r <= 0) && (bx + r >= (int)nbx - 1) && (by + r >= (int)nby - 1);
        if (n >= K || covers_all) {
            uint32_t m = 1; while (m < n) m <<= 1;                       // bitonic sort of keys[0..m), padded with +inf keys
            for (uint32_t k = n + tid; k < m; k += 256) keys[k] = ~0ull;
            __syncthreads();
            for (uint32_t len = 2; len <= m; len <<= 1) {
                for (uint32_t stride = len >> 1; stride > 0; stride >>= 1) {
                    for (uint32_t k = tid; k < (m >> 1); k += 256) {
                        const uint32_t lo = ((k / stride) * stride << 1) + (k % stride), hi = lo + stride;
                        const bool up = ((lo & len) == 0);
                        const unsigned long long a = keys[lo], b = keys[hi];
                        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
                    }
                    __syncthreads();
                }
            }
            // every triangle not gathered yet is at least r*g away (its bucket is > r rings out)
            // (reference ranking: an ungathered triangle is >= reach away in the fp16-rounded coordinates and its fp16 distance is
            //  at most a factor 1 - 2^-9.9 below that, so a K-th key strictly below reach (1 - 2^-9) cannot be undercut or tied)
            const float reach = (float)r * g;
            const float kth = n >= K ? __uint_as_float((uint32_t)(keys[K - 1] >> 32)) : __builtin_inff();
            if (covers_all || (ref ? kth < reach * (1.0f - 1.0f / 512.0f) : kth <= reach * reach)) {
                for (uint32_t k = tid; k < K; k += 256) out[(uint64_t)cell * K + k] = k < n ? (int32_t)(uint32_t)keys[k] : 0;
                return;
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// ray cast in the reference's AS-SHIPPED arithmetic (option ray_precision = 2): Camera.dtype = float16, so every
// elementwise ATen op of ray_casting.py:31-59 rounds to fp16.  Same structure as raycast_binned_kernel, the per-pair
// maths in packed fp16 (v_pk_mul_f16 / v_pk_add_f16, one rounding per op, no contraction); the three quotients are
// taken in f32 (IEEE, shared reciprocal) and rounded to fp16, which equals the fp16 quotient (24 >= 2*11 + 2 bits).
// Bit-identical to the oracle's fp16 mode, which the as-shipped golden fixture pins bit for bit.
// ---------------------------------------------------------------------------------------------------
// (the fp16 (ray, triangle) arithmetic — CellRegsH, set_pair_h, cast_pairs_h — lives in rover_raymath.h: one definition for this
// kernel and the culled ray cast's exact phase)

__global__ void __launch_bounds__(256) raycast_binned_h_kernel(const RayRec* __restrict__ rays, const uint32_t* __restrict__ sorted,
                                                               uint32_t n_sorted, const _Float16* __restrict__ tab0,
                                                               const _Float16* __restrict__ tab1, uint32_t kp0, uint32_t kp1,
                                                               uint32_t run, uint32_t n_blocks, uint32_t nb8,
                                                               uint32_t pre_terrain, uint32_t pre_rocks, float* __restrict__ out) {
    const uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    if (lb >= n_blocks) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(lb * 4u + (threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t i = wave * run;
    if (i >= n_sorted) return;
    const uint32_t i_end = min(i + run, n_sorted);
    uint32_t cur_cell = 0xffffffffu, cur_map = 0xffffffffu;
    CellRegsH<2> t;
    t.poison();
    uint64_t vmask[2][2] = {{0, 0}, {0, 0}};
    for (; i < i_end; ++i) {
        const uint32_t gid = __builtin_amdgcn_readfirstlane(sorted[i]);
        const float4* rp = reinterpret_cast<const float4*>(rays + gid);
        const float4 ra = rp[0], rb = rp[1];
        const uint32_t cell = __builtin_amdgcn_readfirstlane(__float_as_uint(ra.w));
        const uint32_t map = __builtin_amdgcn_readfirstlane(__float_as_uint(rb.w)) & 1u;
        if (cell != cur_cell || map != cur_map) {
            if (map != cur_map && cur_map != 0xffffffffu && kp0 != kp1) t.poison();
            cur_cell = cell; cur_map = map;
            const uint32_t kp = map ? kp1 : kp0;
            const _Float16* base = (map ? tab1 : tab0) + (size_t)cell * 9u * kp + lane * 4u;
            if (lane * 4u < kp) {
                half4 v[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = *reinterpret_cast<const half4*>(base + (size_t)q * kp);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int t0 = 2 * p, t1 = 2 * p + 1;
                    h2 vv[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) vv[q] = h2{v[q][t0], v[q][t1]};
                    set_pair_h(t, p, vv);
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                vmask[p][0] = __builtin_amdgcn_ballot_w64(t.ax[p].x == t.ax[p].x);
                vmask[p][1] = __builtin_amdgcn_ballot_w64(t.ax[p].y == t.ax[p].y);
            }
        }
        // the record holds fp16 values widened to f32 (prep_rays_kernel, precision 2): the casts are exact
        const _Float16 hsx = (_Float16)ra.x, hsy = (_Float16)ra.y, hsz = (_Float16)ra.z;
        const _Float16 hdx = (_Float16)rb.x, hdy = (_Float16)rb.y, hdz = (_Float16)rb.z;
        float best = cast_pairs_h<2>(t, h2{hsx, hsx}, h2{hsy, hsy}, h2{hsz, hsz}, h2{hdx, hdx}, h2{hdy, hdy}, h2{hdz, hdz}, vmask,
                                  map ? pre_rocks : pre_terrain);
        best = wave_min_to_lane63(best);
        if (lane == 63u) out[gid] = best;
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers (host)
// ---------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint64_t n, uint32_t bs) { return (uint32_t)((n + bs - 1) / bs); }

hipError_t launch_repack(const int32_t* map_idx, const int32_t* tris, const uint16_t* verts, uint64_t n_cells, uint32_t K,
                         uint32_t K8, uint32_t T, uint32_t V, uint16_t* table, hipStream_t s) {
    uint64_t n = n_cells * K8;
    hipLaunchKernelGGL(repack_knn_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, map_idx, tris, verts, n_cells, K, K8, T, V, table);
    return hipGetLastError();
}

hipError_t launch_prep(const PrepArgs& a, hipStream_t s) {
    // 64 envs x a range of slot groups per block: as many groups as still leave ~1 024 blocks (the pose work is per block; with the
    // fused histogram's LDS above 40 KB — more than 576 buckets — three blocks fit a CU, not four: 2 048 blocks then)
    const uint32_t target = (a.hist && a.hist_buckets > 576u) ? 2048u : 1024u;
    const uint32_t xb = blocks_for(a.E, 64), n_groups = a.R8 / PREP_SLOTS;                // (R8 is a multiple of 8)
    const uint32_t yb = max(1u, min(n_groups, (target + xb - 1u) / xb)), gpb = (n_groups + yb - 1u) / yb;
    hipLaunchKernelGGL(prep_rays_kernel, dim3(xb, (n_groups + gpb - 1u) / gpb), dim3(64 * PREP_SLOTS), a.hist ? a.hist_buckets * 4u : 0u, s, a, gpb);
    return hipGetLastError();
}

hipError_t launch_raycast(const RayRec* rays, uint32_t n_rays, const uint16_t* tab0, const uint16_t* tab1, uint32_t kp0,
                          uint32_t kp1, float* out, hipStream_t s) {
    hipLaunchKernelGGL(raycast_kernel, dim3(blocks_for(n_rays, 8)), dim3(256), 0, s, rays, n_rays,
                       reinterpret_cast<const _Float16*>(tab0), reinterpret_cast<const _Float16*>(tab1), kp0, kp1, out);
    return hipGetLastError();
}

hipError_t launch_scan_exclusive(uint32_t* data, uint32_t n, uint32_t* block_sums, hipStream_t s);

// tile of the sort's first two passes: 4 096 slots per 256-thread block; from 2 M slots 16 384 per 1 024-thread block — a (block, bucket)
// run of the scatter is then ~23 entries instead of ~6 (its partial-line writes were 9 of its 21 us at 65 536 envs) and the count
// table a quarter of the rows; small batches keep the small tile (more blocks than CUs matter more there)
static inline uint32_t bin_tile_threads(uint32_t n_slots, uint32_t low_bits, bool* packed_out) {
    const bool big = n_slots >= (2u << 20);
    const bool packed = (uint64_t)n_slots <= (1ull << (32u - low_bits));          // slot ids leave room for the low bin bits: one dword per entry
    if (packed_out) *packed_out = packed;
    return big ? (packed ? 1024u : 512u) : 256u;                                  // (two-dword entries: 12 bytes of LDS per slot, 8 192 slots)
}

// The sort's first pass — keys per (tile, coarse bucket) — can run inside prep_rays_kernel when the keys of one of its 64-env blocks lie
// in ONE tile (the tile is a multiple of 64 x R8 slots: R8 = 64 with the default 37 + 26 rays); the blocks of a tile add their LDS
// histograms to the tile's row, which therefore has to be zero when the step starts: bucket_sort_kernel, the sort's last pass, clears
// the table again.  One launch less per step (6.1 us at 65 536 envs, 4.7 at 4 096); other ray counts keep bucket_hist_kernel.
bool bin_hist_fused(uint32_t n_slots, uint32_t R8, uint32_t n_bins, uint32_t low_bits, uint32_t* blocks_per_tile) {
    const uint32_t n_buckets = (n_bins + (1u << low_bits) - 1u) >> low_bits;
    const uint32_t tile = bin_tile_threads(n_slots, low_bits, nullptr) * BKT_ITEMS, keys = 64u * R8;      // keys of a prep_rays block's 64 envs
    if (low_bits < 8u || low_bits > 12u || n_buckets > 1024u || keys > tile || tile % keys != 0u) return false;
    *blocks_per_tile = tile / keys;
    return true;
}

// sort the valid ray slots by bin; work = [counts/offsets table | pairs]; returns hipErrorInvalidValue when the bin space is too large
hipError_t launch_bin_rays(const uint32_t* bins, uint32_t n_slots, uint32_t n_valid, uint32_t n_bins, uint32_t low_bits,
                           uint32_t* table, uint2* pairs, uint32_t* block_sums, uint32_t* sorted, bool hist_done, hipStream_t s) {
    const uint32_t n_buckets = (n_bins + (1u << low_bits) - 1u) >> low_bits;
    if (low_bits < 8u || low_bits > 12u || n_buckets > BKT_MAX) return hipErrorInvalidValue;
    const bool big = n_slots >= (2u << 20);
    bool packed;
    const uint32_t nt = bin_tile_threads(n_slots, low_bits, &packed), tile = nt * BKT_ITEMS;
    const uint32_t n_blocks = blocks_for(n_slots, tile);
    uint32_t* bucket_tot = block_sums;                 // [BKT_MAX]
    uint32_t* const zero = hist_done ? table : nullptr;         // counted by prep_rays_kernel: the table has to be zero again after this sort
    const uint32_t n_zero = n_blocks * n_buckets;
    uint32_t* bucket_base = block_sums + BKT_MAX;      // [BKT_MAX + 1]
    const uint32_t scatter_lds = (2u * n_buckets + (packed ? 2u : 3u) * tile + 2u) * 4u, sort_lds = ((1u << low_bits) + BKT_STAGE) * 4u;
#define BKT_LAUNCH(PK, NT)                                                                                                            \
    do {                                                                                                                              \
        if (!hist_done) hipLaunchKernelGGL(bucket_hist_kernel<NT>, dim3(n_blocks), dim3(NT), 0, s, bins, n_slots, low_bits, n_buckets, n_blocks, table); \
        hipLaunchKernelGGL(bucket_rowscan_kernel, dim3(blocks_for(n_buckets, 16)), dim3(64 * RSCAN_WAVES), 0, s, table, n_blocks, n_buckets, bucket_tot); \
        hipLaunchKernelGGL((bucket_scatter_kernel<PK, NT>), dim3(n_blocks), dim3(NT), scatter_lds, s, bins, n_slots, low_bits, n_buckets, n_blocks, \
                           table, bucket_tot, bucket_base, pairs);                                                                   \
        /* buckets of <= 16 384 entries: 512 threads x 32; mean bucket above 8 192 (dense ray sets: the terrain buckets of 65 536 envs x 120 rays hold 22 k): 1 024 x 32 */ \
        if ((uint64_t)n_valid > 8192ull * n_buckets)                                                                                  \
            hipLaunchKernelGGL((bucket_sort_kernel<PK, 1024, 32>), dim3(n_buckets), dim3(1024), sort_lds, s, pairs, bucket_base, low_bits, sorted, zero, n_zero); \
        else                                                                                                                          \
            hipLaunchKernelGGL((bucket_sort_kernel<PK, 512, 32>), dim3(n_buckets), dim3(512), sort_lds, s, pairs, bucket_base, low_bits, sorted, zero, n_zero);     \
    } while (0)
    if (big) {      // more than 64 KB of dynamic LDS: the kernel has to be told (once per size)
        static uint32_t raised_of[64][2] = {};       // per device: the attribute belongs to the function on the current device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        uint32_t* raised = raised_of[dev];
        if (scatter_lds > raised[packed]) {
            const void* f = packed ? reinterpret_cast<const void*>(&bucket_scatter_kernel<true, 1024>) : reinterpret_cast<const void*>(&bucket_scatter_kernel<false, 512>);
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds);
            if (e != hipSuccess) return e;
            raised[packed] = scatter_lds;
        }
    }
    if (packed) { if (big) BKT_LAUNCH(true, 1024); else BKT_LAUNCH(true, 256); }
    else { if (big) BKT_LAUNCH(false, 512); else BKT_LAUNCH(false, 256); }
#undef BKT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_raycast_binned(const RayRec* rays, const uint32_t* sorted, uint32_t n_sorted, const uint16_t* tab0,
                                 const uint16_t* tab1, uint32_t kp0, uint32_t kp1, uint32_t run, bool fp16_math, uint32_t early_out,
                                 float* out, hipStream_t s) {
    const uint32_t n_waves = blocks_for(n_sorted, run);
    const uint32_t n_blocks = blocks_for(n_waves, 4);
    const uint32_t nb8 = blocks_for(n_blocks, 8);
    const _Float16* t0 = reinterpret_cast<const _Float16*>(tab0);
    const _Float16* t1 = reinterpret_cast<const _Float16*>(tab1);
    // pair 1 = the far half of a cell's list: tested on both maps; pair 0 (near half) only on the rocks map, where most
    // rays are far from any rock triangle
    // (A/B in one process, 65 536 envs, f32 kernel: none 1.424 ms, far pair only 1.228, every pair 1.249, this choice 1.169)
    const uint32_t pre_terrain = early_out ? 2u : 0u, pre_rocks = early_out ? 3u : 0u;
    if (fp16_math)
        hipLaunchKernelGGL(raycast_binned_h_kernel, dim3(nb8 * 8u), dim3(256), 0, s, rays, sorted, n_sorted, t0, t1, kp0, kp1, run,
                           n_blocks, nb8, pre_terrain, pre_rocks, out);
    else
        hipLaunchKernelGGL(raycast_binned_kernel, dim3(nb8 * 8u), dim3(256), 0, s, rays, sorted, n_sorted, t0, t1, kp0, kp1, run,
                           n_blocks, nb8, pre_terrain, pre_rocks, out);
    return hipGetLastError();
}

hipError_t launch_knn_centroids(const float* verts, const int32_t* tris, uint32_t T, uint32_t V, int ref, float* cx, float* cy,
                                hipStream_t s) {
    hipLaunchKernelGGL(knn_centroid_kernel, dim3(blocks_for(T, 256)), dim3(256), 0, s, verts, tris, T, V, ref, cx, cy);
    return hipGetLastError();
}

hipError_t launch_knn_bucket(const float* cx, const float* cy, uint32_t T, float ox, float oy, float inv_g, uint32_t nbx, uint32_t nby,
                             uint32_t* cursor, uint32_t* items, int count, hipStream_t s) {
    hipLaunchKernelGGL(knn_bucket_kernel, dim3(blocks_for(T, 256)), dim3(256), 0, s, cx, cy, T, ox, oy, inv_g, nbx, nby, cursor, items,
                       count);
    return hipGetLastError();
}

hipError_t launch_scan_exclusive(uint32_t* data, uint32_t n, uint32_t* block_sums, hipStream_t s) {
    const uint32_t nb = blocks_for(n, SCAN_TILE);
    if (nb > 1024u * SCAN_ITEMS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, data, n, block_sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, s, block_sums, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, data, n, block_sums);
    return hipGetLastError();
}

hipError_t launch_knn_select(const float* cx, const float* cy, const uint32_t* bucket_start, const uint32_t* items, float ox, float oy,
                             float g, uint32_t nbx, uint32_t nby, uint32_t X, uint32_t Y, float res, uint32_t K, const float* cell_x,
                             const float* cell_y, int32_t* out, int32_t* overflow, hipStream_t s) {
    hipLaunchKernelGGL(knn_select_kernel, dim3(X * Y), dim3(256), 0, s, cx, cy, bucket_start, items, ox, oy, g, nbx, nby, X, Y, res, K,
                       cell_x, cell_y, out, overflow);
    return hipGetLastError();
}

hipError_t launch_assemble_obs(const ObsArgs& a_in, hipStream_t s) {
    ObsArgs a = a_in;
    a.w_div = make_fastdiv(a.W);
    hipLaunchKernelGGL(assemble_obs_kernel, dim3(blocks_for((uint64_t)a.E * a.W, 256 * OBS_ILP)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_export_dist(const float* dist, const RayRec* rays, uint32_t E, uint32_t R8, uint32_t P, int precision, float* ray_dist,
                              float* wheel, float* body, float* ray_src, float* hit_pt, hipStream_t s) {
    hipLaunchKernelGGL(export_dist_kernel, dim3(blocks_for((uint64_t)E * (26u + P), 256)), dim3(256), 0, s, dist, rays, E, R8, P, precision,
                       ray_dist, wheel, body, ray_src, hit_pt);
    return hipGetLastError();
}

hipError_t launch_export_rays(const RayRec* rays, const float* dist, uint32_t E, uint32_t R8, uint32_t P, float* src, float* dir, int32_t* cell,
                              float* out_dist, hipStream_t s) {
    hipLaunchKernelGGL(export_rays_kernel, dim3(blocks_for((uint64_t)E * (26u + P), 256)), dim3(256), 0, s, rays, dist, E, R8, 26u + P, src, dir,
                       cell, out_dist);
    return hipGetLastError();
}

hipError_t launch_import_rays(const float* src, const float* dir, uint32_t E, uint32_t R8, uint32_t P, const KnnDev& terrain, const KnnDev& rocks,
                              uint32_t rocks_bin_offset, int precision, int cell_rcp, RayRec* rays, uint32_t* bin_out, hipStream_t s,
                              uint32_t* not_unit) {
    hipLaunchKernelGGL(import_rays_kernel, dim3(blocks_for((uint64_t)E * R8, 256)), dim3(256), 0, s, src, dir, E, R8, 26u + P, terrain, rocks,
                       rocks_bin_offset, precision, cell_rcp, rays, bin_out, not_unit);
    return hipGetLastError();
}

hipError_t launch_obs_metrics(const ObsArgs& o_in, const MetricsArgs& m, hipStream_t s) {
    ObsArgs o = o_in;
    o.w_div = make_fastdiv(o.W);
    const uint32_t n_met = blocks_for(m.E, 256), n_obs = blocks_for((uint64_t)o.E * o.W, 256 * OBS_ILP);
    hipLaunchKernelGGL(obs_metrics_kernel, dim3(n_met + n_obs), dim3(256), 0, s, o, m, n_met);
    return hipGetLastError();
}

hipError_t launch_metrics_done(const MetricsArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(metrics_done_kernel, dim3(blocks_for(a.E, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_compact(const int64_t* reset, uint32_t n, int64_t offset, uint32_t* block_cnt, bool counted, int64_t* ids,
                          int32_t* count, hipStream_t s) {
    const uint32_t nb = blocks_for(n, 256);
    if (!counted) hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, s, reset, n, block_cnt);
    hipLaunchKernelGGL(compact_write_kernel, dim3(nb), dim3(256), 0, s, reset, n, offset, block_cnt, ids, count);
    return hipGetLastError();
}

hipError_t launch_quat_to_euler(const float* q, float* eul, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(quat_to_euler_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, q, eul, n);
    return hipGetLastError();
}

hipError_t launch_clearance(const float* info7, uint32_t S, const float* xy, uint32_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(clearance_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, info7, S, xy, n, out);
    return hipGetLastError();
}

hipError_t launch_shift_spawns(const StoneGridDev& g, const float* info7, float* pos3, uint32_t n, int32_t max_iter, hipStream_t s) {
    hipLaunchKernelGGL(shift_spawns_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, g, info7, pos3, n, max_iter);
    return hipGetLastError();
}

hipError_t launch_sample_height(const HeightDev& h, const float* xy, uint32_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sample_height_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, h, xy, n, out);
    return hipGetLastError();
}

hipError_t launch_generate_goals(const GoalArgs& a, uint32_t n_max, hipStream_t s) {
    hipLaunchKernelGGL(goals_draw_kernel, dim3(blocks_for((uint64_t)n_max * GOAL_LANES, 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(goals_env0_kernel, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_reset_envs(const ResetArgs& a, uint32_t n_max, hipStream_t s) {
    hipLaunchKernelGGL(reset_envs_kernel, dim3(blocks_for(n_max, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pre_physics(const PrePhysicsArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(pre_physics_kernel, dim3(blocks_for(a.E, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ackermann(const float* lin, const float* ang, uint32_t n, float* steer, float* vel, hipStream_t s) {
    hipLaunchKernelGGL(ackermann_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, lin, ang, n, steer, vel);
    return hipGetLastError();
}

}  // namespace rover
