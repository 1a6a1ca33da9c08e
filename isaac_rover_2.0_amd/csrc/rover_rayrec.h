// rover_rayrec.h — the origin of a ray from its rover's pose, shared by prep_rays_kernel (rover_kernels.hip: it needs the origin for the
// ray's map cell, the key of the bucket sort) and the culled ray cast (rover_cull.hip: a wave rebuilds the records of its run of sorted
// rays from the per-env tables instead of reading 32-byte records prep_rays_kernel would have to write for every ray slot).  One
// definition, compiled with the same flags (-ffp-contract=off), so that both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rover {

struct Trig6 { float sx, cx, sy, cy, sz, cz; };      // sin / cos of -roll, -pitch, -yaw

// rock_detect.py:160-246: the four rays of a wheel (wheel-local origins; [4] = the shared direction point), the wheel and suspension
// joint offsets, the two body rays
static __constant__ float c_wheel_ray[5][3] = {{0.215 / 2, 0.130 / 2, 0.1}, {0.215 / 2, -0.130 / 2, 0.1},
                                               {-0.215 / 2, 0.130 / 2, 0.1}, {-0.215 / 2, -0.130 / 2, 0.1}, {0, 0, -1}};
static __constant__ float c_wp0[6][3] = {{0.286, 0.385, -0.197}, {0.286, -0.385, -0.197}, {-0.146, 0.447, -0.197},
                                         {-0.146, -0.447, -0.197}, {-0.440, 0.385, -0.197}, {-0.440, -0.385, -0.197}};
static __constant__ float c_wp1[6][3] = {{0.153, 0, 0.03}, {0.153, 0, 0.03}, {0.153, 0, 0.03}, {0.153, -0.0, 0.03},
                                         {0, 0, 0.03}, {0, 0, 0.03}};
static __constant__ float c_body_pt[2][3] = {{0.340, 0, -0.01}, {-0.485, 0, -0.01}};

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

// The pose in float64 for the heightmap rays: camera.py:165-212 works in the distribution tensor's float64.
#define ROVER_POSE_F64(t, px, py, pz)                                                                                                     \
    const double dsx = (double)(t).sx, dcx = (double)(t).cx, dsy = (double)(t).sy, dcy = (double)(t).cy, dsz = (double)(t).sz,            \
                 dcz = (double)(t).cz;                                                                                                     \
    const double X = (double)(px), Y = (double)(py), Z = (double)(pz)

// Origin of ray slot `slot` of a rover (0..23 wheel rays, 24..25 body rays, 26.. heightmap rays), before the fp16 rounding of the
// as-shipped modes.  d0 / d1 / d2: (sin, cos) of -steer, susX, susY of the slot's wheel (slots < 24); (x, y, z): the slot's point of
// the heightmap distribution (slots >= 26).
__device__ __forceinline__ void ray_origin(uint32_t slot, const Trig6& t, float px, float py, float pz, float2 d0, float2 d1, float2 d2,
                                           double x, double y, double z, float& sx, float& sy, float& sz) {
    if (slot < 24u) {                   // rock_detect.py:160-319
        const uint32_t wh = slot >> 2, r = slot & 3u;
        wheel_chain(c_wheel_ray[r][0], c_wheel_ray[r][1], c_wheel_ray[r][2], c_wp0[wh], c_wp1[wh], d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, t, px,
                    py, pz, sx, sy, sz);
    } else if (slot < 26u) {            // rock_detect.py:321-371
        const uint32_t r = slot - 24u;
        body_xf(c_body_pt[r][0], c_body_pt[r][1], c_body_pt[r][2], t, px, py, pz, sx, sy, sz);
    } else {                            // camera.py:165-212, float64 like the distribution tensor
        ROVER_POSE_F64(t, px, py, pz);
        const double A = y * dcx + z * dsx, C = z * dcx - y * dsx, B = x * dcy - dsy * C;
        sx = (float)(X + dsz * A + dcz * B);
        sy = (float)(Y + dcz * A - dsz * B);
        sz = (float)(Z + x * dsy + dcy * C);
    }
}

// kind of a slot's direction record (a wheel's four rays share one, the two body rays one, all heightmap rays one)
__device__ __forceinline__ uint32_t ray_kind(uint32_t slot) { return slot < 24u ? slot >> 2 : (slot < 26u ? 6u : 7u); }

}  // namespace rover
