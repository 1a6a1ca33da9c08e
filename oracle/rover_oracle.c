/*
 * TEST INFRASTRUCTURE — CPU restatement of the reference's rover env.step() hot path.
 *
 * This file is the parity oracle for the HIP kernels.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * It restates, operation by operation in the reference's own arithmetic types and evaluation
 * order, what the reference's PyTorch code computes in its "fp32 mode" (Camera.dtype =
 * Rock_Detection.dtype = float32 on fp16-stored vertices; thresholds stay fp16-rounded because
 * ray_casting.py:3 defaults dtype=float16).  Build with -ffp-contract=off: every +,-,*,/ is one
 * IEEE-754 rounding, like one ATen elementwise kernel.  Pinned by tests/golden/*.npz, which were
 * captured from the reference itself (oracle/gen_golden.py).
 *
 * Reference file:line cited per function; paths relative to /root/reference/omniisaacgymenvs/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ---- fp16 storage -> f32 (exact) ------------------------------------------------------- */
static inline float h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 0x3ffu; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

/* f32 -> fp16 -> f32, round to nearest even (what Tensor.type(torch.float16) does, camera.py:212) */
static inline float round_h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = x & 0x80000000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return f;                                   /* inf / nan */
    if (ax >= 0x477ff000u) { uint32_t inf = sign | 0x7f800000u; float r; memcpy(&r, &inf, 4); return r; }   /* >= 65520 -> inf */
    if (ax < 0x38800000u) {                                            /* < 2^-14: fp16 subnormal, spacing 2^-24 */
        float a = fabsf(f), q = rintf(a * 16777216.0f) / 16777216.0f;
        return sign ? -q : q;
    }
    uint32_t lsb = (ax >> 13) & 1u;
    ax += 0x0fffu + lsb;                                               /* round the 13 dropped bits, ties to even */
    ax &= ~0x1fffu;
    x = sign | ax;
    float r; memcpy(&r, &x, 4); return r;
}

/* constants of ray_casting.py:24-27 after fp16 rounding */
#define RAY_NEG_EPS  (-0.0999755859375f)
#define RAY_ONE_EPS  (1.099609375f)
#define RAY_MISS     (11.0f)

typedef struct {
    int32_t X, Y, K;            /* map_indices is [X][Y][K] (camera.py:156-158 after the swaps) */
    const int32_t *map_idx;
    const int32_t *tris;        /* [T][3] */
    const uint16_t *verts;      /* [V][3] fp16 bits */
    float cell;                 /* horizontal = 0.1 (camera.py:48) */
    float shift_x, shift_y;     /* shift[0:2] (camera.py:239) */
} oracle_knn_map;

/* tasks/utils/math/tensor_quat_to_euler.py:6-31 */
ORACLE_API void oracle_quat_to_euler(int n, const float *q, float *eul) {
    const float half_pi = 3.1415927410125732f / 2.0f;      /* (ones*torch.pi)/2, :5,:24 */
    for (int i = 0; i < n; ++i) {
        float w = q[4*i], x = q[4*i+1], y = q[4*i+2], z = q[4*i+3];
        float sinr = 2.0f * (w * x + y * z);
        float cosr = 1.0f - (2.0f * (x * x + y * y));
        eul[3*i] = atan2f(sinr, cosr);
        float sinp = 2.0f * (w * y - z * x);
        float t = sinp - 1.0f;                              /* sign(sinp-1) >= 0  <=>  sinp-1 >= 0 (NaN -> false) */
        eul[3*i+1] = (t >= 0.0f) ? copysignf(half_pi, sinp) : asinf(sinp);
        float siny = 2.0f * (w * z + x * y);
        float cosy = 1.0f - (2.0f * (y * y + z * z));
        eul[3*i+2] = atan2f(siny, cosy);
    }
}

/* rover.py:279-283 */
static float heading_diff_f(const float *target, const float *pos, float yaw) {
    float dx = cosf(yaw), dy = sinf(yaw);
    float tx = target[0] - pos[0], ty = target[1] - pos[1];
    return -atan2f(tx * dy - ty * dx, tx * dx + ty * dy);
}

/* camera.py:165-212 — f64 because the distribution tensor is float64; sin/cos are f32 values */
static void depth_transform(const float *pos, const float *eul, int P, const double *dist,
                            float *src /*[P][3]*/, float *dir /*[3]*/) {
    double sx = (double)sinf(-eul[0]), cx = (double)cosf(-eul[0]);
    double sy = (double)sinf(-eul[1]), cy = (double)cosf(-eul[1]);
    double sz = (double)sinf(-eul[2]), cz = (double)cosf(-eul[2]);
    double X = (double)pos[0], Y = (double)pos[1], Z = (double)pos[2];
    for (int p = 0; p <= P; ++p) {
        double x, y, z;
        if (p < P) { x = dist[3*p]; y = dist[3*p+1]; z = dist[3*p+2]; }
        else { x = 0.0; y = 0.0; z = -1.0; }                 /* :179-181 */
        double A = y * cx + z * sx;
        double C = z * cx - y * sx;
        double B = x * cy - sy * C;
        double xp = X + sz * A + cz * B;                     /* :197 */
        double yp = Y + cz * A - sz * B;                     /* :198 */
        double zp = Z + x * sy + cy * C;                     /* :199 */
        if (p < P) { src[3*p] = (float)xp; src[3*p+1] = (float)yp; src[3*p+2] = (float)zp; }
        else { dir[0] = (float)(xp - X); dir[1] = (float)(yp - Y); dir[2] = (float)(zp - Z); }  /* :202-204 */
    }
}

/* rock_detect.py:160-319 — all f32.  out: src [24][3], dir [6][3] (one per wheel) */
static void wheel_rays(const float *pos, const float *eul, const float *j, float *src, float *dir) {
    static const float ray[5][3] = {{0.215/2, 0.130/2, 0.1}, {0.215/2, -0.130/2, 0.1}, {-0.215/2, 0.130/2, 0.1},
                                    {-0.215/2, -0.130/2, 0.1}, {0, 0, -1}};                       /* :193-197 */
    static const float wp0[6][3] = {{0.286, 0.385, -0.197}, {0.286, -0.385, -0.197}, {-0.146, 0.447, -0.197},
                                    {-0.146, -0.447, -0.197}, {-0.440, 0.385, -0.197}, {-0.440, -0.385, -0.197}}; /* :201-206 */
    static const float wp1[6][3] = {{0.153, 0, 0.03}, {0.153, 0, 0.03}, {0.153, 0, 0.03}, {0.153, -0.0, 0.03},
                                    {0, 0, 0.03}, {0, 0, 0.03}};                                  /* :210-215 */
    const float steer[6] = {j[4], j[6], 0.0f, 0.0f, -j[7], j[8]};                                 /* :248 */
    const float susY[6] = {-j[0], j[1], -j[0], j[1], 0.0f, 0.0f};                                 /* :263 */
    const float susX[6] = {0.0f, 0.0f, 0.0f, 0.0f, -j[2], -j[2]};                                 /* :264 */
    float sxr = sinf(-eul[0]), cxr = cosf(-eul[0]);
    float syr = sinf(-eul[1]), cyr = cosf(-eul[1]);
    float szr = sinf(-eul[2]), czr = cosf(-eul[2]);
    for (int w = 0; w < 6; ++w) {
        float sst = sinf(-steer[w]), cst = cosf(-steer[w]);
        float ssx = sinf(susX[w]), csx = cosf(susX[w]);
        float ssy = sinf(susY[w]), csy = cosf(susY[w]);
        for (int r = 0; r < 5; ++r) {
            int isdir = (r == 4);
            float x = ray[r][0], y = ray[r][1], z = ray[r][2];
            float t0x = isdir ? 0.0f : wp0[w][0], t0y = isdir ? 0.0f : wp0[w][1], t0z = isdir ? 0.0f : wp0[w][2];
            float t1x = isdir ? 0.0f : wp1[w][0], t1y = isdir ? 0.0f : wp1[w][1], t1z = isdir ? 0.0f : wp1[w][2];
            float x1 = t0x + x * cst + y * sst;               /* :256 */
            float y1 = t0y + y * cst - x * sst;               /* :257 */
            float z1 = t0z + z;                               /* :258 */
            float c1 = z1 * csx - y1 * ssx;
            float x2 = t1x + x1 * csy - ssy * c1;             /* :275 */
            float y2 = t1y + y1 * csx + z1 * ssx;             /* :276 */
            float z2 = t1z + x1 * ssy + csy * c1;             /* :277 */
            float A = y2 * cxr + z2 * sxr;
            float C = z2 * cxr - y2 * sxr;
            float B = x2 * cyr - syr * C;
            float px = isdir ? 0.0f : pos[0], py = isdir ? 0.0f : pos[1], pz = isdir ? 0.0f : pos[2];
            float xp = px + szr * A + czr * B;                /* :305 */
            float yp = py + czr * A - szr * B;                /* :306 */
            float zp = pz + x2 * syr + cyr * C;               /* :307 */
            float *o = isdir ? (dir + 3*w) : (src + 3*(4*w + r));
            o[0] = xp; o[1] = yp; o[2] = zp;
        }
    }
}

/* rock_detect.py:321-371 — f32; src [2][3], dir [3] */
static void body_rays(const float *pos, const float *eul, float *src, float *dir) {
    static const float pt[3][3] = {{0.340, 0, -0.01}, {-0.485, 0, -0.01}, {0, 1, 0}};             /* :326,:338-340 */
    float sx = sinf(-eul[0]), cx = cosf(-eul[0]);
    float sy = sinf(-eul[1]), cy = cosf(-eul[1]);
    float sz = sinf(-eul[2]), cz = cosf(-eul[2]);
    for (int p = 0; p < 3; ++p) {
        float x = pt[p][0], y = pt[p][1], z = pt[p][2];
        float A = y * cx + z * sx;
        float C = z * cx - y * sx;
        float B = x * cy - sy * C;
        float xp = pos[0] + sz * A + cz * B;                  /* :356 */
        float yp = pos[1] + cz * A - sz * B;                  /* :357 */
        float zp = pos[2] + x * sy + cy * C;                  /* :358 */
        if (p < 2) { src[3*p] = xp; src[3*p+1] = yp; src[3*p+2] = zp; }
        else { dir[0] = xp - pos[0]; dir[1] = yp - pos[1]; dir[2] = zp - pos[2]; }                /* :361-363 */
    }
}

/* camera.py:233-264 / rock_detect.py:373-401: clamp bound is dim-0 size for BOTH axes */
/* `x / horizontal_scale` with a Python-float divisor: ATen's CPU kernel divides (mode 0, what the golden vectors
 * captured on CPU pin); ATen's CUDA kernel multiplies by opmath_t(1.0) / scalar, the reciprocal rounded to f32
 * (BinaryDivTrueKernel.cu, "compute a * reciprocal(b)": mode 1).  1.0f/0.1f == 10.0f and 1.0f/0.025f == 40.0f exactly, so
 * the two differ only where (v - shift) / 0.1f and (v - shift) * 10.0f round to different floats next to a .5 tie. */
static int g_cell_rcp = 0;
ORACLE_API void oracle_set_cell_index_mode(int rcp) { g_cell_rcp = rcp; }
static inline int64_t cell_index(float v, float shift, float cell, int32_t dim0) {
    float s = g_cell_rcp ? (v - shift) * (1.0f / cell) : (v - shift) / cell;
    float hi = (float)(dim0 - 1);
    s = (s < 0.0f) ? 0.0f : s;                                /* clamp(min=0, max=X-1): NaN passes through */
    s = (s > hi) ? hi : s;
    s = rintf(s);                                             /* torch.round = half-to-even */
    if (!(s == s)) return 0;                                  /* reference would raise; oracle maps NaN to cell 0 */
    return (int64_t)s;
}

/* ray_casting.py:3-66 for one ray against the K triangles of its cell, then min over K
 * (camera.py:116-117).  d_in is the un-normalised direction. */
static float ray_min_core(const oracle_knn_map *m, const float *s, const float *d);
static float ray_min_distance(const oracle_knn_map *m, const float *s, const float *d_in) {
    /* F.normalize (ray_casting.py:31): x / max(||x||_2, 1e-12), then negated */
    float nrm = sqrtf(d_in[0] * d_in[0] + d_in[1] * d_in[1] + d_in[2] * d_in[2]);
    if (nrm < 1e-12f) nrm = 1e-12f;
    float d[3] = {-(d_in[0] / nrm), -(d_in[1] / nrm), -(d_in[2] / nrm)};
    return ray_min_core(m, s, d);
}
/* ray_casting.py:34-59 + the min over K for a ray whose d = -normalize(direction) is given */
static float ray_min_core(const oracle_knn_map *m, const float *s, const float *d) {
    int64_t ix = cell_index(s[0], m->shift_x, m->cell, m->X);
    int64_t iy = cell_index(s[1], m->shift_y, m->cell, m->X);
    if (iy > m->Y - 1) iy = m->Y - 1;                          /* memory safety only (reference indexes out of range) */
    const int32_t *ids = m->map_idx + ((size_t)ix * m->Y + (size_t)iy) * m->K;
    float best = INFINITY;
    for (int t = 0; t < m->K; ++t) {
        const int32_t *tv = m->tris + 3 * (size_t)ids[t];
        float v[3][3];
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) v[a][c] = h2f(m->verts[3 * (size_t)tv[a] + c]);
        float a[3] = {v[2][0], v[2][1], v[2][2]};                                       /* :34 */
        float b[3] = {v[1][0] - a[0], v[1][1] - a[1], v[1][2] - a[2]};                  /* :35 */
        float c[3] = {v[0][0] - a[0], v[0][1] - a[1], v[0][2] - a[2]};                  /* :36 */
        float g[3] = {s[0] - a[0], s[1] - a[1], s[2] - a[2]};                           /* :37 */
        float bc[3] = {b[1]*c[2] - b[2]*c[1], b[2]*c[0] - b[0]*c[2], b[0]*c[1] - b[1]*c[0]};
        float det = bc[0]*d[0] + bc[1]*d[1] + bc[2]*d[2];                               /* :40-41 */
        float gc[3] = {g[1]*c[2] - g[2]*c[1], g[2]*c[0] - g[0]*c[2], g[0]*c[1] - g[1]*c[0]};
        float n = (gc[0]*d[0] + gc[1]*d[1] + gc[2]*d[2]) / det;                         /* :44-45 */
        float bg[3] = {b[1]*g[2] - b[2]*g[1], b[2]*g[0] - b[0]*g[2], b[0]*g[1] - b[1]*g[0]};
        float mm = (bg[0]*d[0] + bg[1]*d[1] + bg[2]*d[2]) / det;                        /* :49-50 */
        float k = (bc[0]*g[0] + bc[1]*g[1] + bc[2]*g[2]) / det;                         /* :54-55 */
        /* :46,:51,:56 compare det with -0.1 / 1.1 and never change a value that :59 would accept */
        if (det == RAY_NEG_EPS) n = RAY_MISS;
        if (det == RAY_ONE_EPS) { mm = RAY_MISS; k = RAY_MISS; }
        float r = ((n >= RAY_NEG_EPS) && (mm >= RAY_NEG_EPS) && (n + mm <= RAY_ONE_EPS)) ? k : RAY_MISS;  /* :59 */
        if (r < best || r != r) best = r;                                               /* torch.min propagates NaN */
        if (best != best) break;
    }
    return best;
}

/* ray_casting.py:3-66 as ATen evaluates it on float16 tensors: every elementwise op rounds its result to fp16
 * (computed in f32, so the double rounding is innocuous), `cross` is mul/mul/sub, the dot products are summed left to
 * right, F.normalize takes the norm with f32 accumulation and rounds it to fp16.  Pinned bit for bit by the as-shipped
 * golden fixture. */
#define RH(x) round_h(x)
static void cross_h(const float *u, const float *v, float *o) {
    o[0] = RH(RH(u[1]*v[2]) - RH(u[2]*v[1]));
    o[1] = RH(RH(u[2]*v[0]) - RH(u[0]*v[2]));
    o[2] = RH(RH(u[0]*v[1]) - RH(u[1]*v[0]));
}
static float dot_h(const float *u, const float *v) { return RH(RH(RH(u[0]*v[0]) + RH(u[1]*v[1])) + RH(u[2]*v[2])); }

static float ray_min_core_h(const oracle_knn_map *m, const float *s, const float *d);
static float ray_min_distance_h(const oracle_knn_map *m, const float *s, const float *d_in) {
    float nrm = RH(sqrtf(d_in[0] * d_in[0] + d_in[1] * d_in[1] + d_in[2] * d_in[2]));     /* eps 1e-12 is 0 in fp16 */
    float d[3] = {-RH(d_in[0] / nrm), -RH(d_in[1] / nrm), -RH(d_in[2] / nrm)};
    return ray_min_core_h(m, s, d);
}
static float ray_min_core_h(const oracle_knn_map *m, const float *s, const float *d) {
    int64_t ix = cell_index(s[0], m->shift_x, m->cell, m->X);
    int64_t iy = cell_index(s[1], m->shift_y, m->cell, m->X);
    if (iy > m->Y - 1) iy = m->Y - 1;
    const int32_t *ids = m->map_idx + ((size_t)ix * m->Y + (size_t)iy) * m->K;
    float best = INFINITY;
    for (int t = 0; t < m->K; ++t) {
        const int32_t *tv = m->tris + 3 * (size_t)ids[t];
        float v[3][3];
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) v[a][c] = h2f(m->verts[3 * (size_t)tv[a] + c]);
        float a[3] = {v[2][0], v[2][1], v[2][2]}, b[3], c[3], g[3], bc[3], gc[3], bg[3];
        for (int q = 0; q < 3; ++q) { b[q] = RH(v[1][q] - a[q]); c[q] = RH(v[0][q] - a[q]); g[q] = RH(s[q] - a[q]); }
        cross_h(b, c, bc);
        float det = dot_h(bc, d);
        cross_h(g, c, gc);
        float n = RH(dot_h(gc, d) / det);
        cross_h(b, g, bg);
        float mm = RH(dot_h(bg, d) / det);
        float k = RH(dot_h(bc, g) / det);
        if (det == RAY_NEG_EPS) n = RAY_MISS;
        if (det == RAY_ONE_EPS) { mm = RAY_MISS; k = RAY_MISS; }
        float r = ((n >= RAY_NEG_EPS) && (mm >= RAY_NEG_EPS) && (RH(n + mm) <= RAY_ONE_EPS)) ? k : RAY_MISS;
        if (r < best || r != r) best = r;
        if (best != best) break;
    }
    return best;
}

/* ---- one full post_physics_step (rl_task.py:250-257 order) ----------------------------- */
typedef struct {
    int32_t num_envs;           /* local envs in this call */
    int32_t num_envs_global;    /* self.num_envs in rover.py:517 */
    int32_t P, Ns, Nd;          /* rays, sparse count, dense count */
    int32_t curriculum_level;   /* rover.py:292,514,645 */
    int32_t max_episode_length; /* rover.py:119 */
    int32_t precision;          /* 0: the reference's fp32 mode (parity target).
                                   1: ray origins / directions rounded to fp16 (camera.py:55,212; rock_detect.py:319,371),
                                      f32 ray maths.
                                   2: the reference AS SHIPPED: (1) + every operation of ray_casting.py:31-59 rounded to
                                      fp16 like ATen's Half kernels do, fp16 collision thresholds (rover.py:667-668) */
    float pos_reward, heading_contraint_reward, motion_contraint_reward, goal_angle_reward,
          boogie_contraint_reward;                      /* cfg/task/Rover.yaml:37-46 */
} oracle_cfg;

typedef struct {
    const float *pos, *quat, *joints, *target;   /* [E,3] [E,4] [E,13] [E,3] */
    const float *lin_hist, *ang_hist;            /* [E,3] newest first (Memory, rover.py:60-77) */
    const float *euler_pre;                      /* [E,3] self.rover_rot of pre_physics_step (rover.py:343) */
    int64_t *progress;                           /* [E] in/out: += 1 (rl_task.py:250) */
    const double *distribution;                  /* [P,3] float64 */
    const int64_t *sparse_idx, *dense_idx;
} oracle_in;

typedef struct {
    float *euler, *heading;                      /* [E,3] [E] */
    float *ray_src, *ray_dist;                   /* [E,P,3] [E,P] */
    float *wheel_dist, *body_dist;               /* [E,24] [E,2] */
    int64_t *rock_collision;                     /* [E] */
    float *obs;                                  /* [E, 4+Ns+Nd] */
    float *rew;                                  /* [E] */
    int64_t *reset;                              /* [E] */
    float *ex_pos_reward; int64_t *ex_collision; float *ex_upright, *ex_heading, *ex_motion,
          *ex_goal_angle, *ex_lin, *ex_ang;      /* rover.py:524-531 */
} oracle_out;

ORACLE_API void oracle_step(const oracle_cfg *cfg, const oracle_knn_map *terrain, const oracle_knn_map *rocks,
                            const oracle_in *in, const oracle_out *out) {
    const int E = cfg->num_envs, P = cfg->P, Ns = cfg->Ns, Nd = cfg->Nd, W = 4 + Ns + Nd;
    oracle_quat_to_euler(E, in->quat, out->euler);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int e = 0; e < E; ++e) {
        const float *pos = in->pos + 3*e, *eul = out->euler + 3*e, *tgt = in->target + 3*e, *jn = in->joints + 13*e;
        in->progress[e] += 1;                                                   /* rl_task.py:250 */
        /* ---- get_observations, rover.py:272-336 ---- */
        float hd = heading_diff_f(tgt, pos, eul[2]);
        out->heading[e] = hd;
        float *src = out->ray_src + (size_t)3*P*e, dir[3];
        depth_transform(pos, eul, P, in->distribution, src, dir);
        if (cfg->precision) { for (int q = 0; q < 3*P; ++q) src[q] = round_h(src[q]); for (int q = 0; q < 3; ++q) dir[q] = round_h(dir[q]); }
        float *rd = out->ray_dist + (size_t)P*e;
        const int half = cfg->precision == 2;
        for (int p = 0; p < P; ++p) rd[p] = half ? ray_min_distance_h(terrain, src + 3*p, dir) : ray_min_distance(terrain, src + 3*p, dir);
        float wsrc[24*3], wdir[6*3], bsrc[2*3], bdir[3];
        wheel_rays(pos, eul, jn, wsrc, wdir);
        body_rays(pos, eul, bsrc, bdir);
        if (cfg->precision) {
            for (int q = 0; q < 72; ++q) wsrc[q] = round_h(wsrc[q]);
            for (int q = 0; q < 18; ++q) wdir[q] = round_h(wdir[q]);
            for (int q = 0; q < 6; ++q) bsrc[q] = round_h(bsrc[q]);
            for (int q = 0; q < 3; ++q) bdir[q] = round_h(bdir[q]);
        }
        float *wd = out->wheel_dist + 24*e, *bd = out->body_dist + 2*e;
        for (int r = 0; r < 24; ++r) wd[r] = half ? ray_min_distance_h(rocks, wsrc + 3*r, wdir + 3*(r/4)) : ray_min_distance(rocks, wsrc + 3*r, wdir + 3*(r/4));
        for (int r = 0; r < 2; ++r) bd[r] = half ? ray_min_distance_h(rocks, bsrc + 3*r, bdir) : ray_min_distance(rocks, bsrc + 3*r, bdir);
        int64_t coll = 0;
        if (cfg->curriculum_level >= 2) {                                        /* rover.py:663-668 */
            float mw = wd[0]; for (int r = 1; r < 24; ++r) if (wd[r] < mw || wd[r] != wd[r]) mw = wd[r];
            float mb = bd[0]; if (bd[1] < mb || bd[1] != bd[1]) mb = bd[1];
            /* rover.py:667-668: on fp16 distances the Python scalars 0.8 / 0.45 are compared as fp16 values */
            const float thr_w = half ? round_h(0.8f) : 0.8f, thr_b = half ? round_h(0.45f) : 0.45f;
            coll = (fabsf(mw) < thr_w) ? 1 : 0;
            if (fabsf(mb) < thr_b) coll = 1;
        }
        out->rock_collision[e] = coll;
        float tx = tgt[0] - pos[0], ty = tgt[1] - pos[1];
        float *ob = out->obs + (size_t)W*e;
        ob[0] = sqrtf(tx * tx + ty * ty) / 9.0f;                                 /* :320 */
        ob[1] = hd / 3.14159265358979323846f;                                    /* :321 math.pi -> f32 */
        ob[2] = in->lin_hist[3*e];                                               /* :322 */
        ob[3] = in->ang_hist[3*e];                                               /* :323 */
        /* :324-325 `sparse / 2`: an fp16 division in the as-shipped mode (it rounds in the fp16 subnormal range) */
        for (int i = 0; i < Ns; ++i) { float v = rd[in->sparse_idx[i]] / 2.0f; ob[4 + i] = half ? round_h(v) : v; }
        for (int i = 0; i < Nd; ++i) { float v = rd[in->dense_idx[i]] / 2.0f; ob[4 + Ns + i] = half ? round_h(v) : v; }
        /* ---- calculate_metrics, rover.py:460-531 ---- */
        float lin = in->lin_hist[3*e], lin_prev = in->lin_hist[3*e+1];
        float ang = in->ang_hist[3*e], ang_prev = in->ang_hist[3*e+1];
        float td = sqrtf(tx * tx + ty * ty);                                     /* :482 */
        float heading_pen = ((lin < 0.0f) ? -1.0f : 0.0f) * cfg->heading_contraint_reward;   /* :486 */
        float boogie = (fabsf(jn[0]) + fabsf(jn[1]) + fabsf(jn[2])) * cfg->boogie_contraint_reward;  /* :492 */
        float goal_pen = (fabsf(hd) > 2.0f) ? -fabsf(hd * 0.3f * cfg->goal_angle_reward) : 0.0f;     /* :495 */
        float dl = fabsf(lin * 3.0f - 3.0f * lin_prev), da = fabsf(ang * 3.0f - 3.0f * ang_prev);
        float p1 = (dl > 0.05f) ? dl * dl : 0.0f;                                /* :498 */
        float p2 = (da > 0.05f) ? da * da : 0.0f;                                /* :499 */
        float motion = (p1 * p1) * cfg->motion_contraint_reward;                 /* :500 */
        motion = motion + (p2 * p2) * cfg->motion_contraint_reward;              /* :502 */
        float pos_rew = (1.0f / (1.0f + ((float)(0.33 * 0.33) * td) * td)) * cfg->pos_reward;  /* :505 */
        if (td <= 0.18f) pos_rew = 1.03f * (float)(cfg->max_episode_length - in->progress[e]); /* :506 */
        float reward = pos_rew + heading_pen + motion + goal_pen;                /* :512 */
        int64_t tracker = 0;
        if (cfg->curriculum_level >= 2) {
            if (coll == 1) { tracker = cfg->num_envs_global; reward = reward - 300.0f; }   /* :517-519 */
        }
        reward = reward / 3000.0f;                                               /* :522 */
        out->rew[e] = reward;
        out->ex_pos_reward[e] = pos_rew; out->ex_collision[e] = tracker; out->ex_upright[e] = boogie;
        out->ex_heading[e] = heading_pen; out->ex_motion[e] = motion; out->ex_goal_angle[e] = goal_pen;
        out->ex_lin[e] = lin; out->ex_ang[e] = ang;
        /* ---- is_done, rover.py:610-647 (tilt from the PRE-physics euler) ---- */
        const float *ep = in->euler_pre + 3*e;
        const float tilt = (float)(0.78 * 1.5);
        int64_t reset = (in->progress[e] >= cfg->max_episode_length) ? 1 : 0;     /* :614 */
        if (fabsf(ep[0]) >= tilt) reset = 1;                                      /* :615 */
        if (fabsf(ep[1]) >= tilt) reset = 1;                                      /* :616 */
        if (td >= 11.0f) reset = 1;                                               /* :618 */
        if (td <= 0.18f) reset = 1;                                               /* :619 */
        if (cfg->curriculum_level >= 2 && coll == 1) reset = 1;                   /* :645-646 */
        out->reset[e] = reset;
    }
}

/* ---- reset path: stone clearance, spawn shift, goal validation, heightfield ------------- */

/* rover.py:536-538 / :655-658 with stone_info from terrain_utils.py:416-424 ([S,7] f32) */
static float clearance(const float *info7, int S, float x, float y) {
    float best = INFINITY;
    for (int s = 0; s < S; ++s) {
        float dx = x - info7[7*s], dy = y - info7[7*s+1];
        float d = sqrtf(dx * dx + dy * dy) - info7[7*s+6];
        if (d < best || d != d) best = d;
    }
    return best;
}

ORACLE_API void oracle_clearance(const float *info7, int S, int n, const float *xy, float *out) {
    for (int i = 0; i < n; ++i) out[i] = clearance(info7, S, xy[2*i], xy[2*i+1]);
}

/* avoid_pos_rock_collision, rover.py:649-661: x += 0.05 while clearance <= 1.4.  The reference
 * iterates globally until nothing moves; per env that is this loop.  max_iter bounds it. */
ORACLE_API int oracle_shift_spawns(const float *info7, int S, int n, float *pos3, int max_iter) {
    int worst = 0;
    for (int i = 0; i < n; ++i) {
        int it = 0;
        while (it < max_iter && clearance(info7, S, pos3[3*i], pos3[3*i+1]) <= 1.4f) { pos3[3*i] = pos3[3*i] + 0.05f; ++it; }
        if (it > worst) worst = it;
    }
    return worst;
}

/* get_pos_height, rover.py:588-608 */
ORACLE_API void oracle_pos_height(const float *hm, int N0, int N1, float hscale, float vscale,
                                  float shift_x, float shift_y, int n, const float *xy, float *out) {
    for (int i = 0; i < n; ++i) {
        int64_t ix = cell_index(xy[2*i], shift_x, hscale, N0);
        int64_t iy = cell_index(xy[2*i+1], shift_y, hscale, N0);
        if (iy > N1 - 1) iy = N1 - 1;
        out[i] = hm[(size_t)ix * N1 + iy] * vscale;
    }
}

/* generate_goals + random_goals + check_goal_collision, rover.py:533-564, with the uniforms of
 * every torch.rand draw supplied by the caller: draws[it] has n values, consumed in env_ids order.
 * Reproduces the env_ids = mask*env_ids aliasing (:540): accepted ids become 0, so env 0 is
 * redrawn on every further iteration.  Returns the number of draws used, or -1 if max_draws ran out. */
ORACLE_API int oracle_generate_goals(const float *info7, int S, int n, const int64_t *env_ids_in,
                                     const float *initial_pos3, float radius, const float *draws, int max_draws,
                                     float *target3) {
    int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    memcpy(ids, env_ids_in, sizeof(int64_t) * (size_t)n);
    int used = 0, bad = 1;
    while (bad > 0) {
        if (used >= max_draws) { free(ids); return -1; }
        const float *u = draws + (size_t)used * n; ++used;
        for (int i = 0; i < n; ++i) {                                             /* :554-564 */
            float alpha = (float)(2 * 3.14159265358979323846) * u[i];
            float x = radius * cosf(alpha) + 0.0f, y = radius * sinf(alpha) + 0.0f;
            int64_t id = ids[i];
            target3[3*id] = x + initial_pos3[3*id];
            target3[3*id+1] = y + initial_pos3[3*id+1];
        }
        bad = 0;
        for (int i = 0; i < n; ++i) {                                             /* :533-542 */
            int64_t id = ids[i];
            int m = clearance(info7, S, target3[3*id], target3[3*id+1]) <= 1.0f;
            ids[i] = m ? id : 0;
            bad += m;
        }
    }
    free(ids);
    return used;
}

/* done compaction, rover.py:356: nonzero(reset_buf) ascending */
ORACLE_API int oracle_compact(int n, const int64_t *reset, int64_t *ids) {
    int c = 0;
    for (int i = 0; i < n; ++i) if (reset[i] != 0) ids[c++] = i;
    return c;
}

/* Ackermann, tasks/utils/kinematics.py:13-67.  steer [n,6], vel [n,6] in wheel order FL,FR,ML,MR,RL,RR */
ORACLE_API void oracle_ackermann(int n, const float *lin_in, const float *ang_in, float *steer, float *vel) {
    static const float wl[6][2] = {{-0.385, 0.438}, {0.385, 0.438}, {-0.447, 0.0}, {0.447, 0.0}, {-0.385, -0.411}, {0.385, -0.411}};
    static const float side[6] = {-1.0f, 1.0f, -1.0f, 1.0f, -1.0f, 1.0f};
    for (int i = 0; i < n; ++i) {
        float lin = lin_in[i], ang = ang_in[i];
        float Px = copysignf(lin / ang, -ang);                                    /* :34-35 */
        Px = (fabsf(Px) > 0.45f) ? Px : 0.0f;                                     /* :38 */
        lin = (Px != 0.0f) ? lin : 0.0f;                                          /* :39 */
        for (int w = 0; w < 6; ++w) {
            float dx = Px - wl[w][0], dy = 0.0f - wl[w][1];
            float dist = sqrtf(dx * dx + dy * dy);                                /* :43 */
            float wheel_linear = copysignf(ang, lin);                             /* :49 */
            float wheel_turning = ang * side[w];                                  /* :51 */
            float av = (lin != 0.0f) ? wheel_linear : wheel_turning;              /* :52 */
            float mv = dist * av;                                                 /* :55 */
            if (dist > 1000.0f) mv = lin;                                         /* :58 */
            vel[6*i + w] = mv / 0.2f;                                             /* :61 */
            float sa = atan2f(wl[w][1], wl[w][0] - Px);                           /* :63 (both branches equal) */
            if (sa < (float)(-3.14 / 2)) sa = sa + 3.14159265358979323846f;       /* :64 */
            if (sa > (float)(3.14 / 2)) sa = sa - 3.14159265358979323846f;        /* :65 */
            steer[6*i + w] = sa;
        }
    }
}

/* the same for rays whose d = -normalize(direction) is given (what a ray record of the HIP path holds): the per-ray arithmetic
 * without the normalisation, in either arithmetic (half: the as-shipped fp16 sequence on fp16-valued s, d) */
ORACLE_API void oracle_raycast_unit(const oracle_knn_map *m, int n, const float *src3, const float *dneg3, int half, float *out) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) out[i] = half ? ray_min_core_h(m, src3 + 3*i, dneg3 + 3*i) : ray_min_core(m, src3 + 3*i, dneg3 + 3*i);
}

/* standalone ray cast of arbitrary rays (used to pin the kernel's inner loop in isolation) */
ORACLE_API void oracle_raycast(const oracle_knn_map *m, int n, const float *src3, const float *dir3, float *out) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) out[i] = ray_min_distance(m, src3 + 3*i, dir3 + 3*i);
}
