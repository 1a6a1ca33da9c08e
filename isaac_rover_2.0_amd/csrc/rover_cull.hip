// rover_cull.hip — the culled ray cast (raycast variant 3: cull_scan_kernel) and the staged ray cast built on its proof (variant 4:
// lane_scan_kernel, further down — the roofline kernel of the step on full batches since round 5).
//
// Reference: tasks/utils/camera/camera.py:60-145 (gather K triangles per ray, ray_distance, min over K),
//            tasks/utils/camera/ray_casting.py:31-59 (the (ray, triangle) test), rock_detect.py:52-149 (same on the rocks map).
//
// The reference evaluates all K = 200 triangles of a ray's cell and takes the min of k-or-11.0; a ray meets 1-3 of them.
// raycast_binned_kernel (rover_kernels.hip) evaluates them all too (134 VALU instructions per ray).  This kernel splits the
// work: PHASE 1 proves, per (ray, triangle), with 11-15 packed instructions per pair of triangles, that ray_casting.py:59
// rejects the triangle (it then contributes the 11.0 sentinel and nothing else); PHASE 2 runs the exact arithmetic of
// rover_raymath.h — the same code the other kernels run — on the few (ray, lane-pair) candidates phase 1 could not
// reject, 64 candidates of different rays side by side.  One kernel (cull_scan_kernel): a wave scans its run of 64 sorted
// rays, then finishes it.  Results are bit-identical to raycast_binned_kernel /
// raycast_kernel (tests/test_hip_parity.py::test_raycast_variants_bit_identical, tools/soak_exact.py).
//
// ---- why a culled triangle is rejected by the reference (the proof behind phase 1) -----------------------------------
// Per triangle the cull table holds a sphere centre m (any point) and, folded into the length of a scaled unit normal,
// r2 >= 1.06 (rho + 1e-4)^2, where rho = max distance from m to the corners of the PADDED triangle
// {a + n b + m c : n, m >= -0.101, n + m <= 1.101} (b = fl(v1 - a), c = fl(v0 - a) as ray_casting.py:35-36 computes them).
// With h = s - m, W = distance from m to the ray's line, phase 1 culls iff
//     (A)  0.99925 |h|^2 - (h.d)^2 > r2        [ => W >= rho + 0.005 (|h| + 2 rho), rounding slack included ]
//     (B)  |n_dec . d| > 1.3e-2                 [ n_dec = the stored fp16 unit normal, within 1e-3 of N / |N|, N = b x c ]
// and the triangle is no sliver (|N| >= 0.05 |b| |c|; slivers, overflows and NaNs are stored as "always a candidate").
// The reference accepts iff fl(nn/det) >= -fp16(0.1), fl(mn/det) >= -fp16(0.1), fl(n + m) <= fp16(1.1)  (ray_casting.py:59),
// i.e. the point a + (nn/det) b + (mn/det) c lies in the padded triangle, hence within rho of m.  Multiplying by det and
// using the Cramer identity  nn* b + mn* c = det* g - kn* d  (exact triple products, g = s - a):
//     |det*| W - err  <=  rho (|det*| + err'),   err, err' <= 1e-6 |b||c| (2|g| + rho)   [f32 rounding of nn, mn, det: 16 eps]
// so with (A):  |det*| <= 2e-6 |b||c| / 0.005 = 4e-4 |b||c|.  But (B) and the sliver bound give
// |det*| = |N . d| >= (1.3e-2 - 1e-3) * 0.05 |b||c| = 6e-4 |b||c|  — a contradiction: the reference rejects.
// (err: the f32 rounding of the three triple products, each a sum of three products of a cross-product component — two products and a
//  subtraction — with a component of d or g: <= 6 eps of the product of the three lengths, eps = 2^-24, against the 16 eps allowed here.)
// (Rounds 2-5 ran (A) with the margin 0.02 (c_a = 0.995, c_rho = 1.19) against tau = 4e-3: the same factor 1.5 between the two sides.  Round 6
//  quartered the margin and paid with tau: a steep ray's candidates are decided by (A) — its sphere about a 0.1 m grid triangle shrinks from
//  0.116 to 0.098 m at |h| = 0.8 m, a quarter fewer pairs reach the exact arithmetic: one launch 338 -> 320 (margin 0.01) -> 308 us —, while (B)
//  matters for rays within 0.75 degrees of a triangle's plane.  (A)'s constants: W^2 >= lambda1 |h|^2 + lambda2 rho^2 gives W >= a |h| + b rho
//  for every a^2 / lambda1 + b^2 / lambda2 <= 1 (Cauchy-Schwarz); lambda1 = 1 - c_a / |d|^2 - 3e-6 >= 0.000737, lambda2 = 1.05999, b = 1.0102:
//  a = 0.00524 >= 0.005 — the static_assert below.)
// NaN / inf anywhere makes (A) or (B) compare false, i.e. keeps the triangle a candidate.  DESIGN.md §5 has the long form.
// (B) for a whole cell at once: with q = min over the cell's triangles of |N_z| / |N| (0 if any is a sliver) and beta the
// ray's angle from the vertical, every triangle has |N . d| / |N| >= cos(acos q + beta), which exceeds 1.25e-2 iff
// q > 1.25e-2 |d_z| + sqrt(1 - 1.25e-2^2) |d_xy| (ROVER_CONE_TAU: 1.25e-2 x sigma = 6.25e-4 against (A)'s 4e-4).  prep_rays_kernel stores the right-hand side (rounded up to 16 bits) in the
// ray record, the id row carries q (rounded down): where q wins, phase 1 runs test (A) only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <vector>
#include <type_traits>
#include "rover_internal.h"
#include "rover_raymath.h"

namespace rover {

// Constants of the rejection proof, per arithmetic of the exact phase (H = 0: f32, the header above; H = 1: the reference's
// as-shipped fp16 arithmetic, derivation below).  Test (A): c_a |h|^2 - (h.d)^2 > r2 with r2 >= c_rho (rho + 1e-4)^2; test (B):
// |n_dec . d| > tau |n_dec| (tau folded into the stored normal's length: r2 = tau^2 |n_stored|^2); a triangle with
// |b x c| < sigma |b||c| (or |b||c| < min_bc) is stored as "always a candidate".
template <int H> struct CullK;
template <> struct CullK<0> {
    static constexpr double pad = 0.101, c_rho = 1.06, tau = 1.3e-2, sigma = 0.05, min_bc = 0.0;
    static constexpr double alpha = 0.005;                   // (A)'s margin: W >= rho + alpha (|h| + 2 rho)
    static constexpr float c_a = 0.99925f, tau2 = 1.6901e-4f;
};
// ---- as-shipped fp16 arithmetic (ray_casting.py:31-59 on Half tensors; u = 2^-11) ------------------------------------------
// The exact phase then works on b = fl16(v1 - a), c = fl16(v0 - a), g' = fl16(s - a) (|g' - g| <= sqrt(3) u |g|: the ray as if
// cast from s' = a + g', within 8.5e-4 |g| of s) and the fp16-normalised d (|d|^2 within 4e-3 of 1); the tables are built from
// those b, c.  First-order error of a computed triple product, e.g. det = fl(fl(fl(N0 d0) + fl(N1 d1)) + fl(N2 d2)) with
// N_i = fl(fl(b_j c_k) - fl(b_k c_j)):  u [ sum_i |d_i| (|b_j c_k| + |b_k c_j|)  (the two products of a cross component: the
// permanent of the matrix (|b|, |c|, |d|), <= 1.155 |b||c||d|)  + sum_i |N_i d_i|  (the subtraction)  + sum_i |N_i d_i|  (the three
// products with d)  + |N0 d0 + N1 d1| + |det|  (the two additions) ]  <= 5.155 u |b||c||d| = 2.53e-3 |b||c| with |d| <= 1.002; the same
// for nn on (g', c) and mn on (b, g').  Products below 2^-14 round with an absolute error of 2^-25 instead: at most 12 of them,
// which adds <= 5.4e-4 |b||c| to the bound below once |b||c| >= 1e-3.  So nn, mn, det are within e |g||c|, e |b||g|, e |b||c| of the
// exact triple products nn*, mn*, det* of (b, c, g', d) with e = 3.07e-3; the tables use e = 3.6e-3.
// The reference accepts iff fl16(nn/det) >= -fp16(0.1), fl16(mn/det) >= -fp16(0.1),
// fl16(n + m) <= fp16(1.1) (all finite), which puts Q = (nn b + mn c) / det inside the triangle padded by 0.105 (0.1 (1 + u),
// 1.1 (1 + u) + u (|n| + |m|) <= 1.1009), i.e. |a + Q - m| <= rho.  With the Cramer identity nn* b + mn* c = det* g' - kn* d:
// det (a + Q - m) = det* (s' - m) - kn* d + Delta, |Delta| <= e |b||c| (2 |g| + rho), and perpendicular to d:
// |det*| W' - |Delta| <= (|det*| + e |b||c|) rho, hence |det*| (W' - rho) <= 2 e |b||c| (|h| + 2 rho).
// Test (A) with c_a = 0.958, c_rho = 1.41 gives W >= rho + 0.061 (|h| + 2 rho) (f32 rounding and |d|^2 != 1 included), i.e.
// W' - rho >= eta (|h| + 2 rho), eta = 0.06, so |det*| <= 2 e / eta |b||c| = kappa |b||c|, kappa = 0.12.  |det*| = |N* . d| =
// |N*| |d| |cos|, |N*| = sin(theta) |b||c| (theta: the angle of the triangle at a), so (B) has to establish |cos(N*, d)| > kappa /
// (0.998 sin(theta)) — a threshold PER TRIANGLE: tau_t = 1.002 (kappa / (0.998 sin(theta)) + 1e-3) on the decoded normal (within
// 1e-3 of N* / |N*|) and the fp16-normalised d.  A right-angled triangle gets 0.122, a 45 degree one 0.172; triangles below
// sin(theta) = 0.34 (tau_t > 3 kappa), with |b||c| < 1e-3 or non-finite are stored as "always a candidate".  The record carries
// F = (tau_t / kappa)^2 as a 12-bit code in the low mantissa bits of the centre's x and y (6 each; rho is computed from the
// centre AS DECODED, so the bits cost no rigour): r2 (A) = kappa^2 |n_stored|^2 as before, r2 (B) = r2 (A) F.
// The cone of a cell: with beta the ray's angle from the vertical and q_t = |N*_z| / |N*|, |cos(N*, d)| >= cos(acos q_t + beta),
// so (B) holds for the whole cell iff beta <= gamma = min_t (acos(tau_t) - acos(q_t)); cells store gamma and rays beta as 16-bit
// fractions of pi / 2 (the f32 proof's cone compares cosines instead: its tau is one constant).
// The margins are an order of magnitude coarser than in f32 because every fp16 operation loses 2^-11; they cost candidates
// (measured: DESIGN.md), never correctness.
template <> struct CullK<1> {
    static constexpr double pad = 0.105, sigma = 0.0, min_bc = 1.0e-3, e_fp16 = 3.6e-3;
};
// eta is the one free parameter of the fp16 proof (a larger eta loosens (A): more candidates by distance; it tightens nothing but
// lowers kappa = 2 e / eta and with it (B)'s thresholds: fewer candidates by orientation, more cells with a cone).  Everything
// else follows: alpha = eta + 1e-3, beta = 1 + 2 alpha, (beta rho + alpha |h|)^2 <= (beta^2 + 2 alpha beta) rho^2 +
// (alpha^2 + alpha beta / 2) |h|^2, then the |d|^2 and f32-rounding allowances.
CullProofH cull_proof_h(double eta, double split) {
    CullProofH k{};
    // (beta rho + alpha |h|)^2 = beta^2 rho^2 + 2 alpha beta rho |h| + alpha^2 |h|^2 <= (beta^2 + split alpha beta) rho^2 + (alpha^2 + alpha beta / split) |h|^2
    // for every split > 0 (2 rho |h| <= split rho^2 + |h|^2 / split); the bound is tight at |h| = split rho: rounds 3-5 used split = 2, but a ray's
    // origin lies 5-10 radii from the triangles it could hit
    const double e = CullK<1>::e_fp16, a = eta + 1.0e-3, b = 1.0 + 2.0 * a;
    k.kappa = 2.0 * e / eta;
    k.c_rho = (b * b + split * a * b) * 1.004 + 0.005;
    k.c_a = (float)(0.996 * (1.0 - (a * a + a * b / split)) - 0.0005);
    k.tau2 = (float)(k.kappa * k.kappa);
    return k;
}
static_assert(0.999 * (CullK<0>::tau - 1.0e-3) * CullK<0>::sigma > 1.45 * 2.0 * 1.0e-6 / CullK<0>::alpha, "f32 cull proof: (B) must contradict (A), with the factor 1.5 of rounds 2-5");
static_assert((double)CullK<0>::tau2 >= CullK<0>::tau * CullK<0>::tau && (double)CullK<0>::tau2 < 1.001 * CullK<0>::tau * CullK<0>::tau, "tau2 = tau^2");
static_assert(ROVER_CONE_TAU * CullK<0>::sigma > 1.45 * 2.0 * 1.0e-6 / CullK<0>::alpha, "f32 cull proof: the set cone's (B) must contradict (A)");
// (A) gives the margin alpha: lambda1 = 1 - c_a / 0.99999 - 3e-6, lambda2 = c_rho 0.99999 (1 - 3e-7), b = 1 + 2 alpha + 2e-4 (what the staged
// kernel's relative records need on top): a^2 = lambda1 (1 - b^2 / lambda2) >= (1.04 alpha)^2
static_assert((1.0 - (double)CullK<0>::c_a / 0.99999 - 3.0e-6) * (1.0 - (1.0 + 2.0 * CullK<0>::alpha + 2.0e-4) * (1.0 + 2.0 * CullK<0>::alpha + 2.0e-4) / (CullK<0>::c_rho * 0.99999 * (1.0 - 3.0e-7)))
              >= (1.04 * CullK<0>::alpha) * (1.04 * CullK<0>::alpha), "f32 cull proof: test (A)'s constants must give W >= rho + alpha (|h| + 2 rho)");
#define CULL_RUNMAX 64               // sorted rays per wave (one result slot per lane)
#define CULL_RING   2                // id rows (one bin each) in flight per wave: global -> LDS loads issued this many bins ahead

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 cvt2(uint32_t d) {
    const half2v h = __builtin_bit_cast(half2v, d);
    return f2{(float)h.x, (float)h.y};
}

// r2 exactly as phase 1 derives it from the decoded normal (the table builder verifies its encoding with this)
__device__ __forceinline__ float cull_r2(float nx, float ny, float nz, float tau2) {
    float q = nx * nx;
    q = __builtin_fmaf(ny, ny, q);
    q = __builtin_fmaf(nz, nz, q);
    return q * tau2;
}

// ---------------------------------------------------------------------------------------------------
// init (rover_set_knn_map): three tables per map, all indexed through the reference's own map_indices
//   rtab [T]        20 B: the triangle's nine fp16 vertex components (v0 xyz, v1 xyz, v2 xyz, pad) — the exact arithmetic's
//                         input, what camera.py:84 gathers through triangles -> vertices
//   ctab [T]        16 B: {centre x, centre y (f32), half2(centre z, nx), half2(ny, nz)} — bounding-sphere centre of the padded
//                         triangle and its unit normal scaled to length r / tau (r2 = tau^2 |n|^2 is the sphere test's radius)
//   idx4 [cell][L][4] i32: the cell's K triangle ids (-1 = empty slot), L = K8 / 4 lanes x 4; ids sorted ascending and dealt
//                         as pairs of neighbours (pair m = sorted[2m], sorted[2m + 1] -> lane m % L, pair slot m / L) so
//                         that one gather instruction of a wave touches neighbouring ctab records (few L2 lines)
// The ids in these tables are INTERNAL: rover_set_knn_map renumbers the caller's triangles along a Morton curve over their
// centroids, so that consecutive ids are neighbours in space whatever order the mesh file lists them in (a decimated .ply has
// none).  A min over the same set of triangles does not depend on how they are numbered: results are unchanged.
// The per-triangle tables are small (26 B x T: 18 MB for 720 k triangles) and stay in L2 / MALL; HBM only streams idx4.
// ---------------------------------------------------------------------------------------------------
// centroid (xy) of every triangle from its fp16 vertices: the key of the internal renumbering (NaN for a broken triangle)
__global__ void __launch_bounds__(256) tri_centroid_kernel(const int32_t* __restrict__ tris, const uint16_t* __restrict__ verts,
                                                           uint32_t T, uint32_t V, float2* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float x = 0.0f, y = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t vi = (uint32_t)tris[3ull * t + a];
        const _Float16* v = reinterpret_cast<const _Float16*>(verts) + 3ull * (vi < V ? vi : 0u);
        x += vi < V ? (float)v[0] : __builtin_nanf("");
        y += vi < V ? (float)v[1] : __builtin_nanf("");
    }
    out[t] = make_float2(x * (1.0f / 3.0f), y * (1.0f / 3.0f));
}

// record t of the per-triangle tables belongs to the caller's triangle order[t] (the internal, spatially sorted numbering)
__global__ void __launch_bounds__(256) rtab_build_kernel(const int32_t* __restrict__ tris, const uint16_t* __restrict__ verts,
                                                         uint32_t T, uint32_t V, const uint32_t* __restrict__ order,
                                                         uint16_t* __restrict__ rtab) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t src = order[t];                                      // 0xffffffff: a hole of the numbering (the partner slot of a single)
    uint16_t v[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) v[q] = 0x7e00u;                        // fp16 NaN: every test fails
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t vi = src != 0xffffffffu ? (uint32_t)tris[3ull * src + a] : 0xffffffffu;
        if (vi < V) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[3 * a + c] = verts[3ull * vi + c];
        }
    }
    v[9] = 0;
#pragma unroll
    for (int q = 0; q < 10; ++q) rtab[10ull * t + q] = v[q];
}

template <int H>
__global__ void __launch_bounds__(256) ctab_build_kernel(const uint16_t* __restrict__ rtab, uint32_t T, const uint32_t* __restrict__ order,
                                                         uint4* __restrict__ ctab, float* __restrict__ nz_abs, uint32_t* __restrict__ counts,
                                                         CullProofH ph /* fp16 proof only */) {
    typedef CullK<H> KK;
    const double k_tau = H ? ph.kappa : CullK<0>::tau, k_crho = H ? ph.c_rho : CullK<0>::c_rho;
    const float k_tau2 = H ? ph.tau2 : CullK<0>::tau2;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const _Float16* src = reinterpret_cast<const _Float16*>(rtab) + 10ull * t;
    float v[9];
    bool valid = true;
#pragma unroll
    for (int q = 0; q < 9; ++q) { v[q] = (float)src[q]; valid = valid && (v[q] == v[q]) && fabsf(v[q]) < 6.0e4f; }
    float mk[3] = {0.0f, 0.0f, 0.0f};                                // the centre as phase 1 decodes it
    float nzq = 2.0f;                                                // |N_z| / |N| of the exact normal; 2 = always a candidate (not part of a cell's cone)
    uint16_t zh = 0, nh[3] = {0x7c00u, 0x7c00u, 0x7c00u};            // infinite normal: r2 = +inf, neither test can hold = always a candidate
    if (valid) {
        // a, b, c exactly as ray_casting.py:34-36 / set_pair (f32) / set_pair_h (each difference rounded to fp16) compute them, widened
        const float af[3] = {v[6], v[7], v[8]};
        float bf[3] = {v[3] - v[6], v[4] - v[7], v[5] - v[8]};
        float cf[3] = {v[0] - v[6], v[1] - v[7], v[2] - v[8]};
        if (H) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { bf[k] = (float)(_Float16)bf[k]; cf[k] = (float)(_Float16)cf[k]; }
        }
        double Q[3][3];                                              // corners of the padded triangle
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = af[k], b = bf[k], c = cf[k];
            Q[0][k] = a - KK::pad * b - KK::pad * c;
            Q[1][k] = a + (1.0 + 2.0 * KK::pad) * b - KK::pad * c;
            Q[2][k] = a - KK::pad * b + (1.0 + 2.0 * KK::pad) * c;
        }
        auto d2 = [&](int x, int y) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (Q[x][k] - Q[y][k]) * (Q[x][k] - Q[y][k]);
            return s;
        };
        // centre of the smallest enclosing sphere: circumcentre, or the midpoint of the longest edge when not acute
        const double A = d2(1, 2), B = d2(0, 2), C = d2(0, 1);
        double w0, w1, w2;
        if (A >= B + C) { w0 = 0.0; w1 = 0.5; w2 = 0.5; }
        else if (B >= A + C) { w0 = 0.5; w1 = 0.0; w2 = 0.5; }
        else if (C >= A + B) { w0 = 0.5; w1 = 0.5; w2 = 0.0; }
        else {
            w0 = A * (B + C - A); w1 = B * (C + A - B); w2 = C * (A + B - C);
            const double ws = w0 + w1 + w2;
            w0 /= ws; w1 /= ws; w2 /= ws;
        }
        double m[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) m[k] = w0 * Q[0][k] + w1 * Q[1][k] + w2 * Q[2][k];
        mk[0] = (float)m[0]; mk[1] = (float)m[1];
        // (normal first: the fp16 proof's per-triangle threshold code goes into the centre's low mantissa bits, rho comes after)
        const double N[3] = {(double)bf[1] * cf[2] - (double)bf[2] * cf[1], (double)bf[2] * cf[0] - (double)bf[0] * cf[2],
                             (double)bf[0] * cf[1] - (double)bf[1] * cf[0]};
        const double nN = sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        const double nb = sqrt((double)bf[0] * bf[0] + (double)bf[1] * bf[1] + (double)bf[2] * bf[2]);
        const double nc = sqrt((double)cf[0] * cf[0] + (double)cf[1] * cf[1] + (double)cf[2] * cf[2]);
        bool ok = nN > 0.0 && nN >= KK::sigma * nb * nc && nb * nc >= KK::min_bc;      // slivers (fp16: tiny triangles too) stay candidates
        double tau_true = k_tau;                                     // what |cos(N*, d)| has to exceed (true normal, unit d)
        if (H && ok) {
            const double sin_t = nN / (nb * nc);
            tau_true = k_tau / (0.998 * sin_t);
            const double tau_t = 1.002 * (tau_true + 1.0e-3);        // on the decoded normal and the fp16-normalised d
            const double F = (tau_t / k_tau) * (tau_t / k_tau);
            const double code = ceil((F - 1.0) * 512.0);
            ok = code >= 0.0 && code <= 4095.0 && tau_t < 0.999;
            if (ok) {
                const uint32_t cc = (uint32_t)code;
                mk[0] = __uint_as_float((__float_as_uint(mk[0]) & ~63u) | (cc & 63u));
                mk[1] = __uint_as_float((__float_as_uint(mk[1]) & ~63u) | (cc >> 6));
            }
        }
        const _Float16 zq = (_Float16)(float)m[2];                   // |z| < 6e4 (the vertices are fp16 values)
        zh = __builtin_bit_cast(uint16_t, zq);
        mk[2] = (float)zq;
        double rho2 = 0.0;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (Q[x][k] - (double)mk[k]) * (Q[x][k] - (double)mk[k]);
            rho2 = s > rho2 ? s : rho2;
        }
        const double rho = sqrt(rho2) + 1.0e-4;
        const double need = k_crho * rho * rho;
        if (ok) {
            double scale = sqrt(need) * 1.002 / k_tau / nN;          // |stored normal| = r / tau
            bool done = false;
            for (int it = 0; it < 8 && !done; ++it, scale *= 1.002) {
                float dec[3];
                bool fin = true;
                for (int k = 0; k < 3; ++k) {
                    const float f = (float)(N[k] * scale);
                    fin = fin && fabsf(f) < 6.0e4f;
                    const _Float16 hn = (_Float16)f;
                    nh[k] = __builtin_bit_cast(uint16_t, hn);
                    dec[k] = (float)hn;
                }
                if (!fin) break;
                done = (double)cull_r2(dec[0], dec[1], dec[2], k_tau2) >= need;
            }
            ok = done;
        }
        if (!ok) nh[0] = nh[1] = nh[2] = 0x7c00u;
        else if (!H) nzq = (float)(fabs(N[2]) / nN * (1.0 - 1.0e-6));
        else {      // gamma_t = acos(tau) - acos(q), as a fraction of pi / 2 (<= 0: the cell has no cone)
            const double g = (acos(fmin(1.0, tau_true * (1.0 + 1.0e-3))) - acos(fmin(1.0, fabs(N[2]) / nN)) - 2.0e-4) / 1.5707963267948966;
            nzq = g > 0.0 ? (float)(g * (1.0 - 1.0e-6)) : 0.0f;
        }
    }
    nz_abs[t] = nzq;
    if (counts && nh[0] == 0x7c00u && order[t] != 0xffffffffu) atomicAdd(counts + 0, 1u);     // always a candidate (rover_get_cull_info); holes of the numbering aside
    ctab[t] = make_uint4(__float_as_uint(mk[0]), __float_as_uint(mk[1]), (uint32_t)zh | ((uint32_t)nh[0] << 16),
                         (uint32_t)nh[1] | ((uint32_t)nh[2] << 16));
}

// one workgroup per cell: the cell's K ids (internal numbering) sorted ascending, paired and dealt to the lanes;
// qrow[cell] = q16: the cell's normal cone as a 16-bit fraction rounded down (0 = none) — f32 proof: q = min |N_z| / |N| over the
// cell's triangles; fp16 proof: the largest angle from the vertical a ray may have, over pi / 2.  prep_rays_kernel compares it with
// the ray's own bound and leaves the verdict in the ray record (flags bit 2)
__global__ void __launch_bounds__(256) idx4_build_kernel(const int32_t* __restrict__ map_idx, uint32_t K, uint32_t K8, uint32_t T,
                                                         const uint32_t* __restrict__ newid, const float* __restrict__ nz_abs,
                                                         const uint16_t* __restrict__ rtab, uint32_t Y, float cell_size, float shift_x,
                                                         float shift_y, int32_t* __restrict__ idx4, uint32_t* __restrict__ qrow,
                                                         uint32_t* __restrict__ counts) {
    __shared__ uint32_t key[256];
    __shared__ float qmin[256];
    const uint32_t cell = blockIdx.x, tid = threadIdx.x, L = K8 >> 2;
    uint32_t k = 0xffffffffu;
    if (tid < K) { const uint32_t t = (uint32_t)map_idx[(uint64_t)cell * K + tid]; if (t < T) k = newid[t]; }     // internal numbering
    key[tid] = k;
    qmin[tid] = k != 0xffffffffu ? nz_abs[k] : 2.0f;
    __syncthreads();
    for (uint32_t len = 2; len <= 256u; len <<= 1) {
        for (uint32_t stride = len >> 1; stride > 0; stride >>= 1) {
            if (tid < 128u) {
                const uint32_t lo = ((tid / stride) * stride << 1) + (tid % stride), hi = lo + stride;
                const bool up = (lo & len) == 0;
                const uint32_t a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (uint32_t sdt = 128u; sdt > 0; sdt >>= 1) {
        if (tid < sdt) qmin[tid] = fminf(qmin[tid], qmin[tid + sdt]);
        __syncthreads();
    }
    // Pairing.  The internal numbering gives two triangles that were matched as spatial partners (rover_set_knn_map: mutual
    // nearest centroids — on a grid mesh the two halves of a mesh cell) the ids 2p, 2p + 1.  Partners that are both in this
    // cell's list share a packed pair of a lane, so that a queue entry tends to hold two candidates; the rest (singles: the
    // partner is not in the list, or the triangle has none) are paired up in id order behind them.  Pair m goes to lane m % L,
    // pair slot m / L, which keeps one gather instruction of a wave on neighbouring records.
    __shared__ int32_t row[256];
    __shared__ uint32_t wsum[2][4];
    row[tid] = -1;
    const uint32_t id = key[tid];
    const bool present = id != 0xffffffffu;
    const bool first = present && !(id & 1u) && tid + 1u < 256u && key[tid + 1u] == (id | 1u);
    const bool second = present && (id & 1u) && tid > 0u && key[tid - 1u] == (id ^ 1u);
    const bool single = present && !first && !second;
    const uint64_t mf = __builtin_amdgcn_ballot_w64(first), ms = __builtin_amdgcn_ballot_w64(single);
    const uint32_t wv = tid >> 6, ln = tid & 63u;
    if (ln == 0u) { wsum[0][wv] = (uint32_t)__builtin_popcountll(mf); wsum[1][wv] = (uint32_t)__builtin_popcountll(ms); }
    __syncthreads();
    uint32_t pf = (uint32_t)__builtin_popcountll(mf & ((1ull << ln) - 1ull)), ps = (uint32_t)__builtin_popcountll(ms & ((1ull << ln) - 1ull));
    uint32_t n_pairs = 0;
    for (uint32_t w2 = 0; w2 < 4u; ++w2) {
        if (w2 < wv) { pf += wsum[0][w2]; ps += wsum[1][w2]; }
        n_pairs += wsum[0][w2];
    }
    // (a `second` sits right behind its `first`: the same pair index; across a wave boundary pf already counts that first)
    const uint32_t m = first ? pf : (second ? pf - 1u : n_pairs + (ps >> 1)), e = first ? 0u : (second ? 1u : (ps & 1u));
    // Near and far.  A lane holds two pairs: slot 0 takes the L pairs NEAREST to the cell's centre (by the nearer centroid of a
    // pair), slot 1 the others — the farther half of a K = 200 list.  A ray of the cell passes close to the centre, so most rays
    // can be shown to clear every far triangle at once (far_build_kernel, cull_scan_kernel) and skip slot 1 altogether.  The choice
    // is a matter of speed only: whatever the split, the far bound is computed from the triangles that ended up in slot 1.
    __shared__ float pkey[128];
    __shared__ uint8_t pfar[128];
    if (tid < 128u) { pkey[tid] = __builtin_inff(); pfar[tid] = 0; }
    __syncthreads();
    if (present && m < 128u) {
        const _Float16* v = reinterpret_cast<const _Float16*>(rtab) + 10ull * id;
        const float cx = ((float)v[0] + (float)v[3] + (float)v[6]) * (1.0f / 3.0f), cy = ((float)v[1] + (float)v[4] + (float)v[7]) * (1.0f / 3.0f);
        const float ccx = (float)(cell / Y) * cell_size + shift_x, ccy = (float)(cell % Y) * cell_size + shift_y;
        float d = sqrtf((cx - ccx) * (cx - ccx) + (cy - ccy) * (cy - ccy));
        if (!(d == d)) d = 0.0f;                                            // a broken triangle: near (it is always a candidate anyway)
        atomicMin(reinterpret_cast<uint32_t*>(&pkey[m]), __float_as_uint(d));   // non-negative floats order like their bits
    }
    __syncthreads();
    const uint32_t np_total = n_pairs + ((wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3] + 1u) >> 1);
    const uint32_t n_far = np_total > L ? min(np_total - L, L) : 0u;
    if (tid < np_total && tid < 128u) {                                     // rank of pair `tid` among the pairs, farthest first (ties: larger index first)
        const float kd = pkey[tid];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < np_total && q < 128u; ++q) rank += (pkey[q] > kd || (pkey[q] == kd && q > tid)) ? 1u : 0u;
        pfar[tid] = rank < n_far ? 1u : 0u;
    }
    __syncthreads();
    if (present && m < 128u) {
        const uint32_t far = pfar[m];
        uint32_t pos = 0;                                                   // pairs of the same set in front of this one (id order)
        for (uint32_t q = 0; q < m; ++q) pos += pfar[q] == far ? 1u : 0u;
        if (pos < L) row[pos * 4u + 2u * far + e] = (int32_t)id;
    }
    __syncthreads();
    if (tid < K8) idx4[(uint64_t)cell * K8 + tid] = row[tid];
    if (tid == 0) {
        const float q = qmin[0];
        uint32_t q16 = 0;
        if (q <= 1.0f && q > 0.0f) { q16 = (uint32_t)floorf(q * 65535.0f); q16 = q16 > 0xfffeu ? 0xfffeu : q16; }
        qrow[cell] = q16;
        if (counts && q16 == 0u) atomicAdd(counts + 1, 1u);             // no normal cone: its rays run both tests on every pair
    }
}

// One wave per cell: the bound of its FAR pairs (slot 1 of every lane), for one proof's tables.  For a ray with origin s and unit
// direction at angle beta from the vertical and a triangle with decoded centre m, r2:  c_a |h|^2 - (h.d)^2 = c_a W^2 - (dd - c_a) a^2
// (W: distance from m to the ray's line, a: the axial part of h, dd >= |d|^2), W >= cos(beta) (dist_xy(m, C) - e), e = dist_xy(s, C) +
// |s_z - m_z| tan(beta) (the line's offset from the cell's centre C at the height of m), |h| <= hmax.  So test (A) holds for EVERY
// far triangle if  min_t (dist_xy(m_t, C) - k1 sqrt(r2_t))  >  e_max + k2 a_max  with k1 = 1 / (0.9 sqrt(c_a)), k2 = sqrt(dd - c_a) k1, a_max >= |a| and
// cos(beta) >= 0.9.  The record holds the left side G (-inf if a far triangle is always a candidate, +inf if slot 1 is empty), the z
// range of the far centres, the largest dist_xy(m_t, C) and C.
// The same bound for the NEAR pairs (slot 0): a ray that clears both sets on the cone path has no candidate at all in this cell — most
// rock rays (the nearest rock triangle of most cells is decimetres to metres away) — and is not scanned; a bin whose rays all do is never set up.
struct FarRec { float G, z0, z1, rho_out, cx, cy; uint32_t q16, pad1; };    // q16: the cell's normal cone (qrow), here so that the scan's prologue needs one gather
// (the near bounds {G, z0, z1, rho} are a table of their own behind the n_cells far records: only one of the scan kernels reads them, and
//  48-byte records cost the others 0.5-1.5 %)
__global__ void __launch_bounds__(256) far_build_kernel(const int4* __restrict__ idx4, const uint4* __restrict__ ctab, uint64_t n_cells, uint32_t K8,
                                                        uint32_t Y, float cell_size, float shift_x, float shift_y, float k1, float tau2,
                                                        const uint32_t* __restrict__ qrow, FarRec* __restrict__ out, float4* __restrict__ out_near,
                                                        uint32_t* __restrict__ n_useful /* cells whose bound can hold for a usual ray, or null */) {
    const uint64_t cell = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u, L = K8 >> 2;
    if (cell >= n_cells) return;
    const float ccx = (float)(cell / Y) * cell_size + shift_x, ccy = (float)(cell % Y) * cell_size + shift_y;
    // set 0: the far pairs (slot 1 of every lane), set 1: the near pairs (slot 0)
    float G[2] = {__builtin_inff(), __builtin_inff()}, z0[2] = {__builtin_inff(), __builtin_inff()}, z1[2] = {-__builtin_inff(), -__builtin_inff()}, ro[2] = {0.0f, 0.0f};
    if (lane < L) {
        const int4 id4 = idx4[cell * L + lane];
        const int32_t ids[4] = {id4.z, id4.w, id4.x, id4.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ids[j] < 0) continue;
            const int q = j >> 1;
            const uint4 r = ctab[ids[j]];
            const f2 zn = cvt2(r.z), w = cvt2(r.w);
            const float mx = __uint_as_float(r.x), my = __uint_as_float(r.y), mz = zn.x;
            const float r2 = cull_r2(zn.y, w.x, w.y, tau2);
            const float dxy = sqrtf((mx - ccx) * (mx - ccx) + (my - ccy) * (my - ccy));
            float g = (dxy - k1 * sqrtf(r2) * 1.00001f) * 0.99999f - 1.0e-6f;
            if (!(g == g) || !(r2 < 3.0e38f)) g = -__builtin_inff();        // always a candidate / broken: never skipped
            G[q] = fminf(G[q], g); z0[q] = fminf(z0[q], mz); z1[q] = fmaxf(z1[q], mz); ro[q] = fmaxf(ro[q], dxy);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            G[q] = fminf(G[q], __shfl_xor(G[q], off)); z0[q] = fminf(z0[q], __shfl_xor(z0[q], off)); z1[q] = fmaxf(z1[q], __shfl_xor(z1[q], off));
            ro[q] = fmaxf(ro[q], __shfl_xor(ro[q], off));
        }
    if (lane == 0u) {
        out[cell] = FarRec{G[0], z0[0], z1[0], ro[0], ccx, ccy, qrow[cell], 0u};
        out_near[cell] = make_float4(G[1], z0[1], z1[1], ro[1]);
        // A ray's side of the inequality is its line's offset from the cell's centre at the far centres' height — 0.1-0.15 m for a
        // heightmap ray of a rover on gentle ground — plus k2 hmax: cells with G under 0.2 m hardly ever skip (rover_capi.cpp: cull_args
        // picks the kernel that fetches the far records with the near ones when most cells are like that)
        if (n_useful && G[0] >= 0.2f) atomicAdd(n_useful, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------
// the kernels
// ---------------------------------------------------------------------------------------------------
struct CullRegs { f2 mx[2], my[2], mz[2], nx[2], ny[2], nz[2], r2[2], r2b[2]; };     // per lane: 4 triangles as 2 packed pairs (r2b: test (B)'s threshold where it is per triangle)

// LDS traffic between the lanes of ONE wave: the hardware keeps a wave's LDS operations in order; this only stops the
// compiler from moving them across
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define CULL_NOID 0x3ffffffu          // "no triangle" in a queue entry (ids are < 2^26 - 1)

struct RawTri { uint32_t d[5]; };    // rtab record: v0 xyz, v1 xyz, v2 xyz, pad as ten fp16 values (4-byte aligned)

__device__ __forceinline__ float lane_bcast(float v, uint32_t src_lane /* wave-uniform */) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)src_lane));
}

// ---------------------------------------------------------------------------------------------------
// PHASE 2 — cull_exact: the wave that scanned a run finishes it.  One lane per queue entry, 64 entries at a time: the exact
// arithmetic of rover_raymath.h (the code every other ray-cast kernel runs) on the entry's pair of triangles, then the min over
// the entries of a ray: a segmented wave reduction (a ray's entries are contiguous), one LDS atomicMin per ray and slice on an
// ordered-u32 key, and at the end one plain store per ray of the run — the 11.0 sentinel where a ray had no candidate at all
// (a culled triangle contributes exactly that, ray_casting.py:27,59).  The phase is a chain of dependent gathers (entry ->
// triangle records, ray record) that leaves the VALU idle; it runs AFTER the wave's scan loop, when the 28 cell registers are
// dead (61 VGPRs, still 8 waves per SIMD), so on every SIMD the scan phases of some waves fill the gaps of the exact
// phases of others — as a second kernel (or on a second stream) the two phases only ran one after the other.
// ---------------------------------------------------------------------------------------------------
template <int H>
__device__ __forceinline__ void cull_exact(const RayRec* __restrict__ rays, const RawTri* __restrict__ rtab0, const RawTri* __restrict__ rtab1,
                                           const uint2* lq, uint32_t lcap, const uint2* qw, uint32_t n,
                                           uint32_t gid /* per lane: ray id of run position `lane` */, uint32_t lane, uint32_t* bk) {
    // entry i of the wave's queue: the first lcap in LDS, the rest in its global region
    // (each side through a pointer of its own address space: one generic pointer selected between the two makes the load a FLAT one, which
    // counts on both wait counters — the compiler then drains every load in flight before it)
    typedef uint32_t U2 __attribute__((ext_vector_type(2)));
    auto entry = [&](uint32_t i) {
        U2 e;
        if (i < lcap) e = ((const U2 __attribute__((address_space(3)))*)lq)[i];
        else e = ((const U2 __attribute__((address_space(1)))*)qw)[i - lcap];
        return make_uint2(e.x, e.y);
    };
    // The queue entries are read one round ahead: a round then waits for ONE memory round trip (the gathers its entries
    // address: two triangle records and the ray, 72 bytes per lane), not two.
    if (n == 0u) return;
    uint2 en_next = entry(min(lane, n - 1u));
    for (uint32_t base = 0; base < n; base += 64u) {                // wave-uniform
        const bool live = base + lane < n;
        const uint2 en = en_next;
        const uint32_t map = en.x >> 31, pos = en.y >> 26;
        const uint32_t id0 = en.x & CULL_NOID, id1 = en.y & CULL_NOID;
        const RawTri* rt = map ? rtab1 : rtab0;
        const RawTri r0 = rt[id0 == CULL_NOID ? 0u : id0], r1 = rt[id1 == CULL_NOID ? 0u : id1];
        const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos << 2), (int)gid);
        const float4* rp = reinterpret_cast<const float4*>(rays + g);
        const float4 ra = rp[0], rb = rp[1];
        if (base + 64u < n) en_next = entry(min(base + 64u + lane, n - 1u));
        float best;
        if (H) {
            // the reference's as-shipped fp16 arithmetic (cast_pairs_h: what raycast_binned_h_kernel runs on every triangle); the
            // ray record holds fp16 values widened to f32 (prep_rays_kernel, precision 2), so the casts below are exact
            h2 v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const half2v x0 = __builtin_bit_cast(half2v, r0.d[q >> 1]), x1 = __builtin_bit_cast(half2v, r1.d[q >> 1]);
                v[q] = (q & 1) ? h2{x0.y, x1.y} : h2{x0.x, x1.x};
            }
            const _Float16 hnan = (_Float16)__builtin_nanf("");
            v[6] = h2{id0 == CULL_NOID ? hnan : v[6].x, id1 == CULL_NOID ? hnan : v[6].y};
            CellRegsH<1> t;
            set_pair_h(t, 0, v);
            const uint64_t none[1][2] = {{0, 0}};
            const _Float16 hsx = (_Float16)ra.x, hsy = (_Float16)ra.y, hsz = (_Float16)ra.z;
            const _Float16 hdx = (_Float16)rb.x, hdy = (_Float16)rb.y, hdz = (_Float16)rb.z;
            best = cast_pairs_h<1>(t, h2{hsx, hsx}, h2{hsy, hsy}, h2{hsz, hsz}, h2{hdx, hdx}, h2{hdy, hdy}, h2{hdz, hdz}, none, 0u);
        } else {
            const float qnan = __builtin_nanf("");
            f2 v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const f2 x0 = cvt2(r0.d[q >> 1]), x1 = cvt2(r1.d[q >> 1]);
                v[q] = (q & 1) ? f2{x0.y, x1.y} : f2{x0.x, x1.x};
            }
            // an empty slot next to a candidate: a NaN vertex a fails every test (as the NaN padding of the re-packed blocks does)
            v[6] = f2{id0 == CULL_NOID ? qnan : v[6].x, id1 == CULL_NOID ? qnan : v[6].y};
            CellRegs<1> t;
            set_pair(t, 0, v);
            const uint64_t none[1][2] = {{0, 0}};
            best = cast_pairs<1>(t, f2{ra.x, ra.x}, f2{ra.y, ra.y}, f2{ra.z, ra.z}, f2{rb.x, rb.x}, f2{rb.y, rb.y},
                                 f2{rb.z, rb.z}, none, 0u);
        }
        // the ray's running min lives in LDS as an ordered-u32 key: only lanes that HIT something take part (a ray meets 1-3 of
        // its ~16 candidates), so the atomic sees a handful of lanes — cheaper than a segmented wave min over all 64 first
        const uint32_t k = fkey(best);
        if (live && k < fkey(RAY_MISS)) atomicMin(bk + pos, k);
    }
}

// Packed f32 operations whose scalar operand is ONE float of an aligned SGPR pair, splat to both halves by op_sel.  The
// ray record arrives by s_load_dwordx8 as four such pairs; left to the compiler every high-half splat (sy, dy) becomes two s_mov
// building a new pair — 12 scalar instructions per ray in a loop that is bound by its total instruction issue.
typedef unsigned long long sgpr2;        // two floats in an aligned SGPR pair
__device__ __forceinline__ sgpr2 sgpr_pair(float lo, float hi) {
    return (sgpr2)__float_as_uint(lo) | ((sgpr2)__float_as_uint(hi) << 32);
}
template <int HI> __device__ __forceinline__ f2 pk_rsub(sgpr2 s, f2 v) {        // {s, s} - v
    f2 r;
    if (HI) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(s), "v"(v));
    else    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(s), "v"(v));
    return r;
}
template <int HI> __device__ __forceinline__ f2 pk_mul_s(f2 v, sgpr2 s) {        // v * {s, s}
    f2 r;
    if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(v), "s"(s));
    else    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(v), "s"(s));
    return r;
}
template <int HI> __device__ __forceinline__ f2 pk_fma_s(f2 a, sgpr2 s, f2 c) {  // a * {s, s} + c
    f2 r;
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "s"(s), "v"(c));
    else    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(s), "v"(c));
    return r;
}

// ---------------------------------------------------------------------------------------------------
// PHASE 1 — cull_scan_kernel.  One wave walks a run of <= 64 sorted rays.  Per (map, cell) bin it gathers the bin's 200
// bounding-sphere / normal records (4 per lane, as 2 packed pairs); per ray it runs tests (A), (B) on all of them and
// appends one 8-byte entry per lane-pair that holds a candidate to the wave's own region of the global candidate queue:
//     entry = { id0 | map << 31,  id1 | position in the run << 26 }          (CULL_NOID = empty slot next to a candidate)
// A wave owns a region of CULL_QCAP entries for the launch (nothing is allocated on the device: one returning atomic per wave on a
// shared counter cost 2.7 ms).  A ray adds at most 128 entries; a wave that has more than CULL_QCAP - 128 after a ray runs phase 2
// on what it has and scans on from the next ray (a "segment"), so the region cannot overflow whatever the mesh.
// Nothing here is heavy in registers or LDS, so 7-8 waves per SIMD hide the latencies of the id rows (HBM, streamed
// through LDS CULL_RING bins ahead) and of the record gathers (L2).
// ---------------------------------------------------------------------------------------------------
// Waves per hardware workgroup.  The kernel has no workgroup-level synchronisation, and a wave's lifetime depends on where its rays are:
// with four waves in a workgroup the slots of the three that finish first idle until the fourth is done (a new workgroup needs a free
// slot on every SIMD) — the resident waves averaged 5.0 per SIMD of the 6 the registers allow.  One wave per workgroup: ray cast
// 0.504 -> 0.487 ms at 65 536 envs, 0.315 -> 0.302 at 32 768 (4 / CULL_WPB consecutive workgroups of an XCD form one block slot of the
// XCD-aware order below; round 2 measured 2 against 4 and found no difference — at 8 waves per SIMD and twice the work per ray).
#define CULL_WPB 1
#define CULL_QGLOBAL 640u            // entries of a wave's GLOBAL queue region: what of its CULL_QCAP entries cannot be in LDS (the eager kernels keep 384 there)
#define CULL_QCAP 1024u              // entries of a wave's queue (LDS part + global region).  A ray adds at most 128, so a wave that finds more than
                                     // CULL_QCAP - 128 entries after a ray finishes (exact phase) what it has and scans on from the next ray
#define CULL_SCAN_ARGS                                                                                                          \
    const RayRec *__restrict__ rays, const uint32_t *__restrict__ sorted, uint32_t n_sorted, const int4 *__restrict__ idx0,     \
        const int4 *__restrict__ idx1, const uint4 *__restrict__ ctab0, const uint4 *__restrict__ ctab1,                         \
        uint32_t kp01 /* K8 of map 0 | K8 of map 1 << 16 */, uint32_t run, uint32_t n_blocks, uint32_t split, uint32_t t8, uint32_t r8, uint32_t chsr /* chs | chr << 8 */, uint32_t run_r, uint2 *__restrict__ queue,                                      \
        const RawTri *__restrict__ rtab0, const RawTri *__restrict__ rtab1, float *__restrict__ out, uint4 *__restrict__ stats, uint32_t j0, float c_a_h, float tau2_h, const float4 *__restrict__ far0, const float4 *__restrict__ far1, float k2_far, const float4 *__restrict__ near0, const float4 *__restrict__ near1

// LAZY: the far pairs of a bin (slot 1) are gathered and unpacked only if one of its rays tests them — a second, dependent round of
// gathers in the bins that do, half the set-up in the bins that do not (most of them when a bin holds few rays).
template <int H, int LAZY, int SKIPT>
__global__ void __launch_bounds__(64 * CULL_WPB) cull_scan_kernel(CULL_SCAN_ARGS) {
    const float k_ca = H ? c_a_h : CullK<0>::c_a, k_tau2 = H ? tau2_h : CullK<0>::tau2;      // (f32 proof: compile-time constants)
    // The id rows of a run's bins travel HBM -> LDS CULL_RING bins ahead of their use (global_load_lds: no registers, one
    // exposed memory latency per run instead of one per bin); s_bk: the run's 64 running minima as ordered-u32 keys.
    __shared__ int4 s_ids[CULL_WPB][CULL_RING][64];
    __shared__ uint32_t s_bk[CULL_WPB][64];
    // The first LCAP entries of the wave's candidate queue live in LDS (a run of 64 terrain rays queues ~230, at most ~600): phase 2 reads
    // them back without a round trip through memory, and phase 1 does not wait for the acknowledgement of 8-byte stores scattered over
    // 120 MB.  What does not fit goes to the wave's global region as before.  (Sized so that the LDS never caps the waves the registers
    // allow: 6 per SIMD x 4.25 + 4 KB, 7 x 4.25 + 3 KB.)
    constexpr uint32_t LCAP = LAZY ? 512u : 384u;
    static_assert(CULL_QCAP - LCAP <= CULL_QGLOBAL, "the global region holds what the LDS part does not");
    __shared__ uint2 s_lq[CULL_WPB][LCAP];
    const uint32_t x = blockIdx.x & 7u, tw = CULL_WPB == 1 ? 0u : threadIdx.x >> 6, lane = threadIdx.x & 63u;      // (one wave per workgroup: tw is a constant, and what derives from it — the queue region's address — is wave-uniform for the compiler too)
    // this wave: wave w (0..3) of block slot jslot of XCD x
    const uint32_t qx = blockIdx.x >> 3, w = (qx % (4u / CULL_WPB)) * CULL_WPB + tw, jslot = qx / (4u / CULL_WPB);
    uint32_t* const bk = s_bk[tw];
    uint2* const lq = s_lq[tw];
    // the wave's region of the candidate queue (CULL_QCAP entries): by its position in THIS launch — a step whose regions would
    // exceed the queue budget is cast in several launches over slices [j0, j0 + n) of the block slots (n x 8 XCDs x 4 / CULL_WPB workgroups), which re-use them
    uint2* const qw = queue + (size_t)((jslot * 8u + x) * 4u + w) * CULL_QGLOBAL;

    // XCD-aware order (blocks b, b + 8, ... run on one XCD): the TERRAIN blocks [0, split) are dealt to the XCDs in chunks of
    // 2^chs consecutive blocks, round robin, then the ROCKS blocks [split, n_blocks) the same way (chunks of 2^chr).  Chunks, so that neighbouring bins
    // (which share most of their triangles' ctab / rtab records) stay on one L2; round robin, because the cost of a ray depends on where
    // it is — with one contiguous eighth of each map per XCD the slowest XCD finished 27 % after the fastest (per-XCD end times of a
    // diagnostic build) and the launch takes as long as the slowest; terrain before rocks on every XCD, so that the cheap rays
    // (0.7 candidate pairs against 8) are the ones that drain at the end.
    // (One workgroup per block slot, started by the hardware as others finish.  Measured against it, round 3: a grid of just the
    //  resident workgroups, each walking the slots j, j + stride, ... of its XCD with a queue region of its own — 56 MB of queue for
    //  any batch — took 0.69 - 0.82 ms instead of 0.58: the cost of a block depends on where its rays are, and the slowest of 1 792
    //  static sums of ~12 blocks ends a third after the mean; the hardware's dynamic order ends one block after it.)
    const uint32_t j = jslot + j0;
    {
    uint32_t lb;
    // (chs / chr = 31: one contiguous eighth per XCD — small batches, where a chunk would be too few bins to share anything)
    const uint32_t chs = chsr & 0xffu, chr = chsr >> 8;
    if (j < t8) {
        lb = chs == 31u ? x * t8 + j : ((((j >> chs) << 3) + x) << chs) + (j & ((1u << chs) - 1u));
        if (lb >= split) return;
    } else {
        const uint32_t jr = j - t8;
        lb = split + (chr == 31u ? x * r8 + jr : ((((jr >> chr) << 3) + x) << chr) + (jr & ((1u << chr) - 1u)));
        if (lb >= n_blocks) return;
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(lb * 4u + w);
    // blocks [0, split) walk runs of `run` rays, the blocks behind them (the rocks part) runs of `run_r`
    const uint32_t my_run = lb < split ? run : run_r;
    const uint32_t i0 = lb < split ? wave * run : split * 4u * run + (wave - split * 4u) * run_r;
    if (i0 >= n_sorted) return;
    const uint32_t n_run = min(my_run, n_sorted - i0);               // <= 64
    // The run's ray ids, one per lane.  The rays' parameters come by scalar loads (s_load_dwordx8 through the ray id of lane r),
    // requested one ray ahead so that their latency passes under the previous ray's arithmetic.
    const uint32_t gid = sorted[i0 + (lane < n_run ? lane : n_run - 1u)];
    wave_lds_sync();
    bk[lane] = fkey(RAY_MISS);                 // 11.0 where a ray has no candidate at all (a culled triangle contributes exactly that)
    uint32_t ctot = 0, n_both = 0, n_bins = 0, n_fskip = 0, n_askip = 0; // queue entries / rays that ran both tests / bins walked / rays that skipped the far pairs / rays not scanned at all (rover_get_cull_info)
    uint32_t r_next = 0;                       // first ray of the run that is not scanned yet
    // One SEGMENT = scan rays [r_next, ...) until the run ends or the queue region could overflow with one more ray, then the exact
    // arithmetic on the entries.  Nearly every run is one segment (a run of 64 terrain rays queues ~500 entries).  Everything a
    // segment needs besides the ray ids is derived here, inside the loop: nothing but `gid` stays live across the exact phase,
    // which is the kernel's register high-water mark.
    while (r_next < n_run) {
    // The run's (map, cell) keys, one per lane, in ONE round of loads: the wave then knows its bins and can request their
    // id rows ahead.
    // (through an opaque copy of the ray id: everything below is invariant across segments, and hoisted out of this loop by the
    //  compiler it would stay in registers through the scan AND the exact phase — 76 VGPRs instead of 62)
    uint32_t gid_s = gid;
    asm volatile("" : "+v"(gid_s));
    const float4 rsa = reinterpret_cast<const float4*>(rays + gid_s)[0], rsb = reinterpret_cast<const float4*>(rays + gid_s)[1];
    const uint32_t rflags = __float_as_uint(rsb.w);
    const uint32_t key = __float_as_uint(rsa.w) | (rflags << 31);                      // cell | map << 31
    const uint32_t prev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ? lane - 1u : 0u) << 2), (int)key);
    // bit i: ray i starts a new (map, cell) bin
    const uint64_t heads = __builtin_amdgcn_ballot_w64(lane < n_run && (lane == 0u || key != prev));
    // Per ray, for all 64 at once: the byte offset of its bin's id row in its map's table with the map in bit 0 (rows are
    // multiples of 16 bytes; tables stay below 4 GB, rover_set_knn_map checks) — a bin then costs one v_readlane for its row, and
    // this register and the ray ids are all the per-run state a lane carries through the scan.
    const uint32_t kmap = key >> 31, kcell = key & 0x7fffffffu;
    const uint32_t rowm = (kcell * (((kmap ? kp01 >> 16 : kp01) & 0xffffu) >> 2) * 16u) | kmap;
    // bit i: ray i clears every FAR triangle of its cell (slot 1 of every lane) at once — far_build_kernel has the derivation —
    // so on the cone path its scan skips slot 1 altogether; each lane decides for its own ray
    // The cell's record (far_build_kernel): the bound of its far pairs and its normal cone, ONE gather per lane.
    // conemask bit i: the cone of ray i's cell covers the ray (the ray's own bound: flags bits 16..31) — test (B) holds for every triangle
    // of the cell, the scan runs test (A) only.  farskip bit i: ray i clears every FAR triangle of its cell (slot 1 of every lane) at
    // once, so on the cone path its scan skips slot 1 altogether (the fp16 proof's kernels: the SKIP ones only — with that proof's c_a the
    // bound holds for 39 % of the rays instead of 84 %).
    const float4* fr = (kmap ? far1 : far0) + 2ull * kcell;
    const float4 fb = fr[1];                                                  // {Cx, Cy, q16, -}
    const uint64_t conemask = __builtin_amdgcn_ballot_w64(__float_as_uint(fb.z) >= (rflags >> 16));
    if (r_next == 0u) n_both += (uint32_t)__builtin_popcountll(~conemask & (n_run >= 64u ? ~0ull : ((1ull << n_run) - 1ull)));
    // skipall bit i: ray i clears the NEAR pairs as well, on the cone path: no candidate in this cell, not scanned.  SKIP kernels only: the
    // host picks them where rays qualify (regular meshes, rock rays a tenth of the set or more: a quarter of the rays with 37 + 26, 12 %
    // with 120 + 26); on the irregular mesh 1 % qualify and dropping dead bins / walking live rays costs 2 % (6 % before the per-lane word).
    constexpr bool SKIP = SKIPT != 0;
    uint64_t farskip = 0, skipall = 0;
    if (!H || SKIP) {
        const float4 fa = fr[0];                                              // {G, z0, z1, rho_out} of the far pairs
        // (hardware square roots, 1 ulp, and no division — the inequality is multiplied through by |d_z| — the margins are 1e-4)
        const float ox = rsa.x - fb.x, oy = rsa.y - fb.y, o = __builtin_amdgcn_sqrtf(ox * ox + oy * oy);
        const float dzm = fmaxf(fabsf(rsa.z - fa.y), fabsf(rsa.z - fa.z));
        const float dxy2 = rsb.x * rsb.x + rsb.y * rsb.y, adz = fabsf(rsb.z);
        const bool steep = adz * adz >= 0.81f * (dxy2 + adz * adz) * 1.0001f;  // cos(beta) >= 0.9
        const float e_adz = o * adz + dzm * __builtin_amdgcn_sqrtf(dxy2) * 1.0001f;                      // e_max |d_z|
        // |a| = |h . d| / |d| <= |h_z| |d_z| + |h_xy| |d_xy| (tighter than |h| for the triangles far out: the ray is steep)
        const float dxy1 = __builtin_amdgcn_sqrtf(dxy2) * 1.0001f;
        const float amax = (dzm * adz + (o + fa.w) * dxy1) * 1.0001f;
        farskip = __builtin_amdgcn_ballot_w64(steep && (fa.x * 0.9999f - 1.0e-5f) * adz > (e_adz + k2_far * amax * adz) * 1.0001f);
        if (SKIP) {     // the same inequality for the near pairs, with their own G, z range and largest distance
            const float4 fn = (kmap ? near1 : near0)[kcell];
            const float dzm_n = fmaxf(fabsf(rsa.z - fn.y), fabsf(rsa.z - fn.z));
            const float e_adz_n = o * adz + dzm_n * __builtin_amdgcn_sqrtf(dxy2) * 1.0001f;
            const float amax_n = (dzm_n * adz + (o + fn.w) * dxy1) * 1.0001f;
            skipall = farskip & conemask &
                      __builtin_amdgcn_ballot_w64(steep && (fn.x * 0.9999f - 1.0e-5f) * adz > (e_adz_n + k2_far * amax_n * adz) * 1.0001f);
        }
    }
    const uint64_t runmask = n_run >= 64u ? ~0ull : ((1ull << n_run) - 1ull);
    if (r_next == 0u) { n_fskip += (uint32_t)__builtin_popcountll(farskip & conemask & runmask); n_askip += (uint32_t)__builtin_popcountll(skipall & runmask); }
    // the rays of the segment that are scanned at all, and its bins: the segment's first ray opens one; bins without a live ray are
    // dropped here — no id row is requested for them, no record gathered
    const uint64_t live = ~skipall & runmask & (~0ull << r_next);
    // Per run position, in ONE register, what the ray loop needs when it reaches ray r — the id of the NEXT live ray (whose record it
    // requests) in bits 0..29, "ray r is on the cone path" in bit 31, "ray r skips the far pairs" in bit 30: one v_readlane per ray instead
    // of a dozen scalar instructions on the 64-bit masks (SKIP: the next LIVE ray)
    uint32_t nxw;
    {
        uint32_t nxt = lane + 1u < n_run ? lane + 1u : lane;               // (every ray is live without SKIP)
        if (SKIP) {
            const uint64_t above = live & (~1ull << lane);                 // (per lane)
            nxt = above ? (uint32_t)__builtin_ctzll(above) : lane;
        }
        const uint32_t gnx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nxt << 2), (int)gid);
        nxw = (gnx & 0x3fffffffu) | ((uint32_t)((conemask >> lane) & 1ull) << 31) | ((uint32_t)((farskip >> lane) & 1ull) << 30);
    }
    const uint64_t hm_all = (heads | (1ull << r_next)) & (~0ull << r_next);
    uint64_t hm = SKIP ? 0ull : hm_all;
    for (uint64_t tt = SKIP ? hm_all : 0ull; tt;) {
        const uint32_t b0 = (uint32_t)__builtin_ctzll(tt);
        tt &= tt - 1ull;
        const uint32_t b1 = tt ? (uint32_t)__builtin_ctzll(tt) : n_run;
        const uint64_t bm = (b1 >= 64u ? ~0ull : ((1ull << b1) - 1ull)) & (~0ull << b0);
        if (live & bm) hm |= 1ull << b0;
    }
    n_bins += (uint32_t)__builtin_popcountll(hm);
    uint64_t pf_heads = hm;                    // bins whose row is not requested yet
    uint32_t pf_n = 0, use_n = 0;              // rows requested / consumed so far (slot = count % CULL_RING)
    // this lane's slot in an id row of map m (lanes past K8 / 4 repeat the last one: a duplicate candidate cannot change a min), as a byte offset
    auto lane_off = [&](uint32_t m) { return min(lane, (((m ? kp01 >> 16 : kp01) & 0xffffu) >> 2) - 1u) << 4; };
    auto prefetch_row = [&]() {
        const uint32_t jj = (uint32_t)__builtin_ctzll(pf_heads);
        pf_heads &= pf_heads - 1ull;
        const uint32_t rm = (uint32_t)__builtin_amdgcn_readlane((int)rowm, (int)jj), m2 = rm & 1u;
        const char* src = reinterpret_cast<const char*>(m2 ? idx1 : idx0) + (rm & ~15u) + lane_off(m2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)&s_ids[tw][pf_n % CULL_RING][0], 16, 0, 0);
        ++pf_n;
    };
#pragma unroll 1
    for (int d = 0; d < CULL_RING; ++d)
        if (pf_heads) prefetch_row();
    uint32_t cused = 0;
    auto load_ray = [&](uint32_t r, float4& a4, float4& b4) {       // wave-uniform address -> scalar loads
        const float4* rp = reinterpret_cast<const float4*>(rays + (uint32_t)__builtin_amdgcn_readlane((int)gid, (int)r));
        a4 = rp[0]; b4 = rp[1];
    };
    float4 nxa, nxb;
    load_ray(live ? (uint32_t)__builtin_ctzll(live) : r_next, nxa, nxb);       // the segment's first live ray
    bool full = false;
    while (hm && !full) {                      // one (map, cell) bin of the run: rays [i, i_end)
        const uint32_t i = (uint32_t)__builtin_ctzll(hm);
        hm &= hm - 1ull;
        uint32_t i_end;
        if (SKIP) {                                                         // (the bin ends at the next head of the run, live or not)
            const uint64_t heads_above = hm_all & (~1ull << i);
            i_end = heads_above ? (uint32_t)__builtin_ctzll(heads_above) : n_run;
        } else {
            i_end = hm ? (uint32_t)__builtin_ctzll(hm) : n_run;
        }
        const uint32_t map = (uint32_t)__builtin_amdgcn_readlane((int)rowm, (int)i) & 1u;
        // the bin's id row, requested CULL_RING bins ago (8 waves per SIMD cover what is left of its latency)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();
        const int4 id4 = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(&s_ids[tw][use_n % CULL_RING][0]) + lane_off(map));
        ++use_n;
        const int32_t id[4] = {id4.x, id4.y, id4.z, id4.w};
        const uint4* ct = map ? ctab1 : ctab0;
        // does any ray of this bin test the FAR pairs (slot 1)?  (off the cone path always, on it unless its far-skip bit is set)
        const uint64_t binmask = (i_end >= 64u ? ~0ull : ((1ull << i_end) - 1ull)) & (~0ull << i);
        uint64_t todo = SKIP ? live & binmask : binmask;                    // the bin's rays that are scanned (at least one)
        const bool need1 = !LAZY || (~(farskip & conemask) & todo) != 0ull;
        uint4 rec[4];
#pragma unroll
        for (int jj = 0; jj < (LAZY ? 2 : 4); ++jj) rec[jj] = ct[id[jj] < 0 ? 0 : id[jj]];
        CullRegs t;
        uint32_t qid[2][2];                    // the lane's ids as queue-entry fields
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (LAZY && p == 1) {
                if (!need1) break;                                  // (nothing below reads slot 1 then: every ray of the bin skips it)
                rec[2] = ct[id[2] < 0 ? 0 : id[2]]; rec[3] = ct[id[3] < 0 ? 0 : id[3]];
            }
            const uint4 a = rec[2 * p], b = rec[2 * p + 1];
            const f2 za = cvt2(a.z), zb = cvt2(b.z), wa = cvt2(a.w), wb = cvt2(b.w);
            t.mx[p] = f2{__uint_as_float(a.x), __uint_as_float(b.x)};
            t.my[p] = f2{__uint_as_float(a.y), __uint_as_float(b.y)};
            t.mz[p] = f2{za.x, zb.x};
            t.nx[p] = f2{za.y, zb.y}; t.ny[p] = f2{wa.x, wb.x}; t.nz[p] = f2{wa.y, wb.y};
            f2 s = t.nx[p] * t.nx[p];
            s = fma2(t.ny[p], t.ny[p], s);
            s = fma2(t.nz[p], t.nz[p], s);
            s = s * f2{k_tau2, k_tau2};                        // = cull_r2()
            // an empty slot is never a candidate: r2 = -inf (its centre / normal are triangle 0's, finite)
            const bool e0 = id[2 * p] >= 0, e1 = id[2 * p + 1] >= 0;
            t.r2[p] = f2{e0 ? s.x : -__builtin_inff(), e1 ? s.y : -__builtin_inff()};
            if (H) {    // the fp16 proof: (B)'s threshold is per triangle, r2 F with F = 1 + code / 512 from the centre's low mantissa bits
                const f2 code = f2{(float)((a.x & 63u) | ((a.y & 63u) << 6)), (float)((b.x & 63u) | ((b.y & 63u) << 6))};
                t.r2b[p] = t.r2[p] * fma2(code, f2{1.0f / 512.0f, 1.0f / 512.0f}, f2{1.0f, 1.0f});
            }
            qid[p][0] = (e0 ? (uint32_t)id[2 * p] : CULL_NOID) | (map << 31);
            qid[p][1] = e1 ? (uint32_t)id[2 * p + 1] : CULL_NOID;
        }
        // The next row request goes out AFTER the records above were unpacked: the compiler waits for them with s_waitcnt vmcnt(0)
        // (vector memory operations retire in order), and issued before the gathers' data are used, the request — an HBM round trip —
        // would be waited for right there, in every bin.  Here nothing waits for it before the next bin's start.
        // (an empty asm that consumes what the unpacking produced pins the order: the compiler otherwise sinks the unpacking below the request)
        asm volatile("" :: "v"(t.mx[0]), "v"(t.mx[1]), "v"(t.nz[0]), "v"(t.nz[1]), "v"(t.r2[0]), "v"(t.r2[1]) : "memory");
        if (pf_heads) { wave_lds_sync(); prefetch_row(); }              // into the slot just read (the ids are in registers)
        uint32_t r = i;
        for (;;) {
            // the next ray of the bin: the next live one through the mask (SKIP), or simply the next
            if (SKIP) {
                if (!todo) break;
                r = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
            } else if (r >= i_end) break;
            // tests (A), (B): lanes whose pair p holds a triangle that they do not both reject
            const float4 ra = nxa, rb = nxb;
            const uint32_t xr = (uint32_t)__builtin_amdgcn_readlane((int)nxw, (int)r);
            {
                const float4* rp = reinterpret_cast<const float4*>(rays + (xr & 0x3fffffffu));      // the next (live) ray, of this bin or a later one
                nxa = rp[0]; nxb = rp[1];
            }
            const sgpr2 sxy = sgpr_pair(ra.x, ra.y), szc = sgpr_pair(ra.z, ra.w), dxy = sgpr_pair(rb.x, rb.y), dzf = sgpr_pair(rb.z, rb.w);
            uint64_t any[2];
            const bool cone = (xr >> 31) != 0u;                                          // (B) holds for every triangle of the cell
            if (cone) {
                auto test_a = [&](int p) {                                                  // (A): c_a |h|^2 - (h.d)^2 > r2
                    const f2 hx = pk_rsub<0>(sxy, t.mx[p]), hy = pk_rsub<1>(sxy, t.my[p]), hz = pk_rsub<0>(szc, t.mz[p]);
                    const f2 hd = pk_fma_s<0>(hz, dzf, pk_fma_s<1>(hy, dxy, pk_mul_s<0>(hx, dxy)));
                    f2 hh = hx * hx; hh = fma2(hy, hy, hh); hh = fma2(hz, hz, hh);
                    const f2 A = fma2(hh, f2{k_ca, k_ca}, -(hd * hd));
                    return ~(__builtin_amdgcn_ballot_w64(A.x > t.r2[p].x) & __builtin_amdgcn_ballot_w64(A.y > t.r2[p].y));
                };
                any[0] = test_a(0);
                any[1] = (xr & 0x40000000u) ? 0ull : test_a(1);                               // every far triangle cleared: slot 1 skipped
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f2 hx = pk_rsub<0>(sxy, t.mx[p]), hy = pk_rsub<1>(sxy, t.my[p]), hz = pk_rsub<0>(szc, t.mz[p]);
                    const f2 hd = pk_fma_s<0>(hz, dzf, pk_fma_s<1>(hy, dxy, pk_mul_s<0>(hx, dxy)));
                    f2 hh = hx * hx; hh = fma2(hy, hy, hh); hh = fma2(hz, hz, hh);
                    const f2 A = fma2(hh, f2{k_ca, k_ca}, -(hd * hd));                   // (A): c_a |h|^2 - (h.d)^2 > r2
                    const f2 Dn = pk_fma_s<0>(t.nz[p], dzf, pk_fma_s<1>(t.ny[p], dxy, pk_mul_s<0>(t.nx[p], dxy)));
                    const f2 B = Dn * Dn;                                               // (B): (n.d)^2 > tau^2 |n|^2 = r2
                    // one ballot per compare (each stays a v_cmp writing an SGPR pair); NaN compares false = stays a candidate
                    const f2 rb2 = H ? t.r2b[p] : t.r2[p];
                    const uint64_t rej0 = __builtin_amdgcn_ballot_w64(A.x > t.r2[p].x) & __builtin_amdgcn_ballot_w64(B.x > rb2.x);
                    const uint64_t rej1 = __builtin_amdgcn_ballot_w64(A.y > t.r2[p].y) & __builtin_amdgcn_ballot_w64(B.y > rb2.y);
                    any[p] = ~(rej0 & rej1);                                            // (all 64 lanes are active here)
                }
            }
            if (any[0] | any[1]) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (any[p]) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(any[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any[p], 0u));
                        if (__builtin_amdgcn_inverse_ballot_w64(any[p])) {
                            const uint32_t at = cused + rank;
                            const uint2 e = make_uint2(qid[p][0], qid[p][1] | (r << 26));
                            if (at < LCAP) lq[at] = e; else qw[at - LCAP] = e;
                        }
                        cused += (uint32_t)__builtin_popcountll(any[p]);
                    }
                }
                // the next ray could add 128 more: finish what is queued first (rare: > 14 candidate pairs per ray over a whole run)
                if (cused > CULL_QCAP - 128u) { full = true; if (!SKIP) ++r; break; }
            }
            if (!SKIP) ++r;
        }
        r_next = SKIP ? r + 1u : r;            // (a full queue: the next segment starts behind the last ray scanned)
    }
    if (!full) r_next = n_run;                 // every live ray of the run is scanned (what is left, if anything, clears its whole cell)
    // PHASE 2 on the wave's own entries.  They were stored to global memory by this wave and are read back by this wave, and no
    // other wave ever touches the region: a wave sees its own stores in program order, so waiting for their acknowledgement is all
    // the ordering that takes.  An agent-scope fence here writes back the whole L2 of the XCD: 3.7 ms instead of 0.6.
    // The wait also retires id-row loads of bins this segment did not reach (the ring restarts with the next segment).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_lds_sync();
    cull_exact<H>(rays, rtab0, rtab1, lq, LCAP, qw, cused, gid, lane, bk);
    ctot += cused;
    }
    wave_lds_sync();
    if (lane < n_run) out[gid] = funkey(bk[lane]);
    // per-wave counters of THIS launch (plain stores, 16 B per wave; summed on the host by rover_get_cull_info)
    if (lane == 0u) stats[wave] = make_uint4(ctot, n_run | (n_fskip << 8), n_both | (n_askip << 8), n_bins);
    }
}

// ---------------------------------------------------------------------------------------------------
// The STAGED ray cast (raycast variant 4): the culled ray cast with lanes and rays exchanged in phase 1.
//
// cull_scan_kernel gives a bin's triangles to the lanes and walks the bin's rays one by one: per ray ~25 wave instructions that test
// nothing (ray parameters, ballots, queue bookkeeping), per bin a gather round trip (id row -> records) and an unpacking of 200 records
// of which a steep ray can meet a few dozen.  Here a wave still takes a run of <= 64 rays (sorted by bin, or 64 slots in env order), but
//   * a cell's pairs are stored as a ROW ready to test — 16 B per pair: two sphere records {centre relative to the cell's own centre,
//     r2} in fp16 — in the order of the group-bound key G (far_build_kernel's), with a bound per SUFFIX of the row every LN_CH = 8 pairs
//     (16 "levels", lane_build_kernel): a ray evaluates the far skip's inequality against the sixteen suffixes once; the first one it
//     clears — and whose normal cone covers it — is its level L, and only the first L chunks of 8 pairs (one 128-byte line each) are tested;
//   * the tests are dealt as ITEMS = (ray, chunk) to the lanes, 64 items per round whatever ray or bin they belong to, ordered so that
//     lanes reading the same line sit side by side: a lane reads its chunk's 8 records straight from the row (L1 / L2; the first
//     version staged them HBM -> LDS: slower), runs test (A) on them — 11 plain instructions per triangle, the fp16 record read by
//     v_fma_mix_f32 without a conversion — and leaves an 8-bit candidate mask: no per-ray overhead, no bin set-up, and a wave's work
//     is the sum of its rays' needs, not 64 times the largest;
//   * a ray its cell's normal cone does not cover (test (B) is not implied: body rays, steep ground, irregular meshes) runs tests (A)
//     and (B) on its prefix, the (B) records {n, r2B} from the row's second half; steep pairs are ordered in FRONT of their row so that
//     the suffixes keep a cone (LN_QGOOD); a wild ray (non-finite, |origin - cell| >= 1e4) takes every pair of its cell as a candidate;
//   * candidates become 2-byte queue entries (ray position, pair position) in LDS; the exact phase (the same arithmetic as every other
//     ray-cast kernel, one lane per entry, two rounds in flight) fetches the pair's triangle ids through the cell's id row.
// One launch serves both maps (a lane's map comes with its ray); on regular rocks meshes the host gives the rocks part of the sorted list
// to cull_scan_kernel instead (rover_capi.cpp, run_raycast).  Measured history: EXPERIMENTS.md 8.2, 8.2b.
//
// Test (A) on these records.  A pair record holds, per triangle, m' = fp16(m - C) (m: ctab's sphere centre, C = (cell centre, z_c) in
// f32) and r2' (fp16, rounded up) >= c_rho (rho~ + e + p)^2 (c_rho = 1.06, p = LN_PAD = 5e-5) with rho~ >= the radius of the padded triangle about m (from ctab's own r2)
// and e = |C + m' - m| the encoding's displacement, computed in double from the decoded values: the sphere about M = C + m' of radius
// rho' = rho~ + e contains the padded triangle.  A lane computes s' = fl(s - C), h^ = fl(s' - m') (each component within 2.4e-7 |h| +
// 5e-7 of the true h = s - M: two roundings of magnitudes <= |h| + |m'|, |m'| <= 4 m enforced by the builder), q = |h^|^2, t = h^ . d,
// u = fl(fl(c_a q - r2') - t^2) (c_a = 0.99925) and culls iff u >= +0.  Rounding of q, t, u moves the inequality by < 3e-6 |h^|^2 + 3e-7 r2', so
// u >= 0 gives W^ ^2 >= 0.000737 |h^|^2 + 1.05999 (rho' + p)^2 (|d|^2 within 1e-5 of 1) for the distance W^ from M to the line through M + h^; by
// Cauchy-Schwarz W^ >= 0.00524 |h^| + 1.0102 (rho' + p), and the true line is within |h - h^| <= 4.2e-7 |h| + 8.7e-7 of that one:
// W >= 0.00523 |h| + 1.0102 rho' + 1.01 p - 9e-7 > rho' + 0.005 (|h| + 2 rho') for every p >= 1e-6, which is all the rejection proof at the
// top of this file uses of test (A).  (p = 1e-3 cost a tenth more candidates: 3.95 pairs per ray against 3.6 with 5e-5.)
// Rays with a non-finite or far-away origin (|s'| >= 1e4: nothing overflows below that) are the wild rays: every pair a candidate.
// ---------------------------------------------------------------------------------------------------
#define LN_CH 8u                     // pairs per chunk: one 128-byte line of a cell's record row, one 8-bit candidate mask (with 16 pairs per chunk a
                                     // terrain ray of configs[2] tested 2.4 chunks = 38 pairs on average, with 8 it tests 4.3 chunks = 34)
#define LN_MAXCH 16u                 // chunks per row at most (K8 <= 256: 128 pairs)
#define LN_LVL 11u                   // float4 per cell of the level table: header {Cx, Cy, z_c, q16}, 16 levels x 8 B {G, z0, z1, rho_out} as fp16 (G, z0
                                     // rounded down, z1, rho_out up), the levels' 16 x 16-bit cones
// A pair with a triangle whose cone value (f32 proof: |N_z| / |N|; fp16 proof: the largest ray angle it admits, over pi / 2) is below this
// is ordered in FRONT of the others, so that the suffixes keep a cone rays lie in (0: off.  Grid mesh: 0 / 0.25 / 0.35 / 0.5 -> ray cast 440 /
// 440 / 440 / 494 us — at 0.5 half of the pairs of a bumpy heightfield are "steep" and every ray's prefix grows —; irregular mesh 811 / 559 / 560 us)
#define LN_QGOOD 0.3f
#define LN_QGOOD_H 0.2f              // (fp16 proof: 0.2 / 0.35 / 0.5 / 0.7 -> 477 / 506 / 536 / 545 us for the terrain part)
#define LN_QCAP 1024u                // 2-byte queue entries per wave
#define LN_PAD 5.0e-5                // added to a triangle's radius: what pays for the relative coordinates' rounding (header comment: >= 1e-6 would do)
#define LN_AB 8u                     // pairs an item keeps in flight: test (A) only (a record per pair) ...
#define LN_ABB 4u                    // ... tests (A) and (B) (two records per pair)
#ifndef LN_ABB4
#define LN_ABB4 4u                   // ... tests (A) and (B4) (f32 proof: 16 + 8 bytes per pair)
#endif
// Test (B) of the f32 proof on 4-byte records ("B4", the second half of a cell's row: 8 B per pair instead of 16).  A record holds the
// triangle's unit normal as three signed 10-bit integers n4 = round(511 n^) (bits 0-9, 10-19, 20-29), n^ the direction of the ctab record's
// fp16 normal (within 5e-4 of N / |N|: fp16 components of a vector of length r / tau >= 1); the builder decodes its own code and keeps it only
// if n4 / |n4| lies within LN_B4_ERR - 5e-4 of n^ and |n4| <= 512 — else, like slivers and overflows, the triangle is stored as n4 = 0, "always
// a candidate".  A lane computes t = fl(n4 . d) (|t| <= 512 |d|, absolute error < 1e-4) and culls iff fl(t^2 - LN_B4_C) >= +0 with
// LN_B4_C = (512 LN_B4_TAU)^2 (1 + 1e-4): then |n4 . d| >= LN_B4_TAU 512 (1 + 4e-5) >= LN_B4_TAU |n4|, i.e. the decoded unit normal — within
// LN_B4_ERR of N / |N| — has |cos| > LN_B4_TAU, which is test (B) of the header comment with (tau, 1e-3) replaced by (LN_B4_TAU, LN_B4_ERR):
// |N . d| / |N| > LN_B4_TAU - LN_B4_ERR.  n4 = 0 gives t = 0, u = -LN_B4_C < 0: a candidate.  (An empty slot decodes to a candidate too; its id
// is CULL_NOID and the exact phase returns a miss for it: the row's last chunk only.)  The band of directions a triangle stays a candidate
// for is 0.89 degrees about its plane (0.75 with the 8-byte records): for a horizontal ray over 200 triangles two exact evaluations by orientation instead of one and a half.
#define LN_B4_TAU 1.55e-2
#define LN_B4_ERR 3.0e-3
#define LN_B4_C ((float)((512.0 * LN_B4_TAU) * (512.0 * LN_B4_TAU) * 1.0001))
static_assert(0.999 * (LN_B4_TAU - LN_B4_ERR) * CullK<0>::sigma > 1.45 * 2.0 * 1.0e-6 / CullK<0>::alpha, "f32 cull proof on B4 records: (B) must contradict (A)");
#define LN_WAVES 5                   // waves per SIMD the kernel is compiled for (95 VGPRs, 7.4 KB LDS): 3 / 4 / 5 -> 317 / 253 / 235 us; 6 spills

__device__ __forceinline__ uint16_t half_bits_up(float v) {      // fp16 >= v (v >= 0, finite or +inf)
    _Float16 h = (_Float16)v;
    uint16_t b = __builtin_bit_cast(uint16_t, h);
    if ((float)h < v) b = (uint16_t)(b + 1u);                    // next fp16 above (0x7bff + 1 = +inf)
    return b;
}

// one workgroup of 128 threads per cell, thread = source pair (idx4's (lane, slot)); see the header comment for the record
__global__ void __launch_bounds__(128) lane_build_kernel(const int4* __restrict__ idx4, const uint4* __restrict__ ctab, uint32_t K8, uint32_t pp,
                                                         uint32_t Y, float cell_size, float shift_x, float shift_y, float k1, float tau2,
                                                         double c_rho, const uint32_t* __restrict__ qrow, const float* __restrict__ nz_abs,
                                                         float4* __restrict__ lvl, uint4* __restrict__ lrec, uint2* __restrict__ lid, int half,
                                                         float q_good) {
    __shared__ float s_z0[128], s_z1[128], s_g[128], s_ro[128], s_qn[128];
    __shared__ uint4 s_nrec[128];
    __shared__ uint32_t s_key[128];
    __shared__ uint8_t s_src[128];
    __shared__ uint4 s_rec[128];
    __shared__ uint2 s_id[128];
    const uint32_t cell = blockIdx.x, p = threadIdx.x, L = K8 >> 2, n_src = K8 >> 1;
    const float ccx = (float)(cell / Y) * cell_size + shift_x, ccy = (float)(cell % Y) * cell_size + shift_y;
    int32_t id[2] = {-1, -1};
    uint4 rec[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (p < n_src) {
        const int4 r4 = idx4[(uint64_t)cell * L + (p % L)];
        const uint32_t slot = p / L;
        id[0] = slot ? r4.z : r4.x; id[1] = slot ? r4.w : r4.y;
#pragma unroll
        for (int e = 0; e < 2; ++e) if (id[e] >= 0) rec[e] = ctab[id[e]];
    }
    // z reference of the cell: the middle of its centres' z range
    float zlo = __builtin_inff(), zhi = -__builtin_inff();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (id[e] < 0) continue;
        const float mz = cvt2(rec[e].z).x;
        if (mz == mz && fabsf(mz) < 6.0e4f) { zlo = fminf(zlo, mz); zhi = fmaxf(zhi, mz); }
    }
    s_z0[p] = zlo; s_z1[p] = zhi;
    __syncthreads();
    for (uint32_t sdt = 64u; sdt > 0; sdt >>= 1) {
        if (p < sdt) { s_z0[p] = fminf(s_z0[p], s_z0[p + sdt]); s_z1[p] = fmaxf(s_z1[p], s_z1[p + sdt]); }
        __syncthreads();
    }
    const float zc = s_z0[0] <= s_z1[0] ? 0.5f * (s_z0[0] + s_z1[0]) : 0.0f;
    __syncthreads();
    // the pair's record
    uint16_t hb[2][4], nb[2][4];
    uint32_t b4[2] = {0u, 0u};                                                 // f32 proof: test (B)'s 4-byte records (0 = always a candidate)
    float G = __builtin_inff(), pz0 = __builtin_inff(), pz1 = -__builtin_inff(), pro = 0.0f, pq = 2.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        hb[e][0] = hb[e][1] = hb[e][2] = 0; hb[e][3] = 0xfc00u;              // empty slot: r2 = -inf, never a candidate
        nb[e][0] = nb[e][1] = nb[e][2] = 0; nb[e][3] = 0xfc00u;              // (test (B): (n . d)^2 - r2B with r2B = -inf holds too)
        if (id[e] < 0) continue;
        const f2 zn = cvt2(rec[e].z), w = cvt2(rec[e].w);
        const float mx = __uint_as_float(rec[e].x), my = __uint_as_float(rec[e].y), mz = zn.x;
        const float r2c = cull_r2(zn.y, w.x, w.y, tau2);                      // as cull_scan_kernel derives it
        const float fx = (float)((double)mx - (double)ccx), fy = (float)((double)my - (double)ccy), fz = (float)((double)mz - (double)zc);
        bool ok = r2c == r2c && r2c < 3.0e38f && r2c >= 0.0f && fabsf(fx) <= 4.0f && fabsf(fy) <= 4.0f && fabsf(fz) <= 4.0f;    // (NaN compares false)
        const _Float16 hx = (_Float16)fx, hy = (_Float16)fy, hz = (_Float16)fz;
        float g = -__builtin_inff(), r2h = __builtin_inff();
        uint16_t r2b = 0x7c00u;                                                // +inf: always a candidate
        if (ok) {
            const double ex = (double)ccx + (double)(float)hx - (double)mx, ey = (double)ccy + (double)(float)hy - (double)my,
                         ez = (double)zc + (double)(float)hz - (double)mz;
            const double enc = sqrt(ex * ex + ey * ey + ez * ez);
            const double rho = sqrt((double)r2c / c_rho);                     // >= rho + 1e-4 of the ctab record (its r2 >= c_rho (rho + 1e-4)^2)
            const double need = c_rho * (rho + enc + LN_PAD) * (rho + enc + LN_PAD) * 1.000001;
            if (need < 6.0e4) {
                r2b = half_bits_up((float)(need * 1.0000001));
                if (r2b < 0x0400u) r2b = 0x0400u;                              // no fp16 denormals
                r2h = (float)__builtin_bit_cast(_Float16, r2b);
            }
        }
        // test (B)'s record: the stored normal as it is (fp16) and r2B >= tau^2 |n|^2, what cull_scan_kernel compares (n . d)^2 with, rounded up
        nb[e][3] = 0x7c00u;                                                    // always a candidate: (B) never holds
        if (r2b != 0x7c00u) {
            nb[e][0] = (uint16_t)(rec[e].z >> 16); nb[e][1] = (uint16_t)(rec[e].w & 0xffffu); nb[e][2] = (uint16_t)(rec[e].w >> 16);
            // (the fp16 proof's (B) threshold is per triangle: r2 F, F = 1 + code / 512 from the centre's low mantissa bits, as cull_scan_kernel decodes it)
            const float Fb = half ? 1.0f + (float)((rec[e].x & 63u) | ((rec[e].y & 63u) << 6)) * (1.0f / 512.0f) : 1.0f;
            nb[e][3] = half_bits_up(r2c * Fb * 1.000001f);
            pq = fminf(pq, nz_abs[id[e]]);                                     // |N_z| / |N| (rounded down) of the exact normal
            if (!half) {    // B4: the normal's direction as 3 x 10 bits, kept only if it decodes to within the allowance of what was encoded
                const double nx = (double)zn.y, ny = (double)w.x, nz = (double)w.y, nn = sqrt(nx * nx + ny * ny + nz * nz);
                if (nn > 0.0 && nn < 1.0e30) {
                    const int qx = (int)rint(nx / nn * 511.0), qy = (int)rint(ny / nn * 511.0), qz = (int)rint(nz / nn * 511.0);
                    const double ql = sqrt((double)(qx * qx + qy * qy + qz * qz));
                    const double ex = qx / ql - nx / nn, ey = qy / ql - ny / nn, ez = qz / ql - nz / nn;
                    if (ql > 0.0 && ql <= 512.0 && sqrt(ex * ex + ey * ey + ez * ez) <= LN_B4_ERR - 5.0e-4 - 1.0e-6 && abs(qx) <= 511 && abs(qy) <= 511 && abs(qz) <= 511)
                        b4[e] = ((uint32_t)qx & 0x3ffu) | (((uint32_t)qy & 0x3ffu) << 10) | (((uint32_t)qz & 0x3ffu) << 20);
                }
            }
        }
        if (r2b != 0x7c00u) {
            const float dxy = sqrtf((float)hx * (float)hx + (float)hy * (float)hy);
            g = (dxy - k1 * sqrtf(r2h) * 1.00001f) * 0.99999f - 1.0e-6f;
            pro = fmaxf(pro, dxy * 1.00001f);
            pz0 = fminf(pz0, (float)hz); pz1 = fmaxf(pz1, (float)hz);
            hb[e][0] = __builtin_bit_cast(uint16_t, hx); hb[e][1] = __builtin_bit_cast(uint16_t, hy); hb[e][2] = __builtin_bit_cast(uint16_t, hz);
        }
        hb[e][3] = r2b;
        G = fminf(G, g);
    }
    // order the pairs by G (ascending; -inf = a pair that is always a candidate first, +inf = an empty pair last); pairs with a
    // steep triangle in front of the rest (their G counts as it is in the bounds: only the ORDER is forced)
    const float Gs = (pq < q_good && G > -3.0e38f && G < 3.0e38f) ? -1.0e30f + G : G;
    const uint32_t gb = __float_as_uint(Gs);
    s_key[p] = (gb & 0x80000000u) ? ~gb : (gb | 0x80000000u);
    s_src[p] = (uint8_t)p;
    s_rec[p] = make_uint4((uint32_t)hb[0][0] | ((uint32_t)hb[0][1] << 16), (uint32_t)hb[0][2] | ((uint32_t)hb[0][3] << 16),
                          (uint32_t)hb[1][0] | ((uint32_t)hb[1][1] << 16), (uint32_t)hb[1][2] | ((uint32_t)hb[1][3] << 16));
    s_nrec[p] = half ? make_uint4((uint32_t)nb[0][0] | ((uint32_t)nb[0][1] << 16), (uint32_t)nb[0][2] | ((uint32_t)nb[0][3] << 16),
                                  (uint32_t)nb[1][0] | ((uint32_t)nb[1][1] << 16), (uint32_t)nb[1][2] | ((uint32_t)nb[1][3] << 16))
                     : make_uint4(b4[0], b4[1], 0u, 0u);
    s_id[p] = make_uint2(id[0] >= 0 ? (uint32_t)id[0] : CULL_NOID, id[1] >= 0 ? (uint32_t)id[1] : CULL_NOID);
    s_g[p] = G; s_z0[p] = pz0; s_z1[p] = pz1; s_ro[p] = pro; s_qn[p] = pq;
    __syncthreads();
    for (uint32_t len = 2; len <= 128u; len <<= 1) {
        for (uint32_t stride = len >> 1; stride > 0; stride >>= 1) {
            if (p < 64u) {
                const uint32_t lo = ((p / stride) * stride << 1) + (p % stride), hi = lo + stride;
                const bool up = (lo & len) == 0;
                const uint32_t a = s_key[lo], b = s_key[hi];
                const uint8_t sa = s_src[lo], sb = s_src[hi];
                if ((a > b || (a == b && sa > sb)) == up) { s_key[lo] = b; s_key[hi] = a; s_src[lo] = sb; s_src[hi] = sa; }
            }
            __syncthreads();
        }
    }
    if (p < pp) {       // a cell's row: the pp records of test (A), then the pp records of test (B)
        const uint32_t src = s_src[p];
        lrec[(uint64_t)cell * 2u * pp + p] = s_rec[src];
        // test (B)'s records behind them: fp16 proof 16 B per pair {n, r2B} x 2; f32 proof 8 B per pair (B4 x 2: the row's last quarter stays unused)
        if (half) lrec[(uint64_t)cell * 2u * pp + pp + p] = s_nrec[src];
        else reinterpret_cast<uint2*>(lrec + (uint64_t)cell * 2u * pp + pp)[p] = make_uint2(s_nrec[src].x, s_nrec[src].y);
        lid[(uint64_t)cell * pp + p] = s_id[src];
    }
    // levels: the bound of the suffix behind the first 16 j pairs
    __shared__ uint32_t s_q16[LN_MAXCH];
    __shared__ uint2 s_lv[LN_MAXCH];
    if (p < LN_MAXCH) {
        float g = __builtin_inff(), z0 = __builtin_inff(), z1 = -__builtin_inff(), ro = 0.0f, qn = 2.0f;
        for (uint32_t q = p * LN_CH; q < 128u; ++q) {
            const uint32_t src = s_src[q];
            g = fminf(g, s_g[src]); z0 = fminf(z0, s_z0[src]); z1 = fmaxf(z1, s_z1[src]); ro = fmaxf(ro, s_ro[src]); qn = fminf(qn, s_qn[src]);
        }
        if (!(z0 <= z1)) { z0 = 0.0f; z1 = 0.0f; }
        // fp16, each rounded to the side that keeps the bound a bound (z0, z1 are fp16 values already: the centres' decoded z)
        auto down = [](float v) { _Float16 h = (_Float16)v; uint16_t b = __builtin_bit_cast(uint16_t, h);
                                  if ((float)h > v) b = (uint16_t)((b & 0x8000u) ? b + 1u : (b == 0u ? 0x8001u : b - 1u)); return b; };
        auto up = [](float v) { _Float16 h = (_Float16)v; uint16_t b = __builtin_bit_cast(uint16_t, h);
                                if ((float)h < v) b = (uint16_t)((b & 0x8000u) ? (b == 0x8000u ? 1u : b - 1u) : b + 1u); return b; };
        // G with the scan's margins folded in (lane_scan_kernel's level loop); the levels behind the row's last chunk clear nothing
        const uint16_t gq = p * LN_CH < pp ? down(g * 0.9999f - 1.0e-5f) : (uint16_t)0xfc00u;
        s_lv[p] = make_uint2((uint32_t)gq | ((uint32_t)down(z0) << 16), (uint32_t)up(z1) | ((uint32_t)up(ro) << 16));
        // the suffix's normal cone, as idx4_build_kernel encodes a cell's (an empty suffix, or one of always-candidates only — whose G is -inf
        // anyway —: the widest)
        uint32_t q16 = 0xfffeu;
        if (qn <= 1.0f) { q16 = qn > 0.0f ? (uint32_t)floorf(qn * 65535.0f) : 0u; q16 = q16 > 0xfffeu ? 0xfffeu : q16; }
        s_q16[p] = q16;
    }
    __syncthreads();
    if (p == 0) lvl[(uint64_t)cell * LN_LVL] = make_float4(ccx, ccy, zc, __uint_as_float(qrow[cell]));
    if (p < 8u) {           // levels 2p, 2p + 1 in one float4
        lvl[(uint64_t)cell * LN_LVL + 1u + p] = make_float4(__uint_as_float(s_lv[2u * p].x), __uint_as_float(s_lv[2u * p].y),
                                                            __uint_as_float(s_lv[2u * p + 1u].x), __uint_as_float(s_lv[2u * p + 1u].y));
    } else if (p < 10u) {   // cones of levels 8 (p - 8) ... + 7
        const uint32_t b = 8u * (p - 8u);
        lvl[(uint64_t)cell * LN_LVL + 9u + (p - 8u)] = make_float4(__uint_as_float(s_q16[b] | (s_q16[b + 1u] << 16)), __uint_as_float(s_q16[b + 2u] | (s_q16[b + 3u] << 16)),
                                                                 __uint_as_float(s_q16[b + 4u] | (s_q16[b + 5u] << 16)), __uint_as_float(s_q16[b + 6u] | (s_q16[b + 7u] << 16)));
    }
}

// s - (fp16 half `HI` of `packed`), and a * b - (fp16 half of packed), in one v_fma_mix_f32 each
template <int HI> __device__ __forceinline__ float mix_rsub(uint32_t packed, float s) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(s));
    else    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(s));
    return r;
}
__device__ __forceinline__ float mix_fms_hi(float a, float b, uint32_t packed) {       // a * b - (high half of packed)
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(packed));
    return r;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= (uint32_t)off) v += o;
    }
    return v;
}

// (n . d)^2-side helpers of test (B) on a fp16 record: a * b + c with a = fp16 half `HI` of `packed`
template <int HI> __device__ __forceinline__ float mix_mul(uint32_t packed, float b) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(b));
    else    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(b));
    return r;
}
template <int HI> __device__ __forceinline__ float mix_fma(uint32_t packed, float b, float c) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(b), "v"(c));
    else    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(b), "v"(c));
    return r;
}

// the exact phase on 2-byte entries {ray position | pair position << 6}: ids through the cell's id row, then as cull_exact
// (cellm: per lane, the cell of run position `lane` with its map in bit 31)
template <int H>
__device__ __forceinline__ void lane_exact(const RayRec* __restrict__ rays, const RawTri* __restrict__ rt0, const RawTri* __restrict__ rt1,
                                           const uint2* __restrict__ lid0, const uint2* __restrict__ lid1, uint32_t pp01, const uint16_t* q, uint32_t n,
                                           const float4* s_abs /* LDS: per run position {origin, -}, {direction, -} as the ray record holds them */,
                                           uint32_t cellm, uint32_t lane, uint32_t* bk) {
    if (n == 0u) return;
    auto ids_of = [&](uint32_t i, uint32_t& pos, uint32_t& map) {
        const uint32_t e = q[min(i, n - 1u)];
        pos = e & 63u;
        const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos << 2), (int)cellm);
        map = c >> 31;
        const uint32_t pp = map ? pp01 >> 16 : pp01 & 0xffffu;
        return (map ? lid1 : lid0)[(uint64_t)(c & 0x7fffffffu) * pp + (e >> 6)];
    };
    // Two rounds in flight: while round k's 64 entries are evaluated, round k + 1's triangle / ray records (addressed through its ids,
    // which arrived during round k - 1) and round k + 2's ids are on their way — a round waits for memory only where the arithmetic of
    // the round before was shorter than a gather's latency.
    struct Recs { RawTri r0, r1; float4 ra, rb; uint32_t id0, id1, pos; };
    auto recs_of = [&](uint2 idp, uint32_t pos, uint32_t map) {
        Recs x;
        const RawTri* rt = map ? rt1 : rt0;
        x.id0 = idp.x; x.id1 = idp.y; x.pos = pos;
        x.r0 = rt[x.id0 == CULL_NOID ? 0u : x.id0]; x.r1 = rt[x.id1 == CULL_NOID ? 0u : x.id1];
        x.ra = s_abs[2u * pos]; x.rb = s_abs[2u * pos + 1u];          // (from LDS: two gathers less per round)
        return x;
    };
    uint32_t pos_next, map_next;
    uint2 id_next = ids_of(lane, pos_next, map_next);
    Recs nx = recs_of(id_next, pos_next, map_next);
    if (64u < n) id_next = ids_of(64u + lane, pos_next, map_next);
    for (uint32_t base = 0; base < n; base += 64u) {
        const bool live = base + lane < n;
        const Recs cur = nx;
        if (base + 64u < n) {
            nx = recs_of(id_next, pos_next, map_next);
            if (base + 128u < n) id_next = ids_of(base + 128u + lane, pos_next, map_next);
        }
        const RawTri r0 = cur.r0, r1 = cur.r1;
        const float4 ra = cur.ra, rb = cur.rb;
        const uint32_t id0 = cur.id0, id1 = cur.id1, pos = cur.pos;
        float best;
        if (H) {
            h2 v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const half2v x0 = __builtin_bit_cast(half2v, r0.d[k >> 1]), x1 = __builtin_bit_cast(half2v, r1.d[k >> 1]);
                v[k] = (k & 1) ? h2{x0.y, x1.y} : h2{x0.x, x1.x};
            }
            const _Float16 hnan = (_Float16)__builtin_nanf("");
            v[6] = h2{id0 == CULL_NOID ? hnan : v[6].x, id1 == CULL_NOID ? hnan : v[6].y};
            CellRegsH<1> t;
            set_pair_h(t, 0, v);
            const uint64_t none[1][2] = {{0, 0}};
            const _Float16 hsx = (_Float16)ra.x, hsy = (_Float16)ra.y, hsz = (_Float16)ra.z;
            const _Float16 hdx = (_Float16)rb.x, hdy = (_Float16)rb.y, hdz = (_Float16)rb.z;
            best = cast_pairs_h<1>(t, h2{hsx, hsx}, h2{hsy, hsy}, h2{hsz, hsz}, h2{hdx, hdx}, h2{hdy, hdy}, h2{hdz, hdz}, none, 0u);
        } else {
            const float qnan = __builtin_nanf("");
            f2 v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const f2 x0 = cvt2(r0.d[k >> 1]), x1 = cvt2(r1.d[k >> 1]);
                v[k] = (k & 1) ? f2{x0.y, x1.y} : f2{x0.x, x1.x};
            }
            v[6] = f2{id0 == CULL_NOID ? qnan : v[6].x, id1 == CULL_NOID ? qnan : v[6].y};
            CellRegs<1> t;
            set_pair(t, 0, v);
            const uint64_t none[1][2] = {{0, 0}};
            best = cast_pairs<1>(t, f2{ra.x, ra.x}, f2{ra.y, ra.y}, f2{ra.z, ra.z}, f2{rb.x, rb.x}, f2{rb.y, rb.y}, f2{rb.z, rb.z}, none, 0u);
        }
        const uint32_t k = fkey(best);
        if (live && k < fkey(RAY_MISS)) atomicMin(bk + pos, k);
    }
}

#define LANE_SCAN_ARGS                                                                                                                        \
    const RayRec *__restrict__ rays, const uint32_t *__restrict__ sorted, uint32_t n_sorted, const float4 *__restrict__ lvl0,                 \
        const float4 *__restrict__ lvl1, const uint4 *__restrict__ lrec0, const uint4 *__restrict__ lrec1, const uint2 *__restrict__ lid0,    \
        const uint2 *__restrict__ lid1, const RawTri *__restrict__ rtab0, const RawTri *__restrict__ rtab1, uint32_t pp01, uint32_t run,      \
        uint32_t n_blocks, uint32_t split, uint32_t t8, uint32_t r8, uint32_t chsr, uint32_t run_r, float *__restrict__ out,                  \
        uint4 *__restrict__ stats, float k2_far, float c_a, uint32_t *__restrict__ diag

#ifdef ROVER_DIAG_SORTED_RECS       // diagnostic builds only: the ray records copied into sorted order first (not timed with the kernel), read coalesced
__device__ const RayRec* g_diag_recs = nullptr;
__global__ void diag_gather_recs(const RayRec* __restrict__ rays, const uint32_t* __restrict__ sorted, uint32_t n, RayRec* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* src = reinterpret_cast<const float4*>(rays + sorted[i]);
    float4* dst = reinterpret_cast<float4*>(out + i);
    dst[0] = src[0]; dst[1] = src[1];
}
#endif

template <int H, int DIAG>
__global__ void __attribute__((amdgpu_waves_per_eu(LN_WAVES, 8))) __launch_bounds__(64) lane_scan_kernel(LANE_SCAN_ARGS) {
    __shared__ float4 s_ray[128];                         // per ray {s'x, s'y, s'z, dx}, {dy, dz, the cell's record row (64-bit address)}
    __shared__ uint16_t s_q[LN_QCAP];                     // the queue of the exact phase: ray | pair position << 6; before it, while the tests run:
    static_assert(LN_QCAP >= 64 * LN_MAXCH, "the items share the queue's array");
    uint16_t* const s_items = s_q;                        // ray | chunk << 6: the items that run test (A) only, behind them the ones that run (A) and (B)
    __shared__ __attribute__((aligned(16))) uint8_t s_cand[64 * LN_MAXCH];             // candidate mask of (ray, chunk): bit 7 - i = pair i of the chunk
    __shared__ uint32_t s_bk[64];
    __shared__ float4 s_abs[128];                         // the ray records' origins and directions, for the exact phase
    const uint32_t lane = threadIdx.x, x = blockIdx.x & 7u, qx = blockIdx.x >> 3, w = qx & 3u, jslot = qx >> 2;
    // blocks [0, split): runs of `run` rays of the terrain part of the sorted list, then runs of `run_r` rays of the rocks part, dealt
    // to the XCDs like cull_scan_kernel's (chunks of blocks round robin: neighbouring bins share an L2)
    uint32_t lb;
    const uint32_t chs = chsr & 0xffu, chr = chsr >> 8;
    if (jslot < t8) {
        lb = chs == 31u ? x * t8 + jslot : ((((jslot >> chs) << 3) + x) << chs) + (jslot & ((1u << chs) - 1u));
        if (lb >= split) return;
    } else {
        const uint32_t jr = jslot - t8;
        lb = split + (chr == 31u ? x * r8 + jr : ((((jr >> chr) << 3) + x) << chr) + (jr & ((1u << chr) - 1u)));
        if (lb >= n_blocks) return;
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(lb * 4u + w);
    const uint32_t my_run = lb < split ? run : run_r;
    const uint32_t i0 = lb < split ? wave * run : split * 4u * run + (wave - split * 4u) * run_r;
    if (i0 >= n_sorted) return;
    const uint32_t n_run = min(my_run, n_sorted - i0);
    const bool in_run = lane < n_run;
    const uint64_t t_start = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    // sorted == NULL: the ray slots in env order (small batches: a bin holds a ray or none, the sort's three launches buy nothing; a run is
    // then `run` consecutive slots — 16, 32 or 64: run_raycast —, padding slots — flags bit 1 clear — take no part)
    const uint32_t gid = sorted ? sorted[i0 + (in_run ? lane : n_run - 1u)] : i0 + (in_run ? lane : n_run - 1u);
#ifdef ROVER_DIAG_SORTED_RECS
    const RayRec* const rrec = (sorted && g_diag_recs) ? g_diag_recs + (i0 + (in_run ? lane : n_run - 1u)) : rays + gid;
#else
    const RayRec* const rrec = rays + gid;
#endif
    const float4 rsa = reinterpret_cast<const float4*>(rrec)[0], rsb = reinterpret_cast<const float4*>(rrec)[1];
    const uint32_t cell = __float_as_uint(rsa.w), rflags = __float_as_uint(rsb.w), map = rflags & 1u;
    const bool act = in_run && (rflags & 2u) != 0u;
    const uint32_t pp = map ? pp01 >> 16 : pp01 & 0xffffu, nch = pp / LN_CH;
    const float4* lp = (map ? lvl1 : lvl0) + (uint64_t)cell * LN_LVL;
    const float4 hdr = lp[0];
    float4 lv[8];                                             // 16 levels x {G, z0 | z1, rho_out} (fp16)
#pragma unroll
    for (int k = 0; k < 8; ++k) lv[k] = lp[1 + k];
    const float4 lq0 = lp[9], lq1 = lp[10];                   // the suffixes' cones, 16 bits each
    s_bk[lane] = fkey(RAY_MISS);
    // the ray relative to its cell; its level = the first suffix it clears as a group: by distance (far_build_kernel's inequality on the
    // suffix's bound) and, for test (B), by lying inside the suffix's normal cone
    const float sx = rsa.x - hdr.x, sy = rsa.y - hdr.y, sz = rsa.z - hdr.z;
    const bool tame = fabsf(sx) < 1.0e4f && fabsf(sy) < 1.0e4f && fabsf(sz) < 1.0e4f && fabsf(rsb.x) <= 2.0f && fabsf(rsb.y) <= 2.0f && fabsf(rsb.z) <= 2.0f;
    const uint32_t rq = rflags >> 16;                         // the ray's own cone bound (0xffff: none)
    const bool cone = __float_as_uint(hdr.w) >= rq;           // the whole cell's cone covers the ray: test (A) alone decides
    uint32_t L = nch;
    {
        // suffix k is cleared as a group when   G_k |dz| > e_max |dz| + k2 a_max |dz|   (far_build_kernel's inequality times |dz|) with
        // e_max |dz| = o |dz| + dzm |dxy|,  a_max = dzm |dz| + (o + rho_k) |dxy|,  o = the origin's distance from the cell centre, dzm =
        // the largest height difference to the suffix.  Sorted by what depends on the level:
        //     |dz| G_k - c3 rho_k - (|dz| + c3) o  >  dzm_k (|dxy| + k2 |dz|^2),     c3 = k2 |dz| |dxy|
        // every term of the right-hand sides with the factor 1.0004 (>= the 1.0001^3 the unsorted form gives its largest term), G_k with
        // its 0.9999 G - 1e-5 from lane_build_kernel: two v_fma_mix and a compare per level instead of twenty operations
        const float o = __builtin_amdgcn_sqrtf(sx * sx + sy * sy);
        const float dxy2 = rsb.x * rsb.x + rsb.y * rsb.y, adz = fabsf(rsb.z), sq = __builtin_amdgcn_sqrtf(dxy2);
        const bool steep = adz * adz >= 0.81f * (dxy2 + adz * adz) * 1.0001f;
        const float c3 = k2_far * adz * sq * 1.0004f, nc3 = -c3;
        const float c4 = (sq + k2_far * adz * adz) * 1.0004f;
        const float base = -((adz * 1.0004f + c3) * o);
        const uint32_t qw[8] = {__float_as_uint(lq0.x), __float_as_uint(lq0.y), __float_as_uint(lq0.z), __float_as_uint(lq0.w),
                                __float_as_uint(lq1.x), __float_as_uint(lq1.y), __float_as_uint(lq1.z), __float_as_uint(lq1.w)};
#pragma unroll
        for (int k = (int)LN_MAXCH - 1; k >= 0; --k) {
            const uint32_t w0 = __float_as_uint((k & 1) ? lv[k >> 1].z : lv[k >> 1].x), w1 = __float_as_uint((k & 1) ? lv[k >> 1].w : lv[k >> 1].y);
            const float dzm = fmaxf(fabsf(mix_rsub<1>(w0, sz)), fabsf(mix_rsub<0>(w1, sz)));
            const float lhs = mix_fma<0>(w0, adz, mix_fma<1>(w1, nc3, base));
            const uint32_t q16 = (k & 1) ? qw[k >> 1] >> 16 : qw[k >> 1] & 0xffffu;
            L = (lhs > dzm * c4) & (q16 >= rq) ? (uint32_t)k : L;
        }
        if (!steep) L = nch;
    }
    const bool allc = act && !tame;                           // a wild ray: every pair of the cell is a candidate, nothing is tested
    const bool ab = act && tame && !cone;                     // off the cell's cone: the prefix runs tests (A) and (B)
    if (allc) L = nch;
    if (!act) L = 0u;
#ifdef ROVER_DIAG_AB_MAXCH      // diagnostic builds only (wrong results, right timing): the (A) + (B) items of a ray stop after this many chunks
    const uint32_t n_a = (act && !allc && !ab) ? L : 0u, n_b = ab ? min(L, (uint32_t)(ROVER_DIAG_AB_MAXCH)) : 0u;
#else
    const uint32_t n_a = (act && !allc && !ab) ? L : 0u, n_b = ab ? L : 0u;
#endif
    // (diagnostic of the library's own: ROVER_LANE_DIAG=1 prints where a wave's time goes — launch_raycast_lane)
    uint64_t tq = 0;
    uint32_t dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto lap = [&](int k) { if (DIAG) { const uint64_t n = __builtin_amdgcn_s_memtime(); dg[k] += (uint32_t)(n - tq); tq = n; } };
    if (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tq = __builtin_amdgcn_s_memtime();
        dg[5] = (uint32_t)(tq - t_start);
        for (uint32_t v = 0; v < (diag[(size_t)n_blocks * 4u * 8u + 39u] ? 19u : 0u); ++v) {    // (ROVER_LANE_DIAG=2) histogram of the rays' levels behind the per-wave rows: its atomics distort the times
            const uint32_t cnt = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act && !map && (allc ? 17u : (ab ? 18u : L)) == v));
            const uint32_t cnt1 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act && map && (allc ? 17u : (ab ? 18u : L)) == v));
            if (lane == 0u && cnt) atomicAdd(diag + (size_t)n_blocks * 4u * 8u + v, cnt);
            if (lane == 0u && cnt1) atomicAdd(diag + (size_t)n_blocks * 4u * 8u + 20u + v, cnt1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tq = __builtin_amdgcn_s_memtime();
    }
    // The items, in the order bin by bin, chunk by chunk, the bin's rays that test the chunk: lanes that read the same 128-byte line of a
    // record row sit next to each other.  Position of item (ray, k) = items of the bins before + items of the bin's chunks before k + the
    // ray's rank among the bin's rays with more than k items.  Two lists: the (A) items, behind them the (A) + (B) items.
    const uint32_t key = cell | (map << 31);
    const uint32_t prevk = (uint32_t)__shfl_up((int)key, 1, 64);
    const bool head = act && (lane == 0u || key != prevk);
    const uint64_t heads = __builtin_amdgcn_ballot_w64(head);
    const uint64_t lt = (1ull << lane) - 1ull, le = (lt << 1) | 1ull;         // bits below / up to this lane
    const uint64_t below = heads & le, above = heads & ~le;
    const uint32_t lo = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u, hi = above ? (uint32_t)__builtin_ctzll(above) : n_run;
    const uint64_t binmask = (hi >= 64u ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
    const uint32_t ia_incl = wave_incl_scan(n_a, lane), ib_incl = wave_incl_scan(n_b, lane);
    const uint32_t ia_tot = (uint32_t)__builtin_amdgcn_readlane((int)ia_incl, 63), ib_tot = (uint32_t)__builtin_amdgcn_readlane((int)ib_incl, 63);
    const uint32_t base_a = (uint32_t)__shfl((int)(ia_incl - n_a), (int)lo, 64), base_b = ia_tot + (uint32_t)__shfl((int)(ib_incl - n_b), (int)lo, 64);
    wave_lds_sync();
    {
        const char* rowp = reinterpret_cast<const char*>(map ? lrec1 : lrec0) + (uint64_t)cell * (2u * 16u) * pp;
        const uint64_t ra64 = (uint64_t)reinterpret_cast<uintptr_t>(rowp);
        s_abs[2u * lane] = rsa; s_abs[2u * lane + 1u] = rsb;
        s_ray[2u * lane] = make_float4(sx, sy, sz, rsb.x);
        s_ray[2u * lane + 1u] = make_float4(rsb.y, rsb.z, __uint_as_float((uint32_t)ra64), __uint_as_float((uint32_t)(ra64 >> 32)));
        uint32_t off_a = base_a, off_b = base_b;
#pragma unroll
        for (uint32_t k = 0; k < LN_MAXCH; ++k) {
            const uint64_t ma = __builtin_amdgcn_ballot_w64(n_a > k) & binmask, mb = __builtin_amdgcn_ballot_w64(n_b > k) & binmask;
            if (k < n_a) s_items[off_a + (uint32_t)__builtin_popcountll(ma & lt)] = (uint16_t)(lane | (k << 6));
            if (k < n_b) s_items[off_b + (uint32_t)__builtin_popcountll(mb & lt)] = (uint16_t)(lane | (k << 6));
            off_a += (uint32_t)__builtin_popcountll(ma); off_b += (uint32_t)__builtin_popcountll(mb);
        }
    }
    uint32_t cused = 0, ctot = 0, n_flush = 0;
    wave_lds_sync();
    lap(0);
    // the tests: 64 items per round, the records straight from the cell's row (L1 / L2: a chunk is read by the bin's rays side by side)
    auto rounds = [&](uint32_t it0, uint32_t it1, auto AB) {
        constexpr bool kAB = decltype(AB)::value;
        for (uint32_t base = it0; base < it1; base += 64u) {
            const bool ok = base + lane < it1;
            const uint32_t it = s_items[ok ? base + lane : it0];
            const uint32_t rl = it & 63u, ch = it >> 6;
            const float4 ra = s_ray[2u * rl], rb = s_ray[2u * rl + 1u];
            // (the row's address comes out of LDS as an integer: say that it is global memory, or the loads are FLAT ones — which count on
            // both wait counters and so cannot be waited for one by one)
            typedef uint32_t U4 __attribute__((ext_vector_type(4)));
            typedef const U4 __attribute__((address_space(1))) * GRow;
            const GRow cp = (GRow)((uintptr_t)((uint64_t)__float_as_uint(rb.z) | ((uint64_t)__float_as_uint(rb.w) << 32))) + ch * LN_CH;
            // (the (B) records of the row lie pp records behind the (A) records; pp by the map of the item's ray)
            const uint32_t ppi = kAB ? ((uint32_t)__builtin_amdgcn_ds_bpermute((int)(rl << 2), (int)map) ? pp01 >> 16 : pp01 & 0xffffu) : 0u;
            uint32_t mask = 0;
            constexpr uint32_t NB = kAB ? (H ? LN_ABB : LN_ABB4) : LN_AB;            // records in flight per batch
#pragma unroll 1                                                // (one batch of LN_AB pairs at a time: with 16-pair chunks fully unrolled the compiler kept all in flight, 128 VGPRs and spills)
            for (uint32_t hf = 0; hf < LN_CH / NB; ++hf) {
                uint4 rec[NB], nrc[kAB ? NB : 1u];
#pragma unroll
                for (uint32_t i = 0; i < NB; ++i) {
                    { const U4 v = cp[hf * NB + i]; rec[i] = make_uint4(v.x, v.y, v.z, v.w); }
                    if (kAB && H) { const U4 v = cp[ppi + hf * NB + i]; nrc[i] = make_uint4(v.x, v.y, v.z, v.w); }
                }
                if (kAB && !H) {      // B4 records, 8 B per pair: two pairs per 16-byte load (the pair's chunk starts ch * 64 bytes into the B4 rows)
                    const GRow bp = (GRow)((uintptr_t)((uint64_t)__float_as_uint(rb.z) | ((uint64_t)__float_as_uint(rb.w) << 32))) + ppi + ch * (LN_CH / 2u);
#pragma unroll
                    for (uint32_t i = 0; i < NB; i += 2) {
                        const U4 v = bp[(hf * NB + i) >> 1];
                        nrc[i] = make_uint4(v.x, v.y, 0u, 0u); nrc[i + 1u] = make_uint4(v.z, v.w, 0u, 0u);
                    }
                }
#pragma unroll
                for (uint32_t i = 0; i < NB; ++i) {
                    const uint4 r = rec[i];
                    uint32_t sg;
                    {
                        const float hx = mix_rsub<0>(r.x, ra.x), hy = mix_rsub<1>(r.x, ra.y), hz = mix_rsub<0>(r.y, ra.z);
                        float t = hx * ra.w; t = __builtin_fmaf(hy, rb.x, t); t = __builtin_fmaf(hz, rb.y, t);
                        float qq = hx * hx; qq = __builtin_fmaf(hy, hy, qq); qq = __builtin_fmaf(hz, hz, qq);
                        float u = mix_fms_hi(qq, c_a, r.y);
                        u = __builtin_fmaf(-t, t, u);
                        sg = __float_as_uint(u);
                    }
                    {
                        const float hx = mix_rsub<0>(r.z, ra.x), hy = mix_rsub<1>(r.z, ra.y), hz = mix_rsub<0>(r.w, ra.z);
                        float t = hx * ra.w; t = __builtin_fmaf(hy, rb.x, t); t = __builtin_fmaf(hz, rb.y, t);
                        float qq = hx * hx; qq = __builtin_fmaf(hy, hy, qq); qq = __builtin_fmaf(hz, hz, qq);
                        float u = mix_fms_hi(qq, c_a, r.w);
                        u = __builtin_fmaf(-t, t, u);
                        sg |= __float_as_uint(u);
                    }
                    if (kAB && !H) {      // (B) on B4 records: (n4 . d)^2 - LN_B4_C >= +0 for both triangles, or the pair stays a candidate
                        const uint4 n = nrc[i];
                        auto b4t = [&](uint32_t c) {
                            const float nx = (float)__builtin_amdgcn_sbfe((int)c, 0, 10), ny = (float)__builtin_amdgcn_sbfe((int)c, 10, 10),
                                        nz = (float)__builtin_amdgcn_sbfe((int)c, 20, 10);
                            float t = nx * ra.w; t = __builtin_fmaf(ny, rb.x, t); t = __builtin_fmaf(nz, rb.y, t);
                            return __builtin_fmaf(t, t, -LN_B4_C);
                        };
                        sg |= __float_as_uint(b4t(n.x)) | __float_as_uint(b4t(n.y));
                    } else if (kAB) {      // (B): (n . d)^2 - r2B >= +0 for both triangles, or the pair stays a candidate
                        const uint4 n = nrc[i];
                        float t0 = mix_mul<0>(n.x, ra.w); t0 = mix_fma<1>(n.x, rb.x, t0); t0 = mix_fma<0>(n.y, rb.y, t0);
                        float t1 = mix_mul<0>(n.z, ra.w); t1 = mix_fma<1>(n.z, rb.x, t1); t1 = mix_fma<0>(n.w, rb.y, t1);
                        sg |= __float_as_uint(mix_fms_hi(t0, t0, n.y)) | __float_as_uint(mix_fms_hi(t1, t1, n.w));
                    }
                    mask = __builtin_amdgcn_alignbit(mask, sg, 31);       // (mask << 1) | sign: a pair is a candidate unless every u >= +0
                }
            }
            if (ok) s_cand[rl * LN_MAXCH + ch] = (uint8_t)mask;
        }
    };
    rounds(0u, ia_tot, std::false_type{});
    if (ib_tot) rounds(ia_tot, ia_tot + ib_tot, std::true_type{});
    wave_lds_sync();
    lap(2);
    // candidates -> queue entries; the exact phase whenever the queue could not take the next ray's entries (rare) and at the end
    auto flush = [&]() {
        wave_lds_sync();
        lap(3);
        lane_exact<H>(rays, rtab0, rtab1, lid0, lid1, pp01, s_q, cused, s_abs, key, lane, s_bk);
        ctot += cused;
        cused = 0;
        ++n_flush;
        wave_lds_sync();
        lap(4);
    };
    {
        uint32_t cm[LN_MAXCH / 4u], cnt = 0;                      // the ray's masks, four chunks per register (chunk k in byte k % 4 of word k / 4)
#pragma unroll
        for (uint32_t k4 = 0; k4 < LN_MAXCH / 4u; ++k4) {
            uint32_t w4 = allc ? 0xffffffffu : *reinterpret_cast<const uint32_t*>(&s_cand[lane * LN_MAXCH + 4u * k4]);
            // chunks at and behind the ray's level hold nothing of this step
            const uint32_t keep = L >= 4u * k4 + 4u ? 0xffffffffu : (L > 4u * k4 ? (1u << (8u * (L - 4u * k4))) - 1u : 0u);
            w4 = act ? w4 & keep : 0u;
            cm[k4] = w4; cnt += (uint32_t)__builtin_popcount(w4);
        }
        const uint32_t e_incl = wave_incl_scan(cnt, lane), e_pre = e_incl - cnt;
        const uint64_t actm = __builtin_amdgcn_ballot_w64(act);
        const uint32_t r_end = actm ? 64u - (uint32_t)__builtin_clzll(actm) : 0u;          // one past the last ray that takes part (env order: padding slots do not)
        uint32_t r_lo = 0;
        while (r_lo < r_end) {
            const uint32_t base_e = (uint32_t)__builtin_amdgcn_readlane((int)e_pre, (int)r_lo);
            // (lanes that take no part have no entries: they "fit" wherever their prefix does, so the fitting lanes stay one contiguous range)
            const uint64_t fm = __builtin_amdgcn_ballot_w64(lane >= r_lo && lane < r_end && e_incl - base_e <= LN_QCAP - cused);
            if (!fm) {
                if (cused == 0u) break;                        // (cannot happen: an empty queue takes any one ray's <= 128 entries)
                flush();
                continue;
            }
            const uint32_t r_hi = 64u - (uint32_t)__builtin_clzll(fm);
            if (act && lane >= r_lo && lane < r_hi) {
                uint32_t at = cused + (e_pre - base_e);
#pragma unroll
                for (uint32_t k4 = 0; k4 < LN_MAXCH / 4u; ++k4) {
                    uint32_t m = cm[k4];
                    while (m) {             // bit 8 c + 7 - i of the word = pair i of chunk 4 k4 + c
                        const uint32_t b = (uint32_t)__builtin_ctz(m);
                        m &= m - 1u;
                        s_q[at++] = (uint16_t)(lane | (((4u * k4 + (b >> 3)) * LN_CH + (7u - (b & 7u))) << 6));
                    }
                }
            }
            cused += (uint32_t)__builtin_amdgcn_readlane((int)e_incl, (int)(r_hi - 1u)) - base_e;
            r_lo = r_hi;
            if (r_lo < r_end) flush();
        }
    }
    lap(3);
    if (cused) flush();
    wave_lds_sync();
    if (act) out[gid] = funkey(s_bk[lane]);
    {
        const uint64_t am = n_run >= 64u ? ~0ull : ((1ull << n_run) - 1ull);
        const uint32_t n_both = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(allc || ab) & am);
        const uint32_t n_askip = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act && L == 0u) & am);
        const uint32_t n_fskip = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act && !allc && 2u * L <= nch) & am);
        const uint32_t n_bins = (uint32_t)__builtin_popcountll(heads);
        const uint32_t n_rays = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act));          // (env order: without the padding slots)
        if (lane == 0u) stats[wave] = make_uint4(ctot, n_rays | (n_fskip << 8), n_both | (n_askip << 8), n_bins | (min(ia_tot + ib_tot, 0x3ffffu) << 8) | (min(n_flush, 63u) << 26));
        if (DIAG && lane == 0u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) diag[(size_t)wave * 8u + k] = dg[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint64_t n, uint32_t bs) { return (uint32_t)((n + bs - 1) / bs); }

hipError_t launch_tri_centroids(const int32_t* tris, const uint16_t* verts, uint32_t T, uint32_t V, float2* out, hipStream_t s) {
    hipLaunchKernelGGL(tri_centroid_kernel, dim3(blocks_for(T, 256)), dim3(256), 0, s, tris, verts, T, V, out);
    return hipGetLastError();
}

// the far-skip constants of a proof: k1 (folded into the far records) and k2 (the scan kernel's), from its c_a and the largest |d|^2
static void cull_far_consts(double c_a, double dd, float* k1, float* k2) {
    const double ca = c_a - 1.0e-5;
    *k1 = (float)(1.00001 / (0.9 * sqrt(ca)));
    *k2 = (float)(1.00001 * sqrt(dd + 1.0e-5 - ca) / (0.9 * sqrt(ca)));
}
float cull_far_k2(int half, CullProofH ph) {
    float k1, k2;
    if (half) cull_far_consts(ph.c_a, 1.004, &k1, &k2); else cull_far_consts(CullK<0>::c_a, 1.00001, &k1, &k2);
    // (the scan kernel bounds |h . d| with the ray's direction as stored; the derivation's a is the axial part along the UNIT direction,
    //  and the fp16-normalised d of the as-shipped arithmetic has |d|^2 >= 0.996: a <= |h . d| / 0.998)
    if (half) k2 *= 1.0021f;
    return k2;
}

uint32_t lane_pairs_per_row(uint32_t K8);
static void lane_build(const int32_t* idx4, const uint4* ctab, uint64_t n_cells, uint32_t K8, uint32_t Y, float cell_size, float shift_x, float shift_y,
                       const uint32_t* qrow, const float* nz_abs, LaneTables t, int half, CullProofH ph, hipStream_t s);
// T: the caller's triangle count (ids in map_idx); T_int: slots of the internal numbering (order [T_int], newid [T]).
// ctab / qrow / far: the f32 proof's tables; ctab_h / qrow_h / far_h: the as-shipped fp16 arithmetic's (CullK<1>); idx4 and rtab serve both.
hipError_t launch_cull_build(const int32_t* map_idx, const int32_t* tris, const uint16_t* verts, uint64_t n_cells, uint32_t K,
                             uint32_t K8, uint32_t T, uint32_t T_int, uint32_t V, const uint32_t* order, const uint32_t* newid,
                             int32_t* idx4, uint4* ctab, uint4* ctab_h, uint16_t* rtab, uint32_t* qrow, uint32_t* qrow_h, float4* far,
                             float4* far_h, float* nz_scratch,
                             uint32_t* counts /* [5], zeroed: always-candidate triangles, cells without a cone; the same for fp16; cells with a useful far bound */,
                             CullProofH ph, uint32_t Y, float cell_size, float shift_x, float shift_y, LaneTables lane, LaneTables lane_h, hipStream_t s) {
    hipLaunchKernelGGL(rtab_build_kernel, dim3(blocks_for(T_int, 256)), dim3(256), 0, s, tris, verts, T_int, V, order, rtab);
    hipLaunchKernelGGL(ctab_build_kernel<1>, dim3(blocks_for(T_int, 256)), dim3(256), 0, s, rtab, T_int, order, ctab_h, nz_scratch, counts + 2, ph);
    hipLaunchKernelGGL(idx4_build_kernel, dim3((uint32_t)n_cells), dim3(256), 0, s, map_idx, K, K8, T, newid, nz_scratch, rtab, Y, cell_size,
                       shift_x, shift_y, idx4, qrow_h, counts + 2);
    lane_build(idx4, ctab_h, n_cells, K8, Y, cell_size, shift_x, shift_y, qrow_h, nz_scratch, lane_h, 1, ph, s);      // (nz_scratch holds the fp16 proof's cone values here)
    hipLaunchKernelGGL(ctab_build_kernel<0>, dim3(blocks_for(T_int, 256)), dim3(256), 0, s, rtab, T_int, order, ctab, nz_scratch, counts, ph);
    hipLaunchKernelGGL(idx4_build_kernel, dim3((uint32_t)n_cells), dim3(256), 0, s, map_idx, K, K8, T, newid, nz_scratch, rtab, Y, cell_size,
                       shift_x, shift_y, idx4, qrow, counts);
    lane_build(idx4, ctab, n_cells, K8, Y, cell_size, shift_x, shift_y, qrow, nz_scratch, lane, 0, ph, s);
    float k1, k2;
    cull_far_consts(CullK<0>::c_a, 1.00001, &k1, &k2);
    hipLaunchKernelGGL(far_build_kernel, dim3(blocks_for(n_cells, 4)), dim3(256), 0, s, reinterpret_cast<const int4*>(idx4), ctab, n_cells, K8, Y,
                       cell_size, shift_x, shift_y, k1, CullK<0>::tau2, qrow, reinterpret_cast<FarRec*>(far), far + 2ull * n_cells, counts + 4);
    cull_far_consts(ph.c_a, 1.004, &k1, &k2);
    hipLaunchKernelGGL(far_build_kernel, dim3(blocks_for(n_cells, 4)), dim3(256), 0, s, reinterpret_cast<const int4*>(idx4), ctab_h, n_cells, K8, Y,
                       cell_size, shift_x, shift_y, k1, ph.tau2, qrow_h, reinterpret_cast<FarRec*>(far_h), far_h + 2ull * n_cells, (uint32_t*)nullptr);
    return hipGetLastError();
}

struct CullGrid { uint32_t run, run_r, split, n_blocks, chs, chr, t8, r8; };
static CullGrid cull_grid(uint32_t n_sorted, uint32_t n_terrain, uint32_t run) {
    CullGrid g{};
    if (run > CULL_RUNMAX) run = CULL_RUNMAX;
    if (run == 0) run = 1;
    g.run = run;
    // the sorted list is all terrain rays, then all rock rays: blocks [0, split) are (but for a few rays) terrain
    // The rocks part comes last on every XCD and drains the launch: with runs of 32 there its waves live half as long and the
    // machine empties faster at the end (last workgroup start to kernel end was 70 us of 580) — 0.575 -> 0.570 ms; 16: 0.581
    // (round 4: runs of 32 take 16 there too — 8 192 envs 53.7 -> 54.9 M env-steps/s, 12 288 envs 63.5 -> 64.2, 16 384 envs the same; 8: slower)
    g.run_r = run >= 32u ? run / 2u : run;
    g.split = blocks_for(blocks_for(n_terrain, run), 4);
    const uint64_t covered = (uint64_t)g.split * 4u * run;
    if (covered >= n_sorted) { g.split = blocks_for(blocks_for(n_sorted, run), 4); g.n_blocks = g.split; }
    else g.n_blocks = g.split + blocks_for(blocks_for(n_sorted - (uint32_t)covered, g.run_r), 4);
    // chunks of blocks dealt round robin to the 8 XCDs, per XCD ceil(chunks / 8) chunks of each part.  A chunk is 16 384 rays (64 blocks of
    // 4 runs of 64: a few map rows of bins, which share most of their triangles) when that gives every XCD >= 12 chunks of the part;
    // otherwise one contiguous eighth per XCD (4 096 envs, one call: chunks of 16 small blocks 0.112 ms, contiguous 0.104)
    // (65 536 envs, one call: 4 / 8 / 16 / 32 blocks 0.584-0.596 ms, 64 blocks 0.579-0.581, 256 blocks 0.631; contiguous eighths 0.590-0.597)
    auto chunk_shift = [](uint32_t blocks, uint32_t r) {
        uint32_t c = 0;
        while ((4u * r << c) < 16384u) ++c;                          // blocks per chunk = 2^c
        return (blocks >> c) >= 96u ? c : 31u;
    };
    g.chs = chunk_shift(g.split, run); g.chr = chunk_shift(g.n_blocks - g.split, g.run_r);
    g.t8 = g.chs == 31u ? blocks_for(g.split, 8) : blocks_for(blocks_for(g.split, 1u << g.chs), 8) << g.chs;
    g.r8 = g.chr == 31u ? blocks_for(g.n_blocks - g.split, 8) : blocks_for(blocks_for(g.n_blocks - g.split, 1u << g.chr), 8) << g.chr;
    return g;
}

// block slots per XCD one launch may cover so that its queue regions (8 XCDs x 4 waves x CULL_QGLOBAL entries per slot) fit `entries`
static uint32_t cull_slots_per_launch(uint64_t entries) {
    const uint64_t s = entries / (8ull * 4ull * CULL_QGLOBAL);
    return (uint32_t)(s < 1 ? 1 : (s > 0x7fffffffull ? 0x7fffffffull : s));
}

hipError_t launch_raycast_culled(CullArgs a, hipStream_t s) {
    const CullGrid g = cull_grid(a.n_sorted, a.n_terrain, a.run);

    const uint32_t slots = g.t8 + g.r8;                              // block slots per XCD
    const uint32_t per = cull_slots_per_launch(a.queue_entries);
    // one launch, unless the queue regions of all slots exceed the budget the queue was sized for (huge batches): then slices
    // of the slot list, one launch each on the same stream, re-using the regions
    for (uint32_t j0 = 0; j0 < slots; j0 += per) {
        const uint32_t n = slots - j0 < per ? slots - j0 : per;
        auto kern = a.half ? (a.skip_clear ? (a.lazy_far ? cull_scan_kernel<1, 1, 1> : cull_scan_kernel<1, 0, 1>) : cull_scan_kernel<1, 0, 0>)
                           : (a.lazy_far ? cull_scan_kernel<0, 1, 1> : (a.skip_clear ? cull_scan_kernel<0, 0, 1> : cull_scan_kernel<0, 0, 0>));
        hipLaunchKernelGGL(kern, dim3(n * 8u * (4u / CULL_WPB)), dim3(64 * CULL_WPB), 0, s, a.rays, a.sorted, a.n_sorted,
                           reinterpret_cast<const int4*>(a.idx0), reinterpret_cast<const int4*>(a.idx1), a.ctab0, a.ctab1,
                           a.kp0 | (a.kp1 << 16), g.run, g.n_blocks, g.split, g.t8, g.r8, g.chs | (g.chr << 8), g.run_r, a.queue,
                           reinterpret_cast<const RawTri*>(a.rtab0), reinterpret_cast<const RawTri*>(a.rtab1), a.out, a.stats, j0, a.c_a_h, a.tau2_h, a.far0, a.far1, a.k2_far, a.near0, a.near1);
    }
    return hipGetLastError();
}

// upper bound of the waves a launch over n_rays rays starts (= slots of the per-wave counter array)
uint32_t cull_stat_slots(uint64_t n_rays, uint32_t run) {
    if (run > CULL_RUNMAX) run = CULL_RUNMAX;
    if (run == 0) run = 1;
    const uint32_t rr = run >= 32u ? run / 2u : run;                // the rocks part walks shorter runs (cull_grid)
    return (uint32_t)(4u * ((n_rays / rr + 4u) / 4u + 5u));        // (+ the rounding of a second part: the staged ray cast counts its terrain waves in front)
}

// Entries of the candidate queue: one region of CULL_QGLOBAL entries per wave of a launch, capped by `budget_bytes` (a launch then
// covers a slice of the block slots; *n_launches says how many a step takes).
uint64_t cull_queue_entries(uint64_t n_rays, uint32_t n_terrain, uint32_t run, uint64_t budget_bytes, uint32_t* n_launches) {
    const CullGrid g = cull_grid((uint32_t)n_rays, n_terrain, run);
    const uint32_t slots = g.t8 + g.r8;
    uint32_t per = cull_slots_per_launch(budget_bytes / sizeof(uint2));
    if (per > slots) per = slots ? slots : 1u;
    if (n_launches) *n_launches = slots ? (slots + per - 1u) / per : 1u;
    return (uint64_t)per * 8u * 4u * CULL_QGLOBAL;
}

// ---- the staged ray cast (variant 4) ----
uint32_t lane_pairs_per_row(uint32_t K8) { return ((K8 / 2u + LN_CH - 1u) / LN_CH) * LN_CH; }
uint32_t lane_lvl_stride() { return LN_LVL; }

static void lane_build(const int32_t* idx4, const uint4* ctab, uint64_t n_cells, uint32_t K8, uint32_t Y, float cell_size, float shift_x, float shift_y,
                       const uint32_t* qrow, const float* nz_abs, LaneTables t, int half, CullProofH ph, hipStream_t s) {
    if (!t.lrec) return;
    float k1, k2;
    if (half) cull_far_consts(ph.c_a, 1.004, &k1, &k2); else cull_far_consts(CullK<0>::c_a, 1.00001, &k1, &k2);
    hipLaunchKernelGGL(lane_build_kernel, dim3((uint32_t)n_cells), dim3(128), 0, s, reinterpret_cast<const int4*>(idx4), ctab, K8, lane_pairs_per_row(K8),
                       Y, cell_size, shift_x, shift_y, k1, half ? ph.tau2 : CullK<0>::tau2, half ? ph.c_rho : CullK<0>::c_rho, qrow, nz_abs, t.lvl, t.lrec,
                       t.lid, half, half ? LN_QGOOD_H : LN_QGOOD);
}

uint32_t lane_waves(uint32_t n_rays, uint32_t run) {
    const CullGrid g = cull_grid(n_rays, n_rays, run);
    return g.n_blocks * 4u;
}

hipError_t launch_raycast_lane(LaneArgs a, hipStream_t s) {
    if (a.n_sorted == 0u) return hipSuccess;
    CullGrid g = cull_grid(a.n_sorted, a.n_terrain, a.run);
    {   // the rocks part in runs as long as the terrain's (its waves are short anyway: most rock rays clear their whole cell: 440 -> 423 us)
        g.run_r = g.run;
        const uint64_t covered = (uint64_t)g.split * 4u * g.run;
        g.n_blocks = covered >= a.n_sorted ? g.split : g.split + blocks_for(blocks_for(a.n_sorted - (uint32_t)covered, g.run_r), 4);
        g.chr = 31u; g.r8 = blocks_for(g.n_blocks - g.split, 8);
    }
    const float k2 = a.k2_far, c_a = a.half ? a.c_a_h : CullK<0>::c_a;
    static const bool want_diag = getenv("ROVER_LANE_DIAG") != nullptr;
    static uint32_t* d_diag = nullptr; static uint32_t diag_waves = 0; static int diag_left = 2;
    const uint32_t waves = g.n_blocks * 4u;
    if (want_diag && diag_waves < waves) { if (d_diag) (void)hipFree(d_diag); (void)hipMalloc((void**)&d_diag, ((size_t)waves * 8u + 40u) * sizeof(uint32_t)); diag_waves = waves; }
    if (want_diag && d_diag) {
        (void)hipMemsetAsync(d_diag, 0, ((size_t)waves * 8u + 40u) * sizeof(uint32_t), s);
        static const bool want_hist = atoi(getenv("ROVER_LANE_DIAG")) >= 2;      // the level histogram too
        if (want_hist) (void)hipMemsetAsync(d_diag + (size_t)waves * 8u + 39u, 1, 1, s);
    }
#ifdef ROVER_DIAG_SORTED_RECS
    if (a.sorted) {
        static RayRec* d_recs = nullptr; static uint32_t cap = 0;
        if (cap < a.n_sorted) { if (d_recs) (void)hipFree(d_recs); (void)hipMalloc((void**)&d_recs, (size_t)a.n_sorted * sizeof(RayRec)); cap = a.n_sorted;
                                const RayRec* p = d_recs; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_recs), &p, sizeof p); }
        hipLaunchKernelGGL(diag_gather_recs, dim3((a.n_sorted + 255u) / 256u), dim3(256), 0, s, a.rays, a.sorted, a.n_sorted, d_recs);
    }
#endif
    auto kern = a.half ? lane_scan_kernel<1, 0> : (want_diag && d_diag ? lane_scan_kernel<0, 1> : lane_scan_kernel<0, 0>);
    hipLaunchKernelGGL(kern, dim3((g.t8 + g.r8) * 8u * 4u), dim3(64), 0, s, a.rays, a.sorted,
                       a.n_sorted, a.lvl[0], a.lvl[1], a.lrec[0], a.lrec[1], a.lid[0], a.lid[1], reinterpret_cast<const RawTri*>(a.rtab[0]),
                       reinterpret_cast<const RawTri*>(a.rtab[1]), a.pp[0] | (a.pp[1] << 16), g.run, g.n_blocks, g.split, g.t8, g.r8, g.chs | (g.chr << 8),
                       g.run_r, a.out, a.stats, k2, c_a, (want_diag && !a.half) ? d_diag : nullptr);
    static int diag_seen = 0;
    if (want_diag && d_diag && ++diag_seen > 8 && diag_left > 0) {       // (not the first launches: cold caches)      // where a wave's time goes: mean shader-clock cycles per wave and phase (synchronises: a diagnostic)
        --diag_left;
        std::vector<uint32_t> h((size_t)waves * 8u + 40u);
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), d_diag, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
            const uint32_t* hg = h.data() + (size_t)waves * 8u;
            for (int part = 0; part < 2; ++part) {           // the waves of the terrain part, then of the rocks part of the sorted list
                const size_t w0 = part ? (size_t)g.split * 4u : 0u, w1 = part ? waves : std::min<size_t>(waves, (size_t)g.split * 4u);
                if (w1 <= w0) continue;
                double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t i = w0 * 8u; i < w1 * 8u; ++i) sum[i & 7u] += h[i];
                const double nw = (double)(w1 - w0);
                fprintf(stderr, "lane_scan_kernel, %s part, %zu waves, mean ticks per wave: prologue loads %.0f | bins/scans %.0f | items %.0f | entries %.0f | exact %.0f\n",
                        part ? "rocks" : "terrain", w1 - w0, sum[5] / nw, sum[0] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw);
            }
            for (int m = 0; m < (hg[39] ? 2 : 0); ++m) {
                fprintf(stderr, "  %s rays by level 0..16:", m ? "rock" : "terrain");
                for (int v = 0; v < 17; ++v) fprintf(stderr, " %u", hg[20 * m + v]);
                fprintf(stderr, ", wild (all pairs candidates): %u, off the cone (tests A and B): %u\n", hg[20 * m + 17], hg[20 * m + 18]);
            }
        }
    }
    return hipGetLastError();
}

}  // namespace rover
