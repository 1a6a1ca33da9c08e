// rover_cull.hip — the culled ray cast (raycast variant 3): the roofline kernel of the step on full batches.
//
// Reference: tasks/utils/camera/camera.py:60-145 (gather K triangles per ray, ray_distance, min over K),
//            tasks/utils/camera/ray_casting.py:31-59 (the (ray, triangle) test), rock_detect.py:52-149 (same on the rocks map).
//
// The reference evaluates all K = 200 triangles of a ray's cell and takes the min of k-or-11.0; a ray meets 1-3 of them.
// raycast_binned_kernel (rover_kernels.hip) evaluates them all too (134 VALU instructions per ray).  This kernel splits the
// work: PHASE 1 proves, per (ray, triangle), with 19 packed instructions per pair of triangles, that ray_casting.py:59
// rejects the triangle (it then contributes the 11.0 sentinel and nothing else); PHASE 2 runs the exact arithmetic of
// rover_raymath.h — the same code the other kernels run — on the few (ray, lane-pair) candidates phase 1 could not
// reject, 64 candidates of different rays side by side.  Results are bit-identical to raycast_binned_kernel /
// raycast_kernel (tests/test_hip_parity.py::test_raycast_variants_bit_identical, tools/soak_exact.py).
//
// ---- why a culled triangle is rejected by the reference (the proof behind phase 1) -----------------------------------
// Per triangle the cull table holds a sphere centre m (any point) and, folded into the length of a scaled unit normal,
// r2 >= 1.19 (rho + 1e-4)^2, where rho = max distance from m to the corners of the PADDED triangle
// {a + n b + m c : n, m >= -0.101, n + m <= 1.101} (b = fl(v1 - a), c = fl(v0 - a) as ray_casting.py:35-36 computes them).
// With h = s - m, W = distance from m to the ray's line, phase 1 culls iff
//     (A)  0.995 |h|^2 - (h.d)^2 > r2          [ => W >= rho + 0.02 (|h| + 2 rho), rounding slack included ]
//     (B)  |n_dec . d| > 4e-3                   [ n_dec = the stored fp16 unit normal, within 1e-3 of N / |N|, N = b x c ]
// and the triangle is no sliver (|N| >= 0.05 |b| |c|; slivers, overflows and NaNs are stored as "always a candidate").
// The reference accepts iff fl(nn/det) >= -fp16(0.1), fl(mn/det) >= -fp16(0.1), fl(n + m) <= fp16(1.1)  (ray_casting.py:59),
// i.e. the point a + (nn/det) b + (mn/det) c lies in the padded triangle, hence within rho of m.  Multiplying by det and
// using the Cramer identity  nn* b + mn* c = det* g - kn* d  (exact triple products, g = s - a):
//     |det*| W - err  <=  rho (|det*| + err'),   err, err' <= 1e-6 |b||c| (2|g| + rho)   [f32 rounding of nn, mn, det: 16 eps]
// so with (A):  |det*| <= 2e-6 |b||c| / 0.02 = 1e-4 |b||c|.  But (B) and the sliver bound give
// |det*| = |N . d| >= (4e-3 - 1e-3) * 0.05 |b||c| = 1.5e-4 |b||c|  — a contradiction: the reference rejects.
// NaN / inf anywhere makes (A) or (B) compare false, i.e. keeps the triangle a candidate.  DESIGN.md §4.3 has the long form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rover_internal.h"
#include "rover_raymath.h"

namespace rover {

#define CULL_TAU2   1.6e-5f          // (4e-3)^2: guard threshold on the unit normal, folded into r2 = TAU2 * |n_stored|^2
#define CULL_TAU    4.0e-3
#define CULL_PAD    0.101            // barycentric padding of the proof (the reference's is fp16(0.1) = 0.09998)
#define CULL_QCAP   320              // queue entries per wave: < 64 left after a flush + at most 128 new ones per ray, with slack
#define CULL_RUNMAX 64               // sorted rays per wave (one result slot per lane)

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 cvt2(uint32_t d) {
    const half2v h = __builtin_bit_cast(half2v, d);
    return f2{(float)h.x, (float)h.y};
}

// r2 exactly as phase 1 derives it from the decoded normal (the table builder verifies its encoding with this)
__device__ __forceinline__ float cull_r2(float nx, float ny, float nz) {
    float q = nx * nx;
    q = __builtin_fmaf(ny, ny, q);
    q = __builtin_fmaf(nz, nz, q);
    return q * CULL_TAU2;
}

// ---------------------------------------------------------------------------------------------------
// init: per-cell reference point (cell centre in xy, mean triangle height in z) — the fp16 offsets of the cull table
// are relative to it
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cull_centre_kernel(const uint16_t* __restrict__ table, uint32_t n_cells, uint32_t K8,
                                                          uint32_t Y, float cell, float shift_x, float shift_y,
                                                          float4* __restrict__ cen) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    const _Float16* az = reinterpret_cast<const _Float16*>(table) + (size_t)c * 9u * K8 + 8u * (size_t)K8;
    float sum = 0.0f; uint32_t n = 0;
    for (uint32_t k = 0; k < K8; ++k) {
        const float z = (float)az[k];
        if (z == z && fabsf(z) < 6.0e4f) { sum += z; ++n; }
    }
    const uint32_t ix = c / Y, iy = c % Y;
    cen[c] = make_float4(shift_x + (float)ix * cell, shift_y + (float)iy * cell, n ? sum / (float)n : 0.0f, 0.0f);
}

__device__ __forceinline__ uint16_t half_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// init: one thread per (cell, slot) of the re-packed table -> the slot's 6 halves of the cull table
//   cull[cell][chunk 0..2][lane] = 16 B;  lane's 12 dwords D[p][q], p = pair, q = (ox, oy, oz, nx, ny, nz),
//   dword = half2{triangle 2p, triangle 2p+1} of the lane (slot 4 lane + 2p + e of the re-packed block)
__global__ void __launch_bounds__(256) cull_build_kernel(const uint16_t* __restrict__ table, uint64_t n_cells, uint32_t K8,
                                                         const float4* __restrict__ cen, uint16_t* __restrict__ cull) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells * K8) return;
    const uint64_t cell = i / K8;
    const uint32_t slot = (uint32_t)(i % K8), lane = slot >> 2, j = slot & 3u, p = j >> 1, e = j & 1u, L = K8 >> 2;
    const _Float16* src = reinterpret_cast<const _Float16*>(table) + cell * 9ull * K8 + slot;
    float v[9];
    bool valid = true;
#pragma unroll
    for (int q = 0; q < 9; ++q) { v[q] = (float)src[(size_t)q * K8]; valid = valid && (v[q] == v[q]) && fabsf(v[q]) < 6.0e4f; }
    const float4 cc = cen[cell];
    uint16_t out[6] = {0x7e00u, 0, 0, 0, 0, 0};                    // invalid slot: NaN offset (phase 1 masks it out)
    if (valid) {
        // a, b, c exactly as ray_casting.py:34-36 / set_pair compute them (f32), widened
        const float af[3] = {v[6], v[7], v[8]};
        const float bf[3] = {v[3] - v[6], v[4] - v[7], v[5] - v[8]};
        const float cf[3] = {v[0] - v[6], v[1] - v[7], v[2] - v[8]};
        double Q[3][3];                                              // corners of the padded triangle
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = af[k], b = bf[k], c = cf[k];
            Q[0][k] = a - CULL_PAD * b - CULL_PAD * c;
            Q[1][k] = a + (1.0 + 2.0 * CULL_PAD) * b - CULL_PAD * c;
            Q[2][k] = a - CULL_PAD * b + (1.0 + 2.0 * CULL_PAD) * c;
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
        const float ccf[3] = {cc.x, cc.y, cc.z};
        float mk[3];                                                 // the centre as phase 1 decodes it
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double m = w0 * Q[0][k] + w1 * Q[1][k] + w2 * Q[2][k];
            float off = (float)(m - (double)ccf[k]);
            if (!(fabsf(off) < 6.0e4f)) { off = 0.0f; ok = false; }
            const _Float16 oh = (_Float16)off;
            out[k] = __builtin_bit_cast(uint16_t, oh);
            mk[k] = ccf[k] + (float)oh;                              // = CullRegs::m in raycast_culled_kernel
        }
        double rho2 = 0.0;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (Q[x][k] - (double)mk[k]) * (Q[x][k] - (double)mk[k]);
            rho2 = s > rho2 ? s : rho2;
        }
        const double rho = sqrt(rho2) + 1.0e-4;
        const double need = 1.19 * rho * rho;
        const double N[3] = {(double)bf[1] * cf[2] - (double)bf[2] * cf[1], (double)bf[2] * cf[0] - (double)bf[0] * cf[2],
                             (double)bf[0] * cf[1] - (double)bf[1] * cf[0]};
        const double nN = sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        const double nb = sqrt((double)bf[0] * bf[0] + (double)bf[1] * bf[1] + (double)bf[2] * bf[2]);
        const double nc = sqrt((double)cf[0] * cf[0] + (double)cf[1] * cf[1] + (double)cf[2] * cf[2]);
        ok = ok && nN > 0.0 && nN >= 0.05 * nb * nc;                 // slivers stay candidates
        if (ok) {
            double scale = sqrt(need) * 1.002 / CULL_TAU / nN;       // |stored normal| = r / tau
            bool done = false;
            for (int it = 0; it < 8 && !done; ++it, scale *= 1.002) {
                float dec[3];
                bool fin = true;
                for (int k = 0; k < 3; ++k) {
                    const float f = (float)(N[k] * scale);
                    fin = fin && fabsf(f) < 6.0e4f;
                    const _Float16 hn = (_Float16)f;
                    out[3 + k] = __builtin_bit_cast(uint16_t, hn);
                    dec[k] = (float)hn;
                }
                if (!fin) break;
                done = (double)cull_r2(dec[0], dec[1], dec[2]) >= need;
            }
            ok = done;
        }
        if (!ok) out[3] = out[4] = out[5] = 0;                       // zero normal: guard (B) never holds
    }
    uint16_t* dst = cull + (cell * 3ull * L + lane) * 8ull;          // halves; chunk c of the lane at + c * L * 8
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const uint32_t h = (p * 6u + (uint32_t)q) * 2u + e;          // index among the lane's 24 halves
        dst[(uint64_t)(h >> 3) * L * 8ull + (h & 7u)] = out[q];
    }
}

// ---------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------
struct CullRegs { f2 mx[2], my[2], mz[2], nx[2], ny[2], nz[2], r2[2]; };     // per lane: 4 triangles as 2 packed pairs

// order-preserving f32 -> u32 key with -0 < +0, the order v_min_f32 gives the other kernels' reductions
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

// LDS traffic between the lanes of ONE wave: the hardware keeps a wave's LDS operations in order; this only stops the
// compiler from moving them across
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// PHASE 2: n <= 64 queue entries, one per lane: the exact arithmetic on the entry's pair of triangles, min into the ray's slot
__device__ __forceinline__ void cull_flush(const RayRec* __restrict__ rays, const uint32_t* __restrict__ sorted,
                                           const _Float16* __restrict__ tab0, const _Float16* __restrict__ tab1, uint32_t kp0,
                                           uint32_t kp1, const uint2* qe, uint32_t n, uint32_t lane, uint32_t i0, uint32_t* bk) {
    if (lane < n) {
        const uint2 en = qe[lane];
        const uint32_t cell = en.x & 0x7fffffffu, map = en.x >> 31;
        const uint32_t pos = en.y >> 8, el = (en.y >> 1) & 63u, p = en.y & 1u;
        const uint32_t kp = map ? kp1 : kp0;
        const _Float16* base = (map ? tab1 : tab0) + (size_t)cell * 9u * kp + el * 4u + p * 2u;        // el * 4 < kp
        f2 v[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) v[q] = cvt2(*reinterpret_cast<const uint32_t*>(base + (size_t)q * kp));
        CellRegs<1> t;
        set_pair(t, 0, v);
        const uint32_t gid = sorted[i0 + pos];
        const float4* rp = reinterpret_cast<const float4*>(rays + gid);
        const float4 ra = rp[0], rb = rp[1];
        const uint64_t none[1][2] = {{0, 0}};
        const float best = cast_pairs<1>(t, f2{ra.x, ra.x}, f2{ra.y, ra.y}, f2{ra.z, ra.z}, f2{rb.x, rb.x}, f2{rb.y, rb.y},
                                         f2{rb.z, rb.z}, none, 0u);
        atomicMin(bk + pos, fkey(best));
    }
}

// WPE: waves per SIMD the register allocation aims at (0 = the compiler's own choice) — option "cull_waves", A/B only
template <int WPE>
__device__ __forceinline__ void raycast_culled_body(const RayRec* __restrict__ rays, const uint32_t* __restrict__ sorted,
                                                             uint32_t n_sorted, const _Float16* __restrict__ tab0,
                                                             const _Float16* __restrict__ tab1, const uint4* __restrict__ cull0,
                                                             const uint4* __restrict__ cull1, const float4* __restrict__ cen0,
                                                             const float4* __restrict__ cen1, uint32_t kp0, uint32_t kp1,
                                                             uint32_t run, uint32_t n_blocks, uint32_t nb8, float* __restrict__ out) {
    __shared__ uint2 s_queue[4][CULL_QCAP];
    __shared__ uint32_t s_best[4][CULL_RUNMAX];
    // XCD-aware order, as raycast_binned_kernel: each XCD walks one contiguous eighth of the sorted rays
    const uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    if (lb >= n_blocks) return;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(lb * 4u + w);
    const uint32_t i0 = wave * run;
    if (i0 >= n_sorted) return;
    const uint32_t i_end = min(i0 + run, n_sorted);
    uint2* q = s_queue[w];
    uint32_t* bk = s_best[w];
    bk[lane] = fkey(RAY_MISS);                 // a culled triangle contributes the 11.0 sentinel (ray_casting.py:27,59)
    uint32_t qn = 0;
    uint32_t cur_cell = 0xffffffffu, cur_map = 0xffffffffu, el = lane;
    CullRegs t;
#pragma unroll
    for (int p = 0; p < 2; ++p) t.mx[p] = t.my[p] = t.mz[p] = t.nx[p] = t.ny[p] = t.nz[p] = t.r2[p] = f2{0.0f, 0.0f};
    for (uint32_t i = i0; i < i_end; ++i) {
        const uint32_t gid = __builtin_amdgcn_readfirstlane(sorted[i]);
        const float4* rp = reinterpret_cast<const float4*>(rays + gid);
        const float4 ra = rp[0], rb = rp[1];
        const uint32_t cell = __builtin_amdgcn_readfirstlane(__float_as_uint(ra.w));
        const uint32_t map = __builtin_amdgcn_readfirstlane(__float_as_uint(rb.w)) & 1u;
        bool change = cell != cur_cell || map != cur_map;
        // Flush full batches of 64 only where the cell registers are dead (before a set-up); mid-cell only when the queue
        // could overflow, and then the cell is set up again.
        if (qn > CULL_QCAP - 128u || (change && qn >= 64u)) {
            wave_lds_sync();
            do {
                qn -= 64u;
                cull_flush(rays, sorted, tab0, tab1, kp0, kp1, q + qn, 64u, lane, i0, bk);
            } while (qn >= 64u);
            wave_lds_sync();
            change = true;
        }
        if (change) {
            cur_cell = cell; cur_map = map;
            const uint32_t L = (map ? kp1 : kp0) >> 2;
            const float4 cc = (map ? cen1 : cen0)[cell];
            // lanes past K (K8 < 256) repeat the last lane's triangles: no divergent set-up (a divergent one keeps the old
            // cell's registers alive across the flush above), and a duplicate candidate cannot change a min
            el = lane < L ? lane : L - 1u;
            const uint4* cb = (map ? cull1 : cull0) + (size_t)cell * 3u * L + el;
            const uint4 c0 = cb[0], c1 = cb[L], c2 = cb[2u * L];
            const uint32_t D[2][6] = {{c0.x, c0.y, c0.z, c0.w, c1.x, c1.y}, {c1.z, c1.w, c2.x, c2.y, c2.z, c2.w}};
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                t.mx[p] = f2{cc.x, cc.x} + cvt2(D[p][0]);
                t.my[p] = f2{cc.y, cc.y} + cvt2(D[p][1]);
                t.mz[p] = f2{cc.z, cc.z} + cvt2(D[p][2]);
                t.nx[p] = cvt2(D[p][3]); t.ny[p] = cvt2(D[p][4]); t.nz[p] = cvt2(D[p][5]);
                f2 s = t.nx[p] * t.nx[p];
                s = fma2(t.ny[p], t.ny[p], s);
                s = fma2(t.nz[p], t.nz[p], s);
                s = s * f2{CULL_TAU2, CULL_TAU2};                  // = cull_r2()
                // an empty slot (NaN x offset, zero normal) is never a candidate: finite centre, r2 = -inf
                const bool e0 = t.mx[p].x == t.mx[p].x, e1 = t.mx[p].y == t.mx[p].y;
                t.mx[p] = f2{e0 ? t.mx[p].x : 0.0f, e1 ? t.mx[p].y : 0.0f};
                t.r2[p] = f2{e0 ? s.x : -__builtin_inff(), e1 ? s.y : -__builtin_inff()};
            }
        }
        // PHASE 1: lanes whose pair p holds a triangle that (A) and (B) do not both reject
        const f2 sx = {ra.x, ra.x}, sy = {ra.y, ra.y}, sz = {ra.z, ra.z};
        const f2 dx = {rb.x, rb.x}, dy = {rb.y, rb.y}, dz = {rb.z, rb.z};
        uint64_t any[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f2 hx = sx - t.mx[p], hy = sy - t.my[p], hz = sz - t.mz[p];
            f2 hd = hx * dx; hd = fma2(hy, dy, hd); hd = fma2(hz, dz, hd);
            f2 hh = hx * hx; hh = fma2(hy, hy, hh); hh = fma2(hz, hz, hh);
            const f2 A = fma2(hh, f2{0.995f, 0.995f}, -(hd * hd));                     // (A): 0.995 |h|^2 - (h.d)^2 > r2
            f2 Dn = t.nx[p] * dx; Dn = fma2(t.ny[p], dy, Dn); Dn = fma2(t.nz[p], dz, Dn);
            const f2 B = Dn * Dn;                                                       // (B): (n.d)^2 > tau^2 |n|^2 = r2
            // one ballot per compare (each stays a v_cmp writing an SGPR pair); NaN compares false = stays a candidate
            const uint64_t rej0 = __builtin_amdgcn_ballot_w64(A.x > t.r2[p].x) & __builtin_amdgcn_ballot_w64(B.x > t.r2[p].x);
            const uint64_t rej1 = __builtin_amdgcn_ballot_w64(A.y > t.r2[p].y) & __builtin_amdgcn_ballot_w64(B.y > t.r2[p].y);
            any[p] = ~(rej0 & rej1);                                                    // (all 64 lanes are active here)
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (any[p]) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(any[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any[p], 0u));
                if (__builtin_amdgcn_inverse_ballot_w64(any[p]))
                    q[qn + rank] = make_uint2(cell | (map << 31), ((i - i0) << 8) | (el << 1) | (uint32_t)p);
                qn += (uint32_t)__builtin_popcountll(any[p]);
            }
        }
    }
    wave_lds_sync();
    while (qn) {
        const uint32_t n = qn < 64u ? qn : 64u;
        qn -= n;
        cull_flush(rays, sorted, tab0, tab1, kp0, kp1, q + qn, n, lane, i0, bk);
    }
    wave_lds_sync();
    if (i0 + lane < i_end) out[sorted[i0 + lane]] = funkey(bk[lane]);
}

#define CULL_KERNEL_ARGS                                                                                                       \
    const RayRec *__restrict__ rays, const uint32_t *__restrict__ sorted, uint32_t n_sorted, const _Float16 *__restrict__ tab0, \
        const _Float16 *__restrict__ tab1, const uint4 *__restrict__ cull0, const uint4 *__restrict__ cull1,                   \
        const float4 *__restrict__ cen0, const float4 *__restrict__ cen1, uint32_t kp0, uint32_t kp1, uint32_t run,            \
        uint32_t n_blocks, uint32_t nb8, float *__restrict__ out
#define CULL_KERNEL_PASS rays, sorted, n_sorted, tab0, tab1, cull0, cull1, cen0, cen1, kp0, kp1, run, n_blocks, nb8, out
__global__ void __launch_bounds__(256) raycast_culled_kernel(CULL_KERNEL_ARGS) { raycast_culled_body<0>(CULL_KERNEL_PASS); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) raycast_culled_w6_kernel(CULL_KERNEL_ARGS) {
    raycast_culled_body<6>(CULL_KERNEL_PASS);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7))) raycast_culled_w7_kernel(CULL_KERNEL_ARGS) {
    raycast_culled_body<7>(CULL_KERNEL_PASS);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) raycast_culled_w8_kernel(CULL_KERNEL_ARGS) {
    raycast_culled_body<8>(CULL_KERNEL_PASS);
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint64_t n, uint32_t bs) { return (uint32_t)((n + bs - 1) / bs); }

hipError_t launch_cull_build(const uint16_t* table, uint64_t n_cells, uint32_t K8, uint32_t Y, float cell, float shift_x,
                             float shift_y, float4* cen, uint16_t* cull, hipStream_t s) {
    hipLaunchKernelGGL(cull_centre_kernel, dim3(blocks_for(n_cells, 256)), dim3(256), 0, s, table, (uint32_t)n_cells, K8, Y, cell,
                       shift_x, shift_y, cen);
    hipLaunchKernelGGL(cull_build_kernel, dim3(blocks_for(n_cells * K8, 256)), dim3(256), 0, s, table, n_cells, K8, cen, cull);
    return hipGetLastError();
}

hipError_t launch_raycast_culled(CullArgs a, hipStream_t s) {
    if (a.run > CULL_RUNMAX) a.run = CULL_RUNMAX;
    if (a.run == 0) a.run = 1;
    const uint32_t n_waves = blocks_for(a.n_sorted, a.run);
    a.n_blocks = blocks_for(n_waves, 4);
    a.nb8 = blocks_for(a.n_blocks, 8);
    auto kern = a.waves == 6 ? raycast_culled_w6_kernel : a.waves == 7 ? raycast_culled_w7_kernel
              : a.waves == 8 ? raycast_culled_w8_kernel : raycast_culled_kernel;
    hipLaunchKernelGGL(kern, dim3(a.nb8 * 8u), dim3(256), 0, s, a.rays, a.sorted, a.n_sorted,
                       reinterpret_cast<const _Float16*>(a.tab0), reinterpret_cast<const _Float16*>(a.tab1), a.cull0, a.cull1,
                       a.cen0, a.cen1, a.kp0, a.kp1, a.run, a.n_blocks, a.nb8, a.out);
    return hipGetLastError();
}

}  // namespace rover
