// rover_raymath.h — the (ray, triangle) arithmetic of ray_casting.py:31-59 shared by the binned ray-cast kernels
// (rover_kernels.hip) and the culled ray cast (rover_cull.hip).  One definition, so that every kernel produces the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rover {

#define RAY_NEG_EPS (-0.0999755859375f)   // fp16(-0.1), ray_casting.py:25
#define RAY_ONE_EPS (1.099609375f)        // fp16(1.1),  ray_casting.py:26
#define RAY_MISS    (11.0f)               // fp16(1.1)*10, ray_casting.py:27

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));       // two triangles side by side -> v_pk_{mul,add,fma}_f32

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct Quot3 { f2 n, m, k; };

// n = nn/det, m = mn/det, k = kn/det for two triangles at once.  Per element this is the instruction sequence of
// the IEEE-754 f32 division expansion (rcp, two Newton steps on the reciprocal, quotient, two residual
// corrections) minus its range scaling / fix-up, which only act when |det| or a quotient is near the ends of
// the f32 range — and every such case fails the barycentric test below either way.
__device__ __forceinline__ Quot3 div3_ieee(f2 det, f2 nn, f2 mn, f2 kn) {
    f2 r = {__builtin_amdgcn_rcpf(det.x), __builtin_amdgcn_rcpf(det.y)};
    const f2 one = {1.0f, 1.0f};
    f2 e = fma2(-det, r, one);
    r = fma2(e, r, r);
    Quot3 q;
#define ROVER_Q(num, dst)                 \
    {                                     \
        f2 t = (num) * r;                 \
        f2 rem = fma2(-det, t, (num));    \
        t = fma2(rem, r, t);              \
        rem = fma2(-det, t, (num));       \
        dst = fma2(rem, r, t);            \
    }
    ROVER_Q(nn, q.n) ROVER_Q(mn, q.m) ROVER_Q(kn, q.k)
#undef ROVER_Q
    return q;
}

// ray_casting.py:59 with the :46,:51,:56 substitutions folded in: det == fp16(-0.1) forces n = 11 and det == fp16(1.1)
// forces m = k = 11, either of which fails n + m <= 1.1.
__device__ __forceinline__ float accept1(float n, float m, float k, float det, float n_plus_m) {
    bool ok = (n >= RAY_NEG_EPS) && (m >= RAY_NEG_EPS) && (n_plus_m <= RAY_ONE_EPS)
              && (det != RAY_NEG_EPS) && (det != RAY_ONE_EPS);
    return ok ? k : RAY_MISS;
}
__device__ __forceinline__ float accept1(float n, float m, float k, float det) { return accept1(n, m, k, det, n + m); }

// min over the 64 lanes with DPP row operations (no LDS crossbar): result valid in lane 63.
// One asm block so the DPP read-after-VALU-write wait states (2, "s_nop 1") are under our control.
__device__ __forceinline__ float wave_min_to_lane63(float v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// per-lane triangle pairs of one cell: a = v2, b = v1 - a, c = v0 - a, n = b x c (ray_casting.py:34-36,40)
template <int NP>
struct CellRegs {
    f2 ax[NP], ay[NP], az[NP], bx[NP], by[NP], bz[NP], cx[NP], cy[NP], cz[NP], nx[NP], ny[NP], nz[NP];
    __device__ __forceinline__ void poison() {       // lanes past K: NaN vertex -> every test fails
        const float qnan = __builtin_nanf("");
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ax[p] = ay[p] = az[p] = f2{qnan, qnan};
            bx[p] = by[p] = bz[p] = cx[p] = cy[p] = cz[p] = nx[p] = ny[p] = nz[p] = f2{0.0f, 0.0f};
        }
    }
};

// Distances as ordered-u32 keys (the culled ray cast reduces with an integer LDS atomicMin): order-preserving f32 -> u32
// with -0 < +0, the order v_min_f32 gives the other kernels' reductions.
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
// one packed pair from its nine vertex components (component q = 3 * vertex + coord, two triangles side by side)
template <int NP>
__device__ __forceinline__ void set_pair(CellRegs<NP>& t, int p, const f2 (&v)[9]) {
    t.ax[p] = v[6]; t.ay[p] = v[7]; t.az[p] = v[8];                      // a = v2
    t.bx[p] = v[3] - t.ax[p]; t.by[p] = v[4] - t.ay[p]; t.bz[p] = v[5] - t.az[p];      // b = v1 - a
    t.cx[p] = v[0] - t.ax[p]; t.cy[p] = v[1] - t.ay[p]; t.cz[p] = v[2] - t.az[p];      // c = v0 - a
    t.nx[p] = t.by[p] * t.cz[p] - t.bz[p] * t.cy[p];                                   // b x c
    t.ny[p] = t.bz[p] * t.cx[p] - t.bx[p] * t.cz[p];
    t.nz[p] = t.bx[p] * t.cy[p] - t.by[p] * t.cx[p];
}

// the ray-dependent part of ray_casting.py:37-59 for the lane's NP pairs; returns the lane's min distance
//
// Whole-pair early out (bit p of pre_bits): before the divisions, every lane tests its two triangles of pair p with
//     A = nn det, B = mn det, D = det^2:   A < -0.11 D - tiny   or   B < -0.11 D - tiny   or   A + B > 1.11 D + tiny
// Any of these proves (with a margin of 0.01 against f32 rounding errors of ~1e-7, and tiny = 1e-30 against the
// absolute errors of the denormal range) that n < -0.1, m < -0.1 or n + m > 1.1 holds for the exactly rounded quotients
// too, i.e. that ray_casting.py:59 rejects the triangle; NaN / det = 0 never pass the test.  When ALL valid triangles
// of the pair are rejected in every lane the wave skips the pair's k numerator, the three divisions and the accept
// logic — the result is unchanged bit for bit, since a rejected triangle only contributes the 11.0 sentinel.
// vmask[p][e]: wave mask of the lanes whose element e of pair p is a real triangle (not NaN padding).
template <int NP>
__device__ __forceinline__ float cast_pairs(const CellRegs<NP>& t, f2 sx, f2 sy, f2 sz, f2 dx, f2 dy, f2 dz,
                                            const uint64_t (&vmask)[NP][2], uint32_t pre_bits) {
    float best = RAY_MISS;            // every cell holds >= 1 real triangle, so the min is <= 11 (ray_casting.py:27)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        f2 gx = sx - t.ax[p], gy = sy - t.ay[p], gz = sz - t.az[p];                                          // :37
        f2 det = t.nx[p] * dx + t.ny[p] * dy + t.nz[p] * dz;                                                 // :41
        f2 gcx = gy * t.cz[p] - gz * t.cy[p], gcy = gz * t.cx[p] - gx * t.cz[p], gcz = gx * t.cy[p] - gy * t.cx[p];
        f2 nn = gcx * dx + gcy * dy + gcz * dz;                                                              // :44-45
        f2 bgx = t.by[p] * gz - t.bz[p] * gy, bgy = t.bz[p] * gx - t.bx[p] * gz, bgz = t.bx[p] * gy - t.by[p] * gx;
        f2 mn = bgx * dx + bgy * dy + bgz * dz;                                                              // :49-50
        if (pre_bits & (1u << p)) {                                                                          // wave-uniform
            const f2 D = det * det, A = nn * det, B = mn * det, S = A + B;
            const f2 lo = fma2(f2{-0.11f, -0.11f}, D, f2{-1e-30f, -1e-30f});
            const f2 hi = fma2(f2{1.11f, 1.11f}, D, f2{1e-30f, 1e-30f});
            // one ballot per compare so that each stays a v_cmp writing an SGPR pair; the masks combine on the scalar unit
            const uint64_t r0 = __builtin_amdgcn_ballot_w64(__builtin_fminf(A.x, B.x) < lo.x) | __builtin_amdgcn_ballot_w64(S.x > hi.x);
            const uint64_t r1 = __builtin_amdgcn_ballot_w64(__builtin_fminf(A.y, B.y) < lo.y) | __builtin_amdgcn_ballot_w64(S.y > hi.y);
            if (((~r0 & vmask[p][0]) | (~r1 & vmask[p][1])) == 0) continue;
        }
        f2 kn = t.nx[p] * gx + t.ny[p] * gy + t.nz[p] * gz;                                                  // :54-55
        Quot3 q = div3_ieee(det, nn, mn, kn);
        float r0 = accept1(q.n.x, q.m.x, q.k.x, det.x);
        float r1 = accept1(q.n.y, q.m.y, q.k.y, det.y);
        best = __builtin_fminf(best, __builtin_fminf(r0, r1));       // no NaN can reach here (accept1 filters)
    }
    return best;
}

// ---------------------------------------------------------------------------------------------------
// The same arithmetic as ATen evaluates it on float16 tensors (the reference AS SHIPPED, Camera.dtype = float16): every
// elementwise op of ray_casting.py:31-59 rounds its result to fp16 (packed fp16 instructions, no contraction); the three
// quotients are taken in f32 (IEEE, shared reciprocal) and rounded to fp16, which equals the fp16 quotient (24 >= 2 * 11 + 2 bits).
// Bit-identical to the oracle's fp16 mode, which the as-shipped golden fixtures pin bit for bit.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int NP>
struct CellRegsH {
    h2 ax[NP], ay[NP], az[NP], bx[NP], by[NP], bz[NP], cx[NP], cy[NP], cz[NP], nx[NP], ny[NP], nz[NP];
    __device__ __forceinline__ void poison() {
        const _Float16 qnan = (_Float16)__builtin_nanf("");
        const _Float16 zero = (_Float16)0.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ax[p] = ay[p] = az[p] = h2{qnan, qnan};
            bx[p] = by[p] = bz[p] = cx[p] = cy[p] = cz[p] = nx[p] = ny[p] = nz[p] = h2{zero, zero};
        }
    }
};

__device__ __forceinline__ f2 h2_to_f2(h2 v) { return f2{(float)v.x, (float)v.y}; }

// one packed pair from its nine fp16 vertex components: a = v2, b = v1 - a, c = v0 - a, n = b x c, each op rounded to fp16
template <int NP>
__device__ __forceinline__ void set_pair_h(CellRegsH<NP>& t, int p, const h2 (&v)[9]) {
    t.ax[p] = v[6]; t.ay[p] = v[7]; t.az[p] = v[8];
    t.bx[p] = v[3] - t.ax[p]; t.by[p] = v[4] - t.ay[p]; t.bz[p] = v[5] - t.az[p];
    t.cx[p] = v[0] - t.ax[p]; t.cy[p] = v[1] - t.ay[p]; t.cz[p] = v[2] - t.az[p];
    t.nx[p] = t.by[p] * t.cz[p] - t.bz[p] * t.cy[p];
    t.ny[p] = t.bz[p] * t.cx[p] - t.bx[p] * t.cz[p];
    t.nz[p] = t.bx[p] * t.cy[p] - t.by[p] * t.cx[p];
}

// Early out as in cast_pairs, on the fp16 numerators the reference's own arithmetic produces: A = nn det, B = mn det and
// D = det^2 are EXACT in f32 (11-bit x 11-bit significands), so A < -0.15 D - tiny proves nn/det < -0.1499, whose fp16
// rounding is < fp16(-0.1); A + B > 1.15 D + tiny proves x + y > 1.1499 for the exact quotients x, y, and then either
// fp16(fp16(x) + fp16(y)) > fp16(1.1) (|x|, |y| <= 32: the roundings move the sum by < 0.032) or one quotient is < -1.
// Infinities / NaN from fp16 overflow never pass the test (and the reference rejects them too).
template <int NP>
__device__ __forceinline__ float cast_pairs_h(const CellRegsH<NP>& t, h2 sx, h2 sy, h2 sz, h2 dx, h2 dy, h2 dz,
                                              const uint64_t (&vmask)[NP][2], uint32_t pre_bits) {
    float best = RAY_MISS;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        h2 gx = sx - t.ax[p], gy = sy - t.ay[p], gz = sz - t.az[p];
        h2 det = t.nx[p] * dx + t.ny[p] * dy + t.nz[p] * dz;
        h2 gcx = gy * t.cz[p] - gz * t.cy[p], gcy = gz * t.cx[p] - gx * t.cz[p], gcz = gx * t.cy[p] - gy * t.cx[p];
        h2 nn = gcx * dx + gcy * dy + gcz * dz;
        h2 bgx = t.by[p] * gz - t.bz[p] * gy, bgy = t.bz[p] * gx - t.bx[p] * gz, bgz = t.bx[p] * gy - t.by[p] * gx;
        h2 mn = bgx * dx + bgy * dy + bgz * dz;
        const f2 detf = h2_to_f2(det), nnf = h2_to_f2(nn), mnf = h2_to_f2(mn);
        if (pre_bits & (1u << p)) {
            const f2 D = detf * detf, A = nnf * detf, B = mnf * detf, S = A + B;
            const f2 lo = fma2(f2{-0.15f, -0.15f}, D, f2{-1e-30f, -1e-30f});
            const f2 hi = fma2(f2{1.15f, 1.15f}, D, f2{1e-30f, 1e-30f});
            const uint64_t r0 = __builtin_amdgcn_ballot_w64(__builtin_fminf(A.x, B.x) < lo.x) | __builtin_amdgcn_ballot_w64(S.x > hi.x);
            const uint64_t r1 = __builtin_amdgcn_ballot_w64(__builtin_fminf(A.y, B.y) < lo.y) | __builtin_amdgcn_ballot_w64(S.y > hi.y);
            if (((~r0 & vmask[p][0]) | (~r1 & vmask[p][1])) == 0) continue;
        }
        h2 kn = t.nx[p] * gx + t.ny[p] * gy + t.nz[p] * gz;
        Quot3 q = div3_ieee(detf, nnf, mnf, h2_to_f2(kn));
        h2 n = h2{(_Float16)q.n.x, (_Float16)q.n.y}, m = h2{(_Float16)q.m.x, (_Float16)q.m.y}, k = h2{(_Float16)q.k.x, (_Float16)q.k.y};
        h2 nm = n + m;                                                            // fp16 sum, then compared (ray_casting.py:59)
        float r0 = accept1((float)n.x, (float)m.x, (float)k.x, detf.x, (float)nm.x);
        float r1 = accept1((float)n.y, (float)m.y, (float)k.y, detf.y, (float)nm.y);
        best = __builtin_fminf(best, __builtin_fminf(r0, r1));
    }
    return best;
}

}  // namespace rover
