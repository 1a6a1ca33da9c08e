// rover_walk.hip — the walked ray cast (raycast variant 4): one LANE per ray, each walking only the part of its cell's triangle
// list that it cannot prove clear as a group.
//
// Reference: tasks/utils/camera/camera.py:60-145 (gather K triangles per ray, ray_distance, min over K),
//            tasks/utils/camera/ray_casting.py:31-59 (the (ray, triangle) test), rock_detect.py:52-149 (same on the rocks map).
//
// The culled ray cast (rover_cull.hip, variant 3) gives a wave one (map, cell) bin at a time: 64 lanes hold the bin's 200 sphere /
// normal records, the bin's rays pass through them one after the other.  Its cost per bin — id row, 2-4 record gathers per lane,
// their unpacking, one exposed gather round trip — does not depend on how many triangles a ray can reach, and with ~8 rays per bin
// it was more than half of the kernel.  Here the mapping is transposed: a lane IS a ray, and per cell the K
// records are stored in the order a ray needs them:
//
//   wrec [cell][K]  16 B: {centre x, centre y (f32), half2(centre z, r2 rounded UP), triangle id | code6 << 26} — the bounding
//                   sphere of the padded triangle exactly as rover_cull.hip's ctab holds it (same decoded centre, r2 >= ctab's r2)
//                   and a 6-bit code of test (B)'s per-triangle bound (below).  Order: the always-candidates first (slivers,
//                   non-finite vertices: no test can reject them), then by ascending G = dist_xy(centre, cell centre) - k1 sqrt(r2):
//                   far_build_kernel's group-bound key.
//   wlvl [cell][12] 16 B: level l = {G, z0, z1, half(rho_out) | cnt << 16 | minc << 25}: a prefix length cnt_l (always-candidates +
//                   0, 16, 24, ... entries) and the group bound of the COMPLEMENT [cnt_l, n): the smallest G, the z range, the largest
//                   dist_xy of its centres and its weakest cone code.  Record 11 = {cell centre x, y, -, n << 16}.
//
// A ray evaluates the group inequality of rover_cull.hip's far skip (far_build_kernel has the derivation: if it holds, test (A)
// holds for EVERY triangle of the set) level by level and walks the first prefix whose complement it clears: 0 entries for most
// rock rays, 30-60 of 200 for a heightmap ray on the bench scene.  Every walked entry gets tests (A) and (B) in 16 plain f32 / integer
// instructions; what they do not reject is a candidate (one 4-byte LDS queue entry: triangle id | run position << 26) for the exact
// arithmetic of rover_raymath.h, run by the same wave after its walk, one lane per (ray, triangle).
// The 64 rays of a wave need prefixes of different lengths, and a wave walks as long as its longest: so a lane walks at most `cap`
// entries itself (cap: all but <= 16 of the wave's rays need no more), and what is left of the long ones — and all of a ray no level
// applies to: the two horizontal body rays of every rover — is walked by the whole WAVE afterwards, 64 entries of one ray per trip.
//
// Soundness.  A triangle contributes nothing but the 11.0 sentinel iff ray_casting.py:59 rejects it; rover_cull.hip's header proves
// that (A) and (B) together imply the rejection.  (A) is evaluated here with the same centre and an r2 that is not smaller (fp16,
// rounded up): it holds less often, never more.  (B) — |n . d| > tau |n| — is first tried in its cone form, the bound rover_cull.hip
// uses per CELL, per triangle: with q_t = |N_z| / |N| (f32 proof) or gamma_t (fp16 proof: an angle), ctab_build_kernel's nz_abs, the
// ray's own 16-bit bound (RayRec.flags >> 16, prep_rays_kernel) and code16_t = floor(65535 q_t): (B) holds if code16_t >= ray16.
// The record keeps code6 = code16_t >> 10 and the test is code6 * 1024 >= ray16: it holds less often, never more.  Where it fails
// the wave-walked part evaluates (B) itself from the triangle's ctab record, exactly as cull_scan_kernel does; the lane-walked part
// keeps the triangle as a candidate.  Skipped entries (the complement of the walked prefix) have code6 >= minc, and a ray only
// skips when minc * 1024 >= ray16 — so (B) holds for each of them — and when the group inequality holds with the complement's own G,
// z range and rho_out — so (A) does.  NaN anywhere compares false: the ray walks everything / the triangle stays a candidate.
// Results are bit-identical to the other ray-cast kernels (tests/test_hip_parity.py, tools/soak_exact.py): the exact phase is the
// one arithmetic (rover_raymath.h: cast_one = cast_pairs element for element), and a min over a superset of the hits is the min.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rover_internal.h"
#include "rover_raymath.h"

namespace rover {

#define WALK_LEVELS 12
#define WALK_NOID 0x3ffffffu
// prefix lengths behind the front entries, levels 0..10 (level 11 = everything)
__constant__ uint32_t c_walk_off[WALK_LEVELS - 1] = {0, 16, 24, 32, 40, 48, 56, 64, 80, 104, 144};

typedef _Float16 half2w __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float half_lo(uint32_t d) { return (float)__builtin_bit_cast(half2w, d).x; }
__device__ __forceinline__ float half_hi(uint32_t d) { return (float)__builtin_bit_cast(half2w, d).y; }

// the smallest fp16 value >= f (f >= 0 or NaN; NaN / overflow -> +inf), as bits
__device__ __forceinline__ uint32_t half_up_bits(float f) {
    if (!(f < 65504.0f)) return 0x7c00u;
    _Float16 h = (_Float16)f;
    uint16_t b = __builtin_bit_cast(uint16_t, h);
    if ((float)h < f) b = (uint16_t)(b + 1u);             // next larger (positive values: the bit pattern is monotone)
    return (uint32_t)b;
}

__device__ __forceinline__ float walk_r2(float nx, float ny, float nz, float tau2) {      // = cull_r2 (rover_cull.hip)
    float q = nx * nx;
    q = __builtin_fmaf(ny, ny, q);
    q = __builtin_fmaf(nz, nz, q);
    return q * tau2;
}

// ---------------------------------------------------------------------------------------------------
// init: one workgroup per cell (K <= 256).  Runs right after ctab_build_kernel / idx4_build_kernel of the same proof (nz_abs valid).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) walk_build_kernel(const int32_t* __restrict__ map_idx, uint32_t K, uint32_t T,
                                                         const uint32_t* __restrict__ newid, const uint4* __restrict__ ctab,
                                                         const float* __restrict__ nz_abs, float tau2, float k1, uint32_t c0, uint32_t Y,
                                                         float cell_size, float shift_x, float shift_y, uint4* __restrict__ wrec,
                                                         uint4* __restrict__ wlvl, uint32_t* __restrict__ counts) {
    __shared__ unsigned long long key[256];
    __shared__ float s_G[256], s_z[256], s_d[256];
    __shared__ uint4 s_rec[256];
    __shared__ uint32_t s_n[2];
    const uint32_t cell = blockIdx.x, tid = threadIdx.x;
    const float ccx = (float)(cell / Y) * cell_size + shift_x, ccy = (float)(cell % Y) * cell_size + shift_y;     // = far_build_kernel's centre
    uint32_t id = 0xffffffffu;
    if (tid < K) { const uint32_t t = (uint32_t)map_idx[(uint64_t)cell * K + tid]; if (t < T) id = newid[t]; }
    uint32_t hi = 0xffffffffu;                            // sort class: 0 front, ordered(G) >= 1 behind it, ~0 empty slot
    float G = __builtin_inff(), mz = 0.0f, dxy = 0.0f;
    uint4 rec = make_uint4(0u, 0u, 0xfc000000u /* r2 = -inf: never a candidate */, WALK_NOID);
    if (id != 0xffffffffu) {
        const uint4 r = ctab[id];
        const float nx = half_hi(r.z), ny = half_lo(r.w), nz = half_hi(r.w);
        const float mx = __uint_as_float(r.x), my = __uint_as_float(r.y);
        mz = half_lo(r.z);
        const float r2 = walk_r2(nx, ny, nz, tau2);
        const float q = nz_abs[id];                       // |N_z| / |N| (f32 proof) / gamma (fp16 proof), 2.0 = always a candidate
        bool always = !(r2 < 3.0e38f) || !(q <= 1.5f);
        uint32_t code16 = 0;
        if (!always && q > 0.0f) { code16 = (uint32_t)floorf(fminf(q, 1.0f) * 65535.0f); code16 = code16 > 0xfffeu ? 0xfffeu : code16; }
        uint32_t code6 = code16 >> 10;
        uint32_t r2h = half_up_bits(r2);
        if (always) { r2h = 0x7c00u; code6 = 0; }
        const float r2f = (float)__builtin_bit_cast(_Float16, (uint16_t)r2h);
        dxy = sqrtf((mx - ccx) * (mx - ccx) + (my - ccy) * (my - ccy));
        G = (dxy - k1 * sqrtf(r2f) * 1.00001f) * 0.99999f - 1.0e-6f;                   // far_build_kernel's g, with the r2 test (A) runs on
        if (!(G == G) || !(r2f < 3.0e38f)) { G = -__builtin_inff(); always = true; }
        if (!(dxy == dxy)) dxy = __builtin_inff();
        const bool front = always;                         // (c0: unused since the levels carry their complement's weakest cone code)
        const uint32_t og = fkey(G);
        hi = front ? 0u : (og < 1u ? 1u : (og > 0xfffffffeu ? 0xfffffffeu : og));
        rec = make_uint4(r.x, r.y, (r.z & 0xffffu) | (r2h << 16), (id & WALK_NOID) | (code6 << 26));
    }
    key[tid] = ((unsigned long long)hi << 32) | tid;
    s_G[tid] = G; s_z[tid] = mz; s_d[tid] = dxy; s_rec[tid] = rec;
    const uint32_t n_valid = (uint32_t)__syncthreads_count(id != 0xffffffffu);
    const uint32_t n_front = (uint32_t)__syncthreads_count(hi == 0u);
    for (uint32_t len = 2; len <= 256u; len <<= 1) {
        for (uint32_t stride = len >> 1; stride > 0; stride >>= 1) {
            if (tid < 128u) {
                const uint32_t lo = ((tid / stride) * stride << 1) + (tid % stride), hi2 = lo + stride;
                const bool up = (lo & len) == 0;
                const unsigned long long a = key[lo], b = key[hi2];
                if ((a > b) == up) { key[lo] = b; key[hi2] = a; }
            }
            __syncthreads();
        }
    }
    const uint32_t src = (uint32_t)(key[tid] & 0xffu);
    if (tid < K) wrec[(uint64_t)cell * K + tid] = s_rec[src];
    if (tid == 0) { s_n[0] = n_valid; s_n[1] = n_front; }
    if (tid < WALK_LEVELS - 1) {
        uint32_t cnt = n_front + c_walk_off[tid];
        cnt = cnt > n_valid ? n_valid : cnt;
        float g = __builtin_inff(), z0 = __builtin_inff(), z1 = -__builtin_inff(), ro = 0.0f;
        uint32_t minc = 63u;
        for (uint32_t j = cnt; j < n_valid; ++j) {         // the complement of the prefix (no always-candidate in it: cnt >= n_front)
            const uint32_t s2 = (uint32_t)(key[j] & 0xffu);
            g = fminf(g, s_G[s2]); z0 = fminf(z0, s_z[s2]); z1 = fmaxf(z1, s_z[s2]); ro = fmaxf(ro, s_d[s2]);
            minc = min(minc, s_rec[s2].w >> 26);
        }
        wlvl[(uint64_t)cell * WALK_LEVELS + tid] = make_uint4(__float_as_uint(g), __float_as_uint(z0), __float_as_uint(z1),
                                                              half_up_bits(ro) | (cnt << 16) | (minc << 25));
    }
    if (tid == WALK_LEVELS - 1) wlvl[(uint64_t)cell * WALK_LEVELS + tid] = make_uint4(__float_as_uint(ccx), __float_as_uint(ccy), n_front, n_valid << 16);
    if (counts && tid == 0) atomicAdd(counts, n_front);
}

// ---------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------
#define WALK_QCAP 1152u              // LDS queue entries per wave (4.5 KB + 256 B of running minima: 8 waves per SIMD fit 160 KB)
#ifndef WALK_UNROLL
#define WALK_UNROLL 2u               // list entries per lane and loop trip: a trip appends at most 64 * WALK_UNROLL entries
#endif
#ifndef WALK_LONG
#define WALK_LONG 16u                // a wave's cap: all but at most this many of its rays walk their whole prefix themselves
#endif
#define WALK_FULL (WALK_QCAP - 64u * WALK_UNROLL)      // more entries than this: the next trip could overflow the queue

struct RawTriW { uint32_t d[5]; };   // rtab record (rover_cull.hip): v0 xyz, v1 xyz, v2 xyz, pad as ten fp16 values

__device__ __forceinline__ void walk_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// queue entry: triangle id (25 bits) | map << 25 | position of the ray in the run << 26
#define WALK_IDMASK 0x1ffffffu

// EXACT phase: one lane per queue entry, the arithmetic of ray_casting.py:31-59 (rover_raymath.h) on the entry's triangle for the
// entry's ray, LDS atomicMin on ordered-u32 keys (only lanes that hit take part).  The entry names the map, so the triangle's
// record and the ray's are requested together: one memory round trip per round of 64 entries.
template <int H>
__device__ __forceinline__ void walk_exact(const RayRec* __restrict__ rays, const RawTriW* __restrict__ rtab0, const RawTriW* __restrict__ rtab1,
                                           const uint32_t* lq, uint32_t n, uint32_t gid, uint32_t lane, uint32_t* bk) {
    if (n == 0u) return;
    uint32_t en_next = lq[min(lane, n - 1u)];
    for (uint32_t base = 0; base < n; base += 64u) {                // wave-uniform
        const bool live = base + lane < n;
        const uint32_t en = en_next;
        const uint32_t pos = en >> 26, id = en & WALK_IDMASK;
        const RawTriW r = ((en >> 25) & 1u ? rtab1 : rtab0)[id];
        const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos << 2), (int)gid);
        const float4* rp = reinterpret_cast<const float4*>(rays + g);
        const float4 ra = rp[0], rb = rp[1];
        if (base + 64u < n) en_next = lq[min(base + 64u + lane, n - 1u)];
        float best;
        if (H) {
            _Float16 v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const half2w x = __builtin_bit_cast(half2w, r.d[q >> 1]);
                v[q] = (q & 1) ? x.y : x.x;
            }
            Tri1H t;
            set_one_h(t, v);
            best = cast_one_h(t, (_Float16)ra.x, (_Float16)ra.y, (_Float16)ra.z, (_Float16)rb.x, (_Float16)rb.y, (_Float16)rb.z);
        } else {
            float v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) v[q] = (q & 1) ? half_hi(r.d[q >> 1]) : half_lo(r.d[q >> 1]);
            Tri1 t;
            set_one(t, v);
            best = cast_one(t, ra.x, ra.y, ra.z, rb.x, rb.y, rb.z);
        }
        const uint32_t k = fkey(best);
        if (live && k < fkey(RAY_MISS)) atomicMin(bk + pos, k);
    }
}

struct WalkArgs {
    const RayRec* rays;
    const uint32_t* sorted;
    uint32_t n_sorted;
    const uint4 *rec0, *rec1;        // wrec of the two maps (for the proof in force)
    const uint4 *lvl0, *lvl1;        // wlvl
    const uint4 *ctab0, *ctab1;      // the proof's per-triangle sphere / normal records (rover_cull.hip): test (B) itself, wave-walked part
    const RawTriW *rtab0, *rtab1;
    uint32_t K0, K1;
    uint32_t n_blocks, split, t8, r8, chsr, run, run_r, j0;
    float c_a, k2_far, tau2;
    float* out;
    uint4* stats;
    uint32_t diag;                   // diagnostic builds (ROVER_WALK_DIAG): parts of the kernel switched off — wrong results, right timing
};

template <int H>
__global__ void __launch_bounds__(64) walk_scan_kernel(WalkArgs a) {
    __shared__ uint32_t s_lq[WALK_QCAP];
    __shared__ uint32_t s_bk[64];
    const uint32_t x = blockIdx.x & 7u, lane = threadIdx.x;
    // XCD-aware order of the runs: the one of cull_scan_kernel (rover_cull.hip): workgroup = one wave, four consecutive ones of an XCD
    // form a block slot; terrain blocks dealt in chunks round robin over the XCDs, then the rocks blocks
    const uint32_t qx = blockIdx.x >> 3, w = qx & 3u, jslot = qx >> 2;
    const uint32_t j = jslot + a.j0;
    uint32_t lb;
    const uint32_t chs = a.chsr & 0xffu, chr = a.chsr >> 8;
    if (j < a.t8) {
        lb = chs == 31u ? x * a.t8 + j : ((((j >> chs) << 3) + x) << chs) + (j & ((1u << chs) - 1u));
        if (lb >= a.split) return;
    } else {
        const uint32_t jr = j - a.t8;
        lb = a.split + (chr == 31u ? x * a.r8 + jr : ((((jr >> chr) << 3) + x) << chr) + (jr & ((1u << chr) - 1u)));
        if (lb >= a.n_blocks) return;
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(lb * 4u + w);
    const uint32_t my_run = lb < a.split ? a.run : a.run_r;
    const uint32_t i0 = lb < a.split ? wave * a.run : a.split * 4u * a.run + (wave - a.split * 4u) * a.run_r;
    if (i0 >= a.n_sorted) return;
    const uint32_t n_run = min(my_run, a.n_sorted - i0);             // <= 64
    const uint32_t gid = a.sorted[i0 + (lane < n_run ? lane : n_run - 1u)];
    s_bk[lane] = fkey(RAY_MISS);
    if (a.diag & 16u) { if (lane < n_run) a.out[gid] = 11.0f; return; }
    // progress of the run across segments (wave-uniform): the lane-walked part is done up to entry j_next; the wave-walked part up
    // to trip t_next of the long ray l_next
    uint32_t j_next = 0, l_next = 0, t_next = 0;
    uint32_t ctot = 0, n_zero = 0, n_part = 0, n_offb = 0, sum_cnt = 0, trips_stat = 0;
    const float k_ca = a.c_a;
    for (;;) {
        // one SEGMENT: everything a lane needs is derived from its ray id here (through an opaque copy, so that nothing but `gid`
        // stays live across the exact phase — the register high-water mark); nearly every run is one segment
        uint32_t gid_s = gid;
        asm volatile("" : "+v"(gid_s));
        const float4 rsa = reinterpret_cast<const float4*>(a.rays + gid_s)[0], rsb = reinterpret_cast<const float4*>(a.rays + gid_s)[1];
        const uint32_t rflags = __float_as_uint(rsb.w), cell = __float_as_uint(rsa.w), map = rflags & 1u;
        const uint32_t ray16 = rflags >> 16, ray6 = (ray16 + 1023u) >> 10;        // (B) holds for an entry if code6 >= ray6 (64: never)
        const uint4* lv = (map ? a.lvl1 : a.lvl0) + (size_t)cell * WALK_LEVELS;
        const uint32_t Kc = map ? a.K1 : a.K0;
        const uint4* base = (map ? a.rec1 : a.rec0) + (size_t)cell * Kc;
        // ONE round of loads: the cell's head record, its first eight levels, and the first two trips of its list (entries j_next ..:
        // the cell has them whatever prefix the ray will need — an entry past the prefix is fetched and not used)
        uint4 b0[WALK_UNROLL], b1[WALK_UNROLL], b2[WALK_UNROLL];
        auto fetch = [&](uint4 (&r)[WALK_UNROLL], uint32_t j_at) {
#pragma unroll
            for (uint32_t u = 0; u < WALK_UNROLL; ++u) r[u] = base[min(j_at + u, Kc - 1u)];
        };
        const uint4 hd = lv[WALK_LEVELS - 1];                                     // {cell centre x, y, -, n << 16}
        uint4 q8[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) q8[u] = lv[u];
        fetch(b0, j_next);
        fetch(b1, j_next + WALK_UNROLL);
        const uint32_t n_all = lane < n_run ? hd.w >> 16 : 0u;
        // the ray's side of the group inequality (far_build_kernel / cull_scan_kernel's far skip: same expressions, same margins)
        const float ox = rsa.x - __uint_as_float(hd.x), oy = rsa.y - __uint_as_float(hd.y), o = __builtin_amdgcn_sqrtf(ox * ox + oy * oy);
        const float dxy2 = rsb.x * rsb.x + rsb.y * rsb.y, adz = fabsf(rsb.z);
        const bool steep = adz * adz >= 0.81f * (dxy2 + adz * adz) * 1.0001f;  // cos(beta) >= 0.9
        const float sq = __builtin_amdgcn_sqrtf(dxy2), dxy1 = sq * 1.0001f;
        uint32_t cnt = n_all;
        bool open = steep && ray6 <= 63u && lane < n_run;                        // still looking for a level whose complement it clears
        auto level = [&](const uint4& q) {
            const float G = __uint_as_float(q.x), z0 = __uint_as_float(q.y), z1 = __uint_as_float(q.z), ro = half_lo(q.w);
            const float dzm = fmaxf(fabsf(rsa.z - z0), fabsf(rsa.z - z1));
            const float e_adz = o * adz + dzm * sq * 1.0001f;
            const float amax = (dzm * adz + (o + ro) * dxy1) * 1.0001f;
            const bool ok = (G * 0.9999f - 1.0e-5f) * adz > (e_adz + a.k2_far * amax * adz) * 1.0001f      // (A) for the whole complement
                            && ((q.w >> 25) & 63u) >= ray6;                                               // (B) for each of its triangles
            if (open && ok) { cnt = (q.w >> 16) & 0x1ffu; open = false; }
        };
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) level(q8[u]);
        if (__builtin_amdgcn_ballot_w64(open)) {                                 // (rare: a second round for levels 8..10)
            uint4 q3[3];
#pragma unroll
            for (uint32_t u = 0; u < 3u; ++u) q3[u] = lv[8u + u];
#pragma unroll
            for (uint32_t u = 0; u < 3u; ++u) level(q3[u]);
        }
        const uint64_t run_m = n_run >= 64u ? ~0ull : ((1ull << n_run) - 1ull);
        // cap: the largest prefix length that more than WALK_LONG of the run's rays reach (a bit-wise search over the nine bits of a count),
        // rounded up to whole trips.  Rays no cone code can serve (ray6 = 64: horizontal rays) are left to the wave entirely.
        const bool nocone = ray6 > 63u;
        uint32_t cap = 0;
#pragma unroll
        for (int b = 8; b >= 0; --b) {
            const uint32_t t = cap | (1u << b);
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(!nocone && cnt >= t)) > WALK_LONG) cap = t;
        }
        cap = (cap + WALK_UNROLL - 1u) / WALK_UNROLL * WALK_UNROLL;
        const uint32_t lcnt = nocone ? 0u : min(cnt, cap);                       // entries this lane walks itself
        const uint64_t longm = __builtin_amdgcn_ballot_w64(lcnt < cnt) & run_m;  // rays with a part left to the wave
        if (j_next == 0u && l_next == 0u && t_next == 0u) {
            n_zero = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cnt == 0u) & run_m);
            n_part = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cnt < n_all) & run_m);
            n_offb = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(nocone) & run_m);
            uint32_t sum = cnt, lt = lcnt < cnt ? (cnt - lcnt + 63u) >> 6 : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { sum += (uint32_t)__shfl_xor((int)sum, off); lt += (uint32_t)__shfl_xor((int)lt, off); }
            sum_cnt = __builtin_amdgcn_readfirstlane(sum);
            trips_stat = cap + __builtin_amdgcn_readfirstlane(lt);               // entry trips of the lane part + 64-entry trips of the wave part
        }
        uint32_t cused = 0;
        uint32_t jj = j_next;
        bool full = false;
        walk_lds_sync();
        if (a.diag & 8u) { if (lane < n_run) a.out[gid] = (float)cnt + b0[0].x + b1[0].x; return; }
        const uint32_t qtag = (map << 25) | (lane << 26);
        // The LANE-walked part, WALK_UNROLL list entries per lane and trip through three buffers: the loads of a trip go out two trips
        // before it is tested (without that every trip was a full round trip to L2 / HBM).
        auto test = [&](const uint4 (&r)[WALK_UNROLL], uint32_t j_at) {
#pragma unroll
            for (uint32_t u = 0; u < WALK_UNROLL; ++u) {
                const bool act = j_at + u < lcnt;
                const float mx = __uint_as_float(r[u].x), my = __uint_as_float(r[u].y), mz = half_lo(r[u].z), r2 = half_hi(r[u].z);
                const float hx = rsa.x - mx, hy = rsa.y - my, hz = rsa.z - mz;
                const float hdot = __builtin_fmaf(hz, rsb.z, __builtin_fmaf(hy, rsb.y, hx * rsb.x));
                float hh = hx * hx; hh = __builtin_fmaf(hy, hy, hh); hh = __builtin_fmaf(hz, hz, hh);
                const float A = __builtin_fmaf(hh, k_ca, -(hdot * hdot));        // (A): c_a |h|^2 - (h.d)^2 > r2
                const bool cull = (A > r2) && ((r[u].w >> 26) >= ray6);          // ... and (B) by the triangle's cone code
                const uint64_t m = __builtin_amdgcn_ballot_w64(act && !cull);
                if (m) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (act && !cull) s_lq[cused + rank] = (r[u].w & WALK_IDMASK) | qtag;
                    cused += (uint32_t)__builtin_popcountll(m);
                }
            }
        };
#pragma unroll 1
        while (jj < cap && !(a.diag & 4u)) {
            fetch(b2, jj + 2u * WALK_UNROLL); test(b0, jj); jj += WALK_UNROLL;
            if (cused > WALK_FULL) { full = true; break; }
            if (jj >= cap) break;
            fetch(b0, jj + 2u * WALK_UNROLL); test(b1, jj); jj += WALK_UNROLL;
            if (cused > WALK_FULL) { full = true; break; }
            if (jj >= cap) break;
            fetch(b1, jj + 2u * WALK_UNROLL); test(b2, jj); jj += WALK_UNROLL;
            if (cused > WALK_FULL) { full = true; break; }
        }
        j_next = jj;
        // The WAVE-walked part: for every long ray, 64 entries per trip, one per lane; the ray's parameters are wave-uniform.  Where the
        // cone code cannot establish (B), (B) itself is evaluated from the triangle's sphere / normal record, as cull_scan_kernel does.
        // Items (ray L, trip t) in order; the record of the NEXT item is requested before the current one is tested.
        if (!full && longm && !(a.diag & 2u)) {
            uint64_t lm = longm & (~0ull << l_next);
            auto bu = [&](uint32_t v, uint32_t L) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)L); };
            auto bf = [&](float v, uint32_t L) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)L)); };
            auto item_addr = [&](uint32_t L, uint32_t t) {        // -> this lane's record of item (L, t) (clamped into the ray's range)
                const uint32_t lo = bu(lcnt, L), hiN = bu(cnt, L), mapL = bu(map, L), cellL = bu(cell, L);
                const uint32_t jx = lo + t * 64u + lane;
                return (mapL ? a.rec1 : a.rec0) + (size_t)cellL * (mapL ? a.K1 : a.K0) + (jx < hiN ? jx : lo);
            };
            uint32_t L = lm ? (uint32_t)__builtin_ctzll(lm) : 64u, t = (L == l_next) ? t_next : 0u;
            if (L < 64u && bu(lcnt, L) + t * 64u >= bu(cnt, L)) { lm &= lm - 1ull; L = lm ? (uint32_t)__builtin_ctzll(lm) : 64u; t = 0u; }   // (resumed exactly at a ray's end)
            uint4 r = L < 64u ? *item_addr(L, t) : make_uint4(0u, 0u, 0u, 0u);
            while (L < 64u) {
                // the next item
                uint32_t L2 = L, t2 = t + 1u;
                uint64_t lm2 = lm;
                if (bu(lcnt, L) + t2 * 64u >= bu(cnt, L)) { lm2 &= lm2 - 1ull; L2 = lm2 ? (uint32_t)__builtin_ctzll(lm2) : 64u; t2 = 0u; }
                uint4 rn = make_uint4(0u, 0u, 0u, 0u);
                if (L2 < 64u) rn = *item_addr(L2, t2);
                // test the current item
                const float sx = bf(rsa.x, L), sy = bf(rsa.y, L), sz = bf(rsa.z, L), dx = bf(rsb.x, L), dy = bf(rsb.y, L), dz = bf(rsb.z, L);
                const uint32_t lo = bu(lcnt, L), hiN = bu(cnt, L), r6 = bu(ray6, L), mapL = bu(map, L);
                const bool act = lo + t * 64u + lane < hiN;
                const float mx = __uint_as_float(r.x), my = __uint_as_float(r.y), mz = half_lo(r.z), r2 = half_hi(r.z);
                const float hx = sx - mx, hy = sy - my, hz = sz - mz;
                const float hdot = __builtin_fmaf(hz, dz, __builtin_fmaf(hy, dy, hx * dx));
                float hh = hx * hx; hh = __builtin_fmaf(hy, hy, hh); hh = __builtin_fmaf(hz, hz, hh);
                const float A = __builtin_fmaf(hh, k_ca, -(hdot * hdot));
                const bool a_ok = A > r2;
                bool b_ok = (r.w >> 26) >= r6;
                if (__builtin_amdgcn_ballot_w64(act && a_ok && !b_ok)) {             // (B) itself: (n . d)^2 > tau^2 |n|^2 (x F: fp16 proof)
                    if (act && a_ok && !b_ok) {
                        const uint4 c = (mapL ? a.ctab1 : a.ctab0)[r.w & WALK_NOID];
                        const float nx = half_hi(c.z), ny = half_lo(c.w), nz = half_hi(c.w);
                        float r2b = walk_r2(nx, ny, nz, a.tau2);
                        if (H) r2b = r2b * __builtin_fmaf((float)((c.x & 63u) | ((c.y & 63u) << 6)), 1.0f / 512.0f, 1.0f);
                        const float dn = __builtin_fmaf(nz, dz, __builtin_fmaf(ny, dy, nx * dx));
                        b_ok = dn * dn > r2b;
                    }
                }
                const uint64_t m = __builtin_amdgcn_ballot_w64(act && !(a_ok && b_ok));
                if (m) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (act && !(a_ok && b_ok)) s_lq[cused + rank] = (r.w & WALK_IDMASK) | (mapL << 25) | (L << 26);
                    cused += (uint32_t)__builtin_popcountll(m);
                }
                L = L2; t = t2; lm = lm2; r = rn;
                if (cused > WALK_FULL && L < 64u) { full = true; l_next = L; t_next = t; break; }
            }
        }
        walk_lds_sync();
        if (!(a.diag & 1u)) walk_exact<H>(a.rays, a.rtab0, a.rtab1, s_lq, cused, gid, lane, s_bk);
        ctot += cused;
        if (!full) break;                                                       // (full: more to walk, in another segment)
    }
    walk_lds_sync();
    if (lane < n_run) a.out[gid] = funkey(s_bk[lane]);
    // per-wave counters (rover_get_cull_info): {queue entries, rays | rays that skipped part of their list << 8,
    //  rays without a cone code (wave-walked from the start) | rays that walked nothing << 8, trips of the wave | entries walked << 9}
    if (lane == 0u) a.stats[wave] = make_uint4(ctot, n_run | (n_part << 8), n_offb | (n_zero << 8), min(trips_stat, 511u) | (sum_cnt << 9));
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
hipError_t launch_walk_build(const int32_t* map_idx, uint64_t n_cells, uint32_t K, uint32_t T, const uint32_t* newid, const uint4* ctab,
                             const float* nz_abs, float tau2, float k1, uint32_t c0, uint32_t Y, float cell_size, float shift_x, float shift_y,
                             uint4* wrec, uint4* wlvl, uint32_t* counts, hipStream_t s) {
    hipLaunchKernelGGL(walk_build_kernel, dim3((uint32_t)n_cells), dim3(256), 0, s, map_idx, K, T, newid, ctab, nz_abs, tau2, k1, c0, Y, cell_size,
                       shift_x, shift_y, wrec, wlvl, counts);
    return hipGetLastError();
}

hipError_t launch_raycast_walk(const WalkLaunch& w, hipStream_t s) {
    WalkArgs a{};
    a.rays = w.rays; a.sorted = w.sorted; a.n_sorted = w.n_sorted;
    a.rec0 = w.rec0; a.rec1 = w.rec1; a.lvl0 = w.lvl0; a.lvl1 = w.lvl1;
    a.rtab0 = reinterpret_cast<const RawTriW*>(w.rtab0); a.rtab1 = reinterpret_cast<const RawTriW*>(w.rtab1);
    a.K0 = w.K0; a.K1 = w.K1;
    a.n_blocks = w.n_blocks; a.split = w.split; a.t8 = w.t8; a.r8 = w.r8; a.chsr = w.chs | (w.chr << 8); a.run = w.run; a.run_r = w.run_r; a.j0 = 0;
    a.c_a = w.c_a; a.k2_far = w.k2_far; a.tau2 = w.tau2; a.ctab0 = w.ctab0; a.ctab1 = w.ctab1;
    a.out = w.out; a.stats = w.stats; a.diag = w.diag;
    const uint32_t slots = w.t8 + w.r8;
    if (slots == 0u) return hipSuccess;
    if (w.half) hipLaunchKernelGGL(walk_scan_kernel<1>, dim3(slots * 8u * 4u), dim3(64), 0, s, a);
    else        hipLaunchKernelGGL(walk_scan_kernel<0>, dim3(slots * 8u * 4u), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace rover
