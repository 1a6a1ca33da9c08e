// rover_mlp.hip — "next" row f-4: the policy-side consumer of the observation layout.
//
// Reference: omniisaacgymenvs/learning/model.py — Layer = nn.Linear + activation (:105-121), Encoder (:122-150),
// StochasticActorHeightmap.compute (:185-195) / DeterministicHeightmap.compute (:231-241): two encoders over the
// sparse / dense slices of obs, concatenated with the 4 proprioceptive values, then an MLP.
//
// One kernel: y[:, 0:N] = act(x[:, 0:K] @ W^T + b) with W in torch's nn.Linear layout [N][K], x and y addressed by
// (pointer, row stride) so a layer reads an obs slice and writes straight into the concat buffer (no torch.cat).
// f32-input MFMA (v_mfma_f32_32x32x2_f32): exact f32 products, one rounding per accumulate — fp32 like the reference's
// nn.Linear, at the matrix pipe's f32 rate.  A workgroup = 4 waves x 32 rows; A (128 x 32) and W (N x 32) k-slabs are
// staged through LDS with a 33-word pitch (conflict-free column reads), the next slab prefetched into registers while the
// current one is multiplied; each wave keeps N/32 accumulator tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rover_internal.h"

namespace rover {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_BK 32
#define MLP_PITCH 33
#define MLP_MAX_NT 8          // N <= 256

__device__ __forceinline__ float mlp_act(float v, int act) {
    switch (act) {
        case 1: return v > 0.0f ? v : 0.01f * v;                    // nn.LeakyReLU() default slope (model.py:112)
        case 2: return tanhf(v);                                    // nn.Tanh (model.py:115,182)
        case 3: return v > 0.0f ? v : 0.0f;                         // nn.ReLU
        case 4: return v > 0.0f ? v : expm1f(v);                    // nn.ELU (alpha 1)
        default: return v;
    }
}

// NW waves per workgroup, 32 rows each: 4 for large batches, 1 when M / 128 workgroups would leave most of the 256 CUs idle
template <int NT, int NW>
__global__ void __launch_bounds__(64 * NW) linear_act_kernel(LinearArgs a) {
    constexpr uint32_t BM = 32u * NW, RSTEP = 2u * NW;               // rows per workgroup; rows staged per pass
    __shared__ float As[BM * MLP_PITCH];
    __shared__ float Ws[NT * 32 * MLP_PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t row0 = blockIdx.x * BM, n0 = blockIdx.y * (NT * 32u);     // blockIdx.y: column tile (small-batch launch)
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const uint32_t ar = lane & 31u, ak = lane >> 5;                  // A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
    // k-slab staging: thread (r = tid / 32 + RSTEP j, c = tid % 32) — 128-byte coalesced rows, zero-padded past M / N / K.
    // The slab for step s + 1 is fetched into registers while step s is multiplied out of LDS (one LDS buffer, two syncs).
    const uint32_t sc = tid & 31u, sr = tid >> 5;
    float pa[BM / RSTEP], pw[NT * 32 / RSTEP];
    auto fetch = [&](uint32_t k0) {
        const uint32_t gk = k0 + sc;
        const bool kin = gk < (uint32_t)a.K;
#pragma unroll
        for (int j = 0; j < (int)(BM / RSTEP); ++j) {
            const uint32_t gr = row0 + sr + RSTEP * j;
            pa[j] = (kin && gr < (uint32_t)a.M) ? a.x[(size_t)gr * a.x_stride + gk] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < (int)(NT * 32 / RSTEP); ++j) {
            const uint32_t n = n0 + sr + RSTEP * j;
            pw[j] = (kin && n < (uint32_t)a.N) ? a.w[(size_t)n * a.K + gk] : 0.0f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < (int)(BM / RSTEP); ++j) As[(sr + RSTEP * j) * MLP_PITCH + sc] = pa[j];
#pragma unroll
        for (int j = 0; j < (int)(NT * 32 / RSTEP); ++j) Ws[(sr + RSTEP * j) * MLP_PITCH + sc] = pw[j];
    };
    fetch(0);
    for (uint32_t k0 = 0; k0 < (uint32_t)a.K; k0 += MLP_BK) {
        stash();
        __syncthreads();
        if (k0 + MLP_BK < (uint32_t)a.K) fetch(k0 + MLP_BK);           // in flight during the MFMAs below
#pragma unroll 4
        for (uint32_t kk = 0; kk < MLP_BK; kk += 2) {
            const float av = As[(wave * 32u + ar) * MLP_PITCH + kk + ak];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float bv = Ws[(t * 32u + ar) * MLP_PITCH + kk + ak];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint32_t col = n0 + t * 32u + (lane & 31u);
        if (col >= (uint32_t)a.N) continue;
        const float bias = a.b ? a.b[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = row0 + wave * 32u + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * (lane >> 5);
            if (row < (uint32_t)a.M) a.y[(size_t)row * a.y_stride + col] = mlp_act(acc[t][r] + bias, a.act);
        }
    }
}

template <int NW>
static hipError_t launch_linear_nw(const LinearArgs& a, hipStream_t s) {
    const uint32_t nb = (uint32_t)((a.M + 32 * NW - 1) / (32 * NW));
    const int nt_all = (a.N + 31) / 32;
    // 6-8 accumulator tiles per wave (248-312 VGPRs) leave one wave per SIMD and nothing to overlap the staging with: those
    // layers are split into column halves over grid.y (the A tile is read twice, from L2): 124 -> 256 at 65 536 rows 134 -> 88 us
    // (5 tiles as 3 + 2 was slower: 78 -> 113 us)
    const int ny = nt_all > 5 ? 2 : 1, nt = (nt_all + ny - 1) / ny;
    const dim3 grid(nb, (uint32_t)ny);
    switch (nt) {
        case 1: hipLaunchKernelGGL((linear_act_kernel<1, NW>), grid, dim3(64 * NW), 0, s, a); break;
        case 2: hipLaunchKernelGGL((linear_act_kernel<2, NW>), grid, dim3(64 * NW), 0, s, a); break;
        case 3: hipLaunchKernelGGL((linear_act_kernel<3, NW>), grid, dim3(64 * NW), 0, s, a); break;
        case 4: hipLaunchKernelGGL((linear_act_kernel<4, NW>), grid, dim3(64 * NW), 0, s, a); break;
        case 5: hipLaunchKernelGGL((linear_act_kernel<5, NW>), grid, dim3(64 * NW), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_linear_act(const LinearArgs& a, hipStream_t s) {
    // 128-row workgroups holding all N columns need M >= 128 * 512 rows to put two of them on each of the 256 CUs; below that
    // one wave per workgroup and one 32-column tile per workgroup (grid.y), e.g. 4 096 x 80 -> 128 x 3 workgroups
    if (a.M >= 128 * 512) return launch_linear_nw<4>(a, s);
    if (a.N > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL((linear_act_kernel<1, 1>), dim3((uint32_t)((a.M + 31) / 32), (uint32_t)((a.N + 31) / 32)), dim3(64), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// A CHAIN of two or four layers in one kernel (an encoder 634 -> 80 -> 60, or the MLP 124 -> 256 -> 160 -> 128 -> 2 of
// learning/model.py:122-150,176-195): only the chain's input and its last output touch HBM.
//
// The MFMA runs transposed, D[n][m] = W[n][k] X^T[k][m]: the accumulator tile then has the batch row m on the LANE and the output
// feature n in the registers — exactly the B-operand shape of the next layer's MFMA.  Activations never leave the registers, no
// LDS transpose, and only weight slabs are staged through LDS.  Layer 1 takes its B operand from an LDS slab of the input rows.
//
// 16 x 16 x 4 MFMAs (v_mfma_f32_16x16x4_f32), 16 batch rows per wave, 8 waves per workgroup.  (Round 2's first version used
// 32 x 32 x 2 tiles and 32 rows per wave: 0.54 ms for the actor forward at 65 536 rows, this one 0.34 ms.)
//   * a wave's state is 4 VGPRs per 16 output features instead of 16 per 32: the whole MLP chain fits in ~128 VGPRs (the 32-wide
//     version needed 384: one wave per SIMD, every barrier and every exposed load stalled the matrix pipe), the encoder chain in ~70;
//     with 65 536 rows there are 4 waves per SIMD to overlap one workgroup's staging with another's MFMAs;
//   * 80 output features are 5 tiles, not 3 x 32 = 96: no padded MFMA work in the two layers that hold 60 % of the flops;
//   * operands come from LDS as 16-byte reads: lane (m = lane & 15, g = lane >> 4) reads k = 4 g .. 4 g + 3 of its row, which
//     feeds FOUR MFMAs (MFMA j takes element j: its four k-indices are j, 4 + j, 8 + j, 12 + j on both operands — a sum over k does
//     not care).  Row pitches of 36 / 20 floats put the 8 lanes of a 16-byte read phase on 8 different bank groups.
// Transposed as above: D[n][m] = W[n][k] X^T[k][m].  D: lane holds features n = 16 t + 4 g + r (r < 4) of batch row m — as the next
// layer's B operand (k-index g <-> lane group g) register r pairs with weight column 16 t + 4 g + r, i.e. element r of the 16-byte
// read at column 4 g of the [n_out][16] weight slab of input tile t.
// ---------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));       // a float4 at any float address (global_load_dwordx4)
#define C16_XP 36
#define C16_WP 20

template <int T, int ACT>
__device__ __forceinline__ void c16_bias_act_as(f32x4 (&acc)[T], const float* __restrict__ bias, int n_valid, uint32_t g) {
    n_valid = n_valid < 0 ? 0 : n_valid;          // a half past the layer's width (n[0] < 128 in the 4-layer chain): all lanes invalid
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t n = 16u * t + 4u * g + (uint32_t)r;
            const float b = (bias && n < (uint32_t)n_valid) ? bias[n] : 0.0f;
            acc[t][r] = n < (uint32_t)n_valid ? mlp_act(acc[t][r] + b, ACT) : 0.0f;
        }
}
template <int T>
__device__ __forceinline__ void c16_bias_act(f32x4 (&acc)[T], const float* __restrict__ bias, int n_valid, int act, uint32_t g) {
    switch (act) {                                                   // (wave-uniform; outside the unrolled body, see chain_bias_act)
        case 1: c16_bias_act_as<T, 1>(acc, bias, n_valid, g); break;
        case 2: c16_bias_act_as<T, 2>(acc, bias, n_valid, g); break;
        case 3: c16_bias_act_as<T, 3>(acc, bias, n_valid, g); break;
        case 4: c16_bias_act_as<T, 4>(acc, bias, n_valid, g); break;
        default: c16_bias_act_as<T, 0>(acc, bias, n_valid, g); break;
    }
}
// hidden layers of the 4-layer chain (up to 64 values per lane): none / LeakyReLU / ReLU only, as one select per value — five
// unrolled copies of 64 activations, tanhf and expm1f among them, made the kernel 280 KB (launch_chain sends other nets layer by layer)
template <int T>
__device__ __forceinline__ void c16_bias_act_hidden(f32x4 (&acc)[T], const float* __restrict__ bias, int n_valid, int act, uint32_t g) {
    const float slope = act == 1 ? 0.01f : (act == 3 ? 0.0f : 1.0f);
    n_valid = n_valid < 0 ? 0 : n_valid;          // second half of a first layer narrower than 128: no bias read, every value 0
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t n = 16u * t + 4u * g + (uint32_t)r;
            const float b = (bias && n < (uint32_t)n_valid) ? bias[n] : 0.0f;
            const float v = acc[t][r] + b;
            const float neg = act == 3 ? 0.0f : slope * v;
            acc[t][r] = n < (uint32_t)n_valid ? (v > 0.0f ? v : neg) : 0.0f;
        }
}

// one register-fed layer: out[TO] = W[:, features of in[TI]] . in.  The [n_out][16] weight slab of input tile ti + 1 is fetched into
// registers before the MFMAs of tile ti and stored into the other LDS buffer after them: one barrier per slab.
template <int TI, int TO>
__device__ __forceinline__ void c16_layer(const f32x4 (&in)[TI], f32x4 (&out)[TO], const float* __restrict__ w, int n_in, int n_out,
                                          uint32_t col0 /* feature index of in[0]'s first row */, float* __restrict__ Wb /* 2 x [TO * 16][C16_WP] */,
                                          uint32_t tid, uint32_t m, uint32_t g) {
    constexpr uint32_t BUF = TO * 16u * C16_WP;
    constexpr int NJ = (TO * 16 + 127) / 128;
    const uint32_t c4 = tid & 3u, sr = tid >> 2;                     // 16-byte column of the slab; first slab row of this thread (128 rows per pass)
    f32x4 pw[NJ];
    auto fetch = [&](int ti) {
        const uint32_t col = col0 + 16u * ti + 4u * c4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t n = sr + 128u * j;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (n < (uint32_t)n_out && n < TO * 16u) {
                const float* __restrict__ p = w + (size_t)n * n_in;
                if (col + 4u <= (uint32_t)n_in) v = *reinterpret_cast<const f32x4u*>(p + col);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < (uint32_t)n_in) v[e] = p[col + e];
                }
            }
            pw[j] = v;
        }
    };
    auto stash = [&](uint32_t buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t n = sr + 128u * j;
            if (n < TO * 16u) *reinterpret_cast<f32x4*>(Wb + buf * BUF + n * C16_WP + 4u * c4) = pw[j];
        }
    };
    __syncthreads();                                                 // the previous user of the LDS is done
    fetch(0);
    stash(0);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
        const uint32_t buf = (uint32_t)ti & 1u;
        if (ti + 1 < TI) fetch(ti + 1);                              // in flight during the MFMAs below
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(Wb + buf * BUF + (16u * to + m) * C16_WP + 4u * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) out[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[r], in[ti][r], out[to], 0, 0, 0);
        }
        if (ti + 1 < TI) {
            stash(buf ^ 1u);                                         // nobody reads the other buffer now
            __syncthreads();
        }
    }
}

template <int T>
__device__ __forceinline__ void c16_zero(f32x4 (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// the chain's last activations -> y[row][n]: lane (m, g) holds n = 16 t + 4 g .. + 3 of row m (16 bytes in a row)
template <int T>
__device__ __forceinline__ void c16_store(const f32x4 (&acc)[T], float* __restrict__ y, int64_t y_stride, int n_valid, int M, uint32_t row,
                                          uint32_t g) {
    if (row >= (uint32_t)M) return;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t n = 16u * t + 4u * g + (uint32_t)r;
            if (n < (uint32_t)n_valid) y[(size_t)row * y_stride + n] = acc[t][r];
        }
}

// layer 1 for the TN output tiles that start at weight row n_off: both operands from LDS slabs of 32 k.  Staging is 16 bytes per
// lane and instruction (global_load_dwordx4 at 4-byte alignment -> ds_write_b128): 2 + 2 loads and stores per thread and slab where
// dword staging took 13 + 13 with a 64-bit address and two bounds tests each — measured, the staging instructions do not hide
// under the MFMAs of the other waves of the SIMD, they add to them.  The 16 bytes that hold the end of the k range are fetched element by element.
// PF: the next slab is fetched into registers under the MFMAs (the encoders' long k ranges); without it the 16 staging registers are
// not live across the MFMAs (the MLP's first layer: k = 124 is four slabs, and the kernel has no registers to spare).
template <int TN, bool PF>
__device__ __forceinline__ void c16_layer1(f32x4 (&a1)[TN], const ChainArgs& a, uint32_t n_off, uint32_t row0, float* __restrict__ Xs,
                                           float* __restrict__ Ws, uint32_t tid, uint32_t wave, uint32_t m, uint32_t g) {
    constexpr int WJ = (TN * 16 + 63) / 64;                         // weight-row passes of 64 rows
    const uint32_t c4 = tid & 7u, r0 = tid >> 3;                     // 16-byte column of the slab; first slab row of this thread
    f32x4 px[2], pw[WJ];
    auto load4 = [&](const float* __restrict__ p, uint32_t k, bool row_ok) {     // elements k .. k + 3 of a row, zero past K0 / the rows
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row_ok) {
            if (k + 4u <= (uint32_t)a.K0) v = *reinterpret_cast<const f32x4u*>(p + k);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < (uint32_t)a.K0) v[e] = p[k + e];
            }
        }
        return v;
    };
    auto fetch = [&](uint32_t k0) {
        const uint32_t k = k0 + 4u * c4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t gr = row0 + r0 + 64u * j;
            px[j] = load4(a.x + (size_t)gr * a.x_stride, k, gr < (uint32_t)a.M);
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const uint32_t nl = r0 + 64u * j, n = n_off + nl;
            pw[j] = load4(a.w[0] + (size_t)n * a.K0, k, nl < TN * 16u && n < (uint32_t)a.n[0]);
        }
    };
    if (PF) fetch(0);
    for (uint32_t k0 = 0; k0 < (uint32_t)a.K0; k0 += 32u) {
        if (!PF) fetch(k0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(Xs + (r0 + 64u * j) * C16_XP + 4u * c4) = px[j];
#pragma unroll
        for (int j = 0; j < WJ; ++j)
            if (r0 + 64u * j < TN * 16u) *reinterpret_cast<f32x4*>(Ws + (r0 + 64u * j) * C16_XP + 4u * c4) = pw[j];
        __syncthreads();
        if (PF && k0 + 32u < (uint32_t)a.K0) fetch(k0 + 32u);          // in flight during the MFMAs below
#pragma unroll
        for (uint32_t kq = 0; kq < 2; ++kq) {
            const f32x4 xb = *reinterpret_cast<const f32x4*>(Xs + (wave * 16u + m) * C16_XP + 16u * kq + 4u * g);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(Ws + (16u * t + m) * C16_XP + 16u * kq + 4u * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) a1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb[j], a1[t], 0, 0, 0);
            }
        }
    }
}

// T1..T4: 16-wide output tiles of the layers (T3 = T4 = 0: a 2-layer chain).  The 4-layer chain computes layer 1 in two halves of
// T1 / 2 tiles, each fed into layer 2's accumulators as soon as it is done (the input rows are staged twice: k = 124 is four
// slabs): 32 + 40 live accumulator registers instead of 64 + 40, and the kernel stays under 128 VGPRs without spilling.
template <int T1, int T2, int T3, int T4>
__global__ void __launch_bounds__(512, 4) chain16_kernel(ChainArgs a) {       // four waves per SIMD (two workgroups per CU): <= 128 VGPRs
    constexpr bool LONG = T3 != 0;
    constexpr int T1H = LONG ? T1 / 2 : T1;
    constexpr uint32_t L1 = (128u + T1H * 16u) * C16_XP;                                 // layer 1: input rows + weight rows, 32 k each
    constexpr uint32_t TOMAX = T2 > T3 ? (T2 > T4 ? T2 : T4) : (T3 > T4 ? T3 : T4);
    constexpr uint32_t LN = 2u * TOMAX * 16u * C16_WP;                                   // later layers: two weight slabs
    __shared__ __attribute__((aligned(16))) float lds[L1 > LN ? L1 : LN];
    float* Xs = lds;
    float* Ws = lds + 128u * C16_XP;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t m = lane & 15u, g = lane >> 4;
    const uint32_t row0 = blockIdx.x * 128u;
    const uint32_t row = row0 + wave * 16u + m;
    if constexpr (!LONG) {
        f32x4 a1[T1];
        c16_zero(a1);
        c16_layer1<T1, true>(a1, a, 0u, row0, Xs, Ws, tid, wave, m, g);
        c16_bias_act(a1, a.b[0], a.n[0], a.act[0], g);
        if constexpr (T2 == 0) {
            c16_store(a1, a.y, a.y_stride, a.n[0], a.M, row, g);
        } else {
            f32x4 a2[T2];
            c16_zero(a2);
            c16_layer<T1, T2>(a1, a2, a.w[1], a.n[0], a.n[1], 0u, lds, tid, m, g);
            c16_bias_act(a2, a.b[1], a.n[1], a.act[1], g);
            c16_store(a2, a.y, a.y_stride, a.n[1], a.M, row, g);
        }
    } else {
        f32x4 a2[T2];
        c16_zero(a2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a1[T1H];
            c16_zero(a1);
            if (h) __syncthreads();                                  // layer 2's weight slabs share the LDS with layer 1's
            c16_layer1<T1H, false>(a1, a, (uint32_t)h * T1H * 16u, row0, Xs, Ws, tid, wave, m, g);
            {   // bias + activation of this half: feature index = h T1H 16 + 16 t + 4 g + r
                const float* bh = a.b[0] ? a.b[0] + h * T1H * 16 : nullptr;
                c16_bias_act_hidden(a1, bh, a.n[0] - h * T1H * 16, a.act[0], g);
            }
            c16_layer<T1H, T2>(a1, a2, a.w[1], a.n[0], a.n[1], (uint32_t)h * T1H * 16u, lds, tid, m, g);
        }
        c16_bias_act_hidden(a2, a.b[1], a.n[1], a.act[1], g);
        f32x4 a3[T3];
        c16_zero(a3);
        c16_layer<T2, T3>(a2, a3, a.w[2], a.n[1], a.n[2], 0u, lds, tid, m, g);
        if constexpr (T4 == 0) {
            c16_bias_act(a3, a.b[2], a.n[2], a.act[2], g);
            c16_store(a3, a.y, a.y_stride, a.n[2], a.M, row, g);
        } else {
            c16_bias_act_hidden(a3, a.b[2], a.n[2], a.act[2], g);
            f32x4 a4[T4];
            c16_zero(a4);
            c16_layer<T3, T4>(a3, a4, a.w[3], a.n[2], a.n[3], 0u, lds, tid, m, g);
            c16_bias_act(a4, a.b[3], a.n[3], a.act[3], g);
            c16_store(a4, a.y, a.y_stride, a.n[3], a.M, row, g);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Small batches (round 3): a 2-layer chain (an Encoder, learning/model.py:122-150) with its first layer SPLIT ALONG K.
// Below ~16 k rows the chain kernel above is latency-bound, not throughput-bound: M / 128 workgroups each walk the whole k range of
// layer 1 (35 slabs of 32 for the 1 112-wide dense slice, ~1.3 us each) one after the other — 0.16-0.18 ms for the actor forward from
// 512 to 8 192 rows alike.  Here one wave takes 16 rows and ONE chunk of the k range (grid = row tiles x S chunks, every SIMD of the
// chip busy from a few thousand rows on) and streams both MFMA operands straight from memory in the transposed product's own layout
// (lane (m, g) needs 16 contiguous bytes of x row m and of weight row 16 t + m per 16 k: no LDS, no barrier); the partial sums go to a
// scratch [S][M][TN * 16], and a second small kernel adds them in a fixed order (deterministic), applies bias + activation and runs
// the narrow second layer with plain FMAs.
// ---------------------------------------------------------------------------------------------------
// Up to TWO chains over the same M rows side by side in one launch (blockIdx.z picks the chain: the actor's two encoders do not depend
// on each other, learning/model.py:188-190): their first layers together put two waves on every SIMD instead of one after the other.
struct SplitkL1 { const float* x; int64_t x_stride; int K; const float* w; int N; int chunk /* multiple of 16 */; int S; float* part; };
struct SplitkL1Pair { SplitkL1 c[2]; int M; };
template <int TN, int RT>                  // RT row tiles of 16 rows per wave: a weight operand, once loaded, multiplies RT x operands
__global__ void __launch_bounds__(64) splitk_layer1_kernel(SplitkL1Pair args) {
    const SplitkL1& a = args.c[blockIdx.z];
    const float* __restrict__ x = a.x; const float* __restrict__ w = a.w; float* __restrict__ part = a.part;
    const int64_t x_stride = a.x_stride;
    const int M = args.M, K = a.K, N = a.N, chunk = a.chunk;
    const uint32_t lane = threadIdx.x, m = lane & 15u, g = lane >> 4;
    const uint32_t sidx = blockIdx.y;
    if (sidx >= (uint32_t)a.S) return;                              // (the grid covers the chain with more chunks)
    const uint32_t k_lo = sidx * (uint32_t)chunk, k_hi = min((uint32_t)K, k_lo + (uint32_t)chunk);
    f32x4 acc[RT][TN];
#pragma unroll
    for (int q = 0; q < RT; ++q) c16_zero(acc[q]);
    auto load4 = [&](const float* __restrict__ p, uint32_t k, bool ok) {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) {
            if (k + 4u <= k_hi) v = *reinterpret_cast<const f32x4u*>(p + k);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < k_hi) v[e] = p[k + e];
            }
        }
        return v;
    };
    uint32_t row[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) row[q] = (blockIdx.x * RT + q) * 16u + m;
    // four 16-k steps per trip: all their loads (4 x (RT + TN) x 16 bytes per lane) are issued before the first MFMA — a wave has
    // only ~5 steps to do and nothing else to hide a round trip per step behind (1-2 waves per SIMD: registers are free)
    for (uint32_t k0 = k_lo; k0 < k_hi; k0 += 64u) {
        f32x4 xb[4][RT], wa[4][TN];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t k = k0 + 16u * u + 4u * g;
#pragma unroll
            for (int q = 0; q < RT; ++q) xb[u][q] = load4(x + (size_t)row[q] * x_stride, k, row[q] < (uint32_t)M && k < k_hi);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const uint32_t n = 16u * t + m;
                wa[u][t] = load4(w + (size_t)n * K, k, n < (uint32_t)N && k < k_hi);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int q = 0; q < RT; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u][t][j], xb[u][q][j], acc[q][t], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        if (row[q] >= (uint32_t)M) continue;
        float* __restrict__ o = part + ((size_t)sidx * M + row[q]) * (TN * 16u);
#pragma unroll
        for (int t = 0; t < TN; ++t) *reinterpret_cast<f32x4*>(o + 16u * t + 4u * g) = acc[q][t];
    }
}

// 16 rows per workgroup: h = act1(sum_s part[s] + b1) and W2 (transposed) in LDS, then y = act2(W2 h + b2): thread (o = tid % 64,
// q = tid / 64) computes output o of rows q, q + 4, q + 8, q + 12 — W2 reads of a wave are consecutive words, h reads broadcasts
struct SplitkFin { const float* part; int S; const float* b1; int n1, act1; const float* w2; const float* b2; int n2, act2; float* y; int64_t y_stride; };
// copy: columns [0, copy_cols) of copy_src's rows to copy_dst (the actor's proprioception columns on their way into the concat buffer,
// model.py:191 — one launch less), done by the workgroups of chain 0
struct SplitkFinPair { SplitkFin c[2]; int M; const float* copy_src; int64_t copy_src_stride; float* copy_dst; int64_t copy_dst_stride; int copy_cols; };
template <int TN>
__global__ void __launch_bounds__(256) splitk_finish_kernel(SplitkFinPair args) {
    const SplitkFin& a = args.c[blockIdx.z];
    const float* __restrict__ part = a.part; const float* __restrict__ b1 = a.b1; const float* __restrict__ w2 = a.w2;
    const float* __restrict__ b2 = a.b2; float* __restrict__ y = a.y;
    const int S = a.S, M = args.M, n1 = a.n1, act1 = a.act1, n2 = a.n2, act2 = a.act2;
    const int64_t y_stride = a.y_stride;
    if (blockIdx.z == 0 && args.copy_cols > 0) {
        for (uint32_t i = threadIdx.x; i < 16u * (uint32_t)args.copy_cols; i += 256u) {
            const uint32_t r = i / (uint32_t)args.copy_cols, col = i % (uint32_t)args.copy_cols, row = blockIdx.x * 16u + r;
            if (row < (uint32_t)M) args.copy_dst[(size_t)row * args.copy_dst_stride + col] = args.copy_src[(size_t)row * args.copy_src_stride + col];
        }
    }
    constexpr uint32_t NP = TN * 16u;
    __shared__ float h[16][NP + 1];
    __shared__ float w2s[NP][65];                                    // [k][o], n2 <= 64
    const uint32_t row0 = blockIdx.x * 16u, tid = threadIdx.x;
    for (uint32_t i = tid; i < (uint32_t)n2 * (uint32_t)n1; i += 256u) w2s[i % (uint32_t)n1][i / (uint32_t)n1] = w2[i];
    for (uint32_t i = tid; i < 16u * NP; i += 256u) {
        const uint32_t r = i / NP, n = i % NP, row = row0 + r;
        float v = 0.0f;
        if (row < (uint32_t)M && n < (uint32_t)n1) {
            float p[16];                                             // S <= 16: all partial loads in flight, then a fixed summation order
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) p[sidx] = sidx < S ? part[((size_t)sidx * M + row) * NP + n] : 0.0f;
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) v += p[sidx];
            v = mlp_act(v + (b1 ? b1[n] : 0.0f), act1);
        }
        h[r][n] = v;
    }
    __syncthreads();
    const uint32_t o = tid & 63u, q = tid >> 6;
    if (o >= (uint32_t)n2) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < n1; ++k) {
        const float wv = w2s[k][o];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(wv, h[q + 4u * j][k], acc[j]);
    }
    const float bo = b2 ? b2[o] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t row = row0 + q + 4u * j;
        if (row < (uint32_t)M) y[(size_t)row * y_stride + o] = mlp_act(acc[j] + bo, act2);
    }
}

// chunks of the k range so that (row tiles / RT) x chunks ~ one wave per SIMD of the chip (1 024), at least 64 k per chunk
static int splitk_chunks(int M, int K, int* chunk, int* rt) {
    *rt = M >= 2048 ? 2 : 1;
    const int tiles = (M + 16 * *rt - 1) / (16 * *rt);
    int S = (1024 + tiles - 1) / tiles;
    const int s_max = (K + 63) / 64;
    if (S > s_max) S = s_max;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    const int c = ((K + S - 1) / S + 15) / 16 * 16;
    *chunk = c;
    return (K + c - 1) / c;
}
size_t chain_splitk_scratch_floats(int M, int K0, int n0) {
    int chunk, rt;
    const int S = splitk_chunks(M, K0, &chunk, &rt);
    return (size_t)S * (size_t)M * (size_t)(n0 <= 80 ? 80 : 96);
}
// (the same nets at every batch size: what the large-batch 2-layer kernel is built for, <= 96 -> <= 64)
bool chain_wants_splitk(const ChainArgs& a) { return a.n_layers == 2 && a.M < 20480 && a.K0 >= 128 && a.n[0] <= 96 && a.n[1] <= 64; }

static int splitk_tn(const ChainArgs& a) { return a.n[0] <= 80 ? 5 : 6; }
static SplitkL1 splitk_l1_of(const ChainArgs& a, float* scratch, int* rt) {
    int chunk;
    const int S = splitk_chunks(a.M, a.K0, &chunk, rt);
    return SplitkL1{a.x, a.x_stride, a.K0, a.w[0], a.n[0], chunk, S, scratch};
}
static SplitkFin splitk_fin_of(const ChainArgs& a, const SplitkL1& l) {
    return SplitkFin{l.part, l.S, a.b[0], a.n[0], a.act[0], a.w[1], a.b[1], a.n[1], a.act[1], a.y, a.y_stride};
}
template <int TN>
static void launch_splitk_tn(const SplitkL1Pair& l, const SplitkFinPair& f, int n_chains, int rt, hipStream_t s) {
    const int S = n_chains == 2 && l.c[1].S > l.c[0].S ? l.c[1].S : l.c[0].S;
    const dim3 g1((uint32_t)((l.M + 16 * rt - 1) / (16 * rt)), (uint32_t)S, (uint32_t)n_chains), g2((uint32_t)((l.M + 15) / 16), 1u, (uint32_t)n_chains);
    if (rt == 2) hipLaunchKernelGGL((splitk_layer1_kernel<TN, 2>), g1, dim3(64), 0, s, l);
    else hipLaunchKernelGGL((splitk_layer1_kernel<TN, 1>), g1, dim3(64), 0, s, l);
    hipLaunchKernelGGL((splitk_finish_kernel<TN>), g2, dim3(256), 0, s, f);
}
hipError_t launch_chain_splitk(const ChainArgs& a, float* scratch, hipStream_t s) {
    int rt;
    SplitkL1Pair l{}; SplitkFinPair f{};
    l.M = f.M = a.M;
    l.c[0] = splitk_l1_of(a, scratch, &rt);
    f.c[0] = splitk_fin_of(a, l.c[0]);
    if (splitk_tn(a) == 5) launch_splitk_tn<5>(l, f, 1, rt, s); else launch_splitk_tn<6>(l, f, 1, rt, s);
    return hipGetLastError();
}
// two chains over the same rows side by side (both chain_wants_splitk, the same tile shape: the caller checks); scratch_b follows
// chain a's part of the scratch buffer
bool chain_pair_fits(const ChainArgs& a, const ChainArgs& b) {
    return a.M == b.M && chain_wants_splitk(a) && chain_wants_splitk(b) && splitk_tn(a) == splitk_tn(b);
}
hipError_t launch_chain_splitk_pair(const ChainArgs& a, const ChainArgs& b, float* scratch_a, float* scratch_b, const float* copy_src,
                                    int64_t copy_src_stride, float* copy_dst, int64_t copy_dst_stride, int copy_cols, hipStream_t s) {
    int rt, rt_b;
    SplitkL1Pair l{}; SplitkFinPair f{};
    l.M = f.M = a.M;
    l.c[0] = splitk_l1_of(a, scratch_a, &rt);
    l.c[1] = splitk_l1_of(b, scratch_b, &rt_b);                       // (rt depends on M only)
    f.c[0] = splitk_fin_of(a, l.c[0]);
    f.c[1] = splitk_fin_of(b, l.c[1]);
    f.copy_src = copy_src; f.copy_src_stride = copy_src_stride; f.copy_dst = copy_dst; f.copy_dst_stride = copy_dst_stride; f.copy_cols = copy_cols;
    if (splitk_tn(a) == 5) launch_splitk_tn<5>(l, f, 2, rt, s); else launch_splitk_tn<6>(l, f, 2, rt, s);
    return hipGetLastError();
}

// Small batches, the MLP with its head (124 -> 256 -> 160 -> 128 -> 2): 16 rows per workgroup, its EIGHT waves split every layer's
// OUTPUT tiles between them and hand the activations on through LDS (16 rows x <= 256 floats, ping-pong).  No weight staging and no
// barrier inside a layer: a wave streams the weight rows of its own tiles from L2 in the MFMA's operand layout, the x operand is one
// ds_read_b128 per 16 k.  chain16_kernel walks ~42 barrier-synchronised slab steps per 128 rows (67 us whatever the batch); here a
// layer is <= 2 tiles x 16 steps per wave (25 us at 512 rows, 30 us at 4 096).
#define MS_PITCH 260
__global__ void __launch_bounds__(512) mlp_small_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float buf[2][16][MS_PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, m = lane & 15u, g = lane >> 4;
    const uint32_t row0 = blockIdx.x * 16u;
    // the 16 input rows, zero-padded to a multiple of 16 columns
    const uint32_t kp0 = ((uint32_t)a.K0 + 15u) & ~15u;
    for (uint32_t i = tid; i < 16u * kp0; i += 512u) {
        const uint32_t r = i / kp0, k = i % kp0, row = row0 + r;
        buf[0][r][k] = (row < (uint32_t)a.M && k < (uint32_t)a.K0) ? a.x[(size_t)row * a.x_stride + k] : 0.0f;
    }
    __syncthreads();
    uint32_t cur = 0, K = (uint32_t)a.K0;
    for (int l = 0; l < a.n_layers; ++l) {
        const uint32_t N = (uint32_t)a.n[l], n_tiles = (N + 15u) >> 4, kp = (K + 15u) & ~15u;
        const float* __restrict__ w = a.w[l];
        const bool last = l == a.n_layers - 1;
        for (uint32_t t = wave; t < n_tiles; t += 8u) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            const uint32_t n = 16u * t + m;                          // the weight row this lane supplies
            const float* __restrict__ wr = w + (size_t)n * K;
            for (uint32_t k0 = 0; k0 < kp; k0 += 16u) {
                const uint32_t k = k0 + 4u * g;
                f32x4 wa = {0.0f, 0.0f, 0.0f, 0.0f};
                if (n < N) {
                    if (k + 4u <= K) wa = *reinterpret_cast<const f32x4u*>(wr + k);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) wa[e] = wr[k + e];
                    }
                }
                const f32x4 xb = *reinterpret_cast<const f32x4*>(&buf[cur][m][k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb[j], acc, 0, 0, 0);
            }
            // lane (m, g) holds features 16 t + 4 g + r of batch row m; features past N are written as zeros: they are the padding
            // columns the next layer's 16-wide k steps read
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t f = 16u * t + 4u * g + (uint32_t)r;
                const float v = f < N ? mlp_act(acc[r] + (a.b[l] ? a.b[l][f] : 0.0f), a.act[l]) : 0.0f;
                if (last) { if (f < N && row0 + m < (uint32_t)a.M) a.y[(size_t)(row0 + m) * a.y_stride + f] = v; }
                else buf[cur ^ 1u][m][f] = v;
            }
        }
        if (!last) {
            __syncthreads();
            cur ^= 1u;
            K = N;
        }
    }
}

// tile shapes instantiated: the reference's encoder (<= 80 -> <= 64, or <= 96 -> <= 64) and MLP (<= 256 -> <= 160 -> <= 128 -> <= 16,
// hidden activations none / LeakyReLU / ReLU); anything else: hipErrorInvalidValue (the caller runs layer by layer)
hipError_t launch_chain(const ChainArgs& a, hipStream_t s) {
    auto cheap = [](int act) { return act == 0 || act == 1 || act == 3; };
    // (the same nets at every batch size: what the large-batch kernel is built for)
    const bool mlp4 = a.n_layers == 4 && a.n[0] <= 256 && a.n[1] <= 160 && a.n[2] <= 128 && a.n[3] <= 16 && cheap(a.act[0]) && cheap(a.act[1]) &&
                      cheap(a.act[2]);
    if (mlp4 && a.M < 20480 && a.K0 <= 256) {
        hipLaunchKernelGGL(mlp_small_kernel, dim3((uint32_t)((a.M + 15) / 16)), dim3(512), 0, s, a);      // small batches: latency, not throughput
        return hipGetLastError();
    }
    const dim3 grid((uint32_t)((a.M + 127) / 128));
    if (a.n_layers == 2 && a.n[0] <= 80 && a.n[1] <= 64) {
        hipLaunchKernelGGL((chain16_kernel<5, 4, 0, 0>), grid, dim3(512), 0, s, a);
    } else if (a.n_layers == 2 && a.n[0] <= 96 && a.n[1] <= 64) {
        hipLaunchKernelGGL((chain16_kernel<6, 4, 0, 0>), grid, dim3(512), 0, s, a);
    } else if (mlp4) {
        hipLaunchKernelGGL((chain16_kernel<16, 10, 8, 1>), grid, dim3(512), 0, s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rover
