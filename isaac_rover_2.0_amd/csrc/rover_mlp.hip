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
// A CHAIN of up to four layers in one kernel (an encoder 634 -> 80 -> 60, or the MLP 124 -> 256 -> 160 -> 128 -> 2 of
// learning/model.py:122-150,176-195): only the chain's input and its last output touch HBM.
//
// The MFMA runs transposed, D[n][m] = W[n][k] X^T[k][m]: the 32 x 32 accumulator tile then has the batch row m on the LANE
// (col = lane & 31) and the output feature n in the registers (n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r < 16) — exactly
// the B-operand shape of the next layer's MFMA (B[k][m]: lane half h takes k-step entry h).  Register r of an input tile is
// k-step r of the next layer, fed with the weight column kmap(r, h) as its A operand: activations never leave the
// registers, no LDS transpose, and only the weight slabs (N x 32 floats per input tile) are staged through LDS.
// Layer 1 takes its B operand from an LDS slab of the input rows (128 rows x 32 k, coalesced loads, next slab prefetched
// into registers under the MFMAs).  Every wave owns 32 batch rows through the whole chain.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mlp_kmap(uint32_t r, uint32_t h) { return (r & 3u) + 8u * (r >> 2) + 4u * h; }

// bias + activation on a layer's accumulator tiles.  The activation is a template parameter of the unrolled body and the
// (wave-uniform) switch sits outside it: with the switch inside, every one of the T x 16 elements carried tanhf, expm1f
// and the rest — 49 000 instructions for the 4-layer chain, far beyond the instruction cache.
template <int T, int ACT>
__device__ __forceinline__ void chain_bias_act_as(f32x16 (&acc)[T], const float* __restrict__ bias, int n_valid, uint32_t lane) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t n = 32u * t + mlp_kmap((uint32_t)r, lane >> 5);
            const float b = (bias && n < (uint32_t)n_valid) ? bias[n] : 0.0f;
            acc[t][r] = n < (uint32_t)n_valid ? mlp_act(acc[t][r] + b, ACT) : 0.0f;
        }
}
template <int T>
__device__ __forceinline__ void chain_bias_act(f32x16 (&acc)[T], const float* __restrict__ bias, int n_valid, int act, uint32_t lane) {
    switch (act) {
        case 1: chain_bias_act_as<T, 1>(acc, bias, n_valid, lane); break;
        case 2: chain_bias_act_as<T, 2>(acc, bias, n_valid, lane); break;
        case 3: chain_bias_act_as<T, 3>(acc, bias, n_valid, lane); break;
        case 4: chain_bias_act_as<T, 4>(acc, bias, n_valid, lane); break;
        default: chain_bias_act_as<T, 0>(acc, bias, n_valid, lane); break;
    }
}

// one register-fed layer: out[TO] (+)= W[:, cols of `in`] . in[TI]; W is [n_out][ldw] row-major, its column for feature f of the
// input is col0[tile] + f (f < cols[tile]; the rest of a tile is padding).  The weight slab of input tile ti + 1 is fetched into
// registers before the MFMAs of tile ti and stored after them — with DB to the OTHER LDS buffer (one barrier per slab; the
// 4-layer chain runs one wave per SIMD anyway and nothing else hides the latency), without DB to the same buffer between two
// barriers (the 2-layer encoder chain: 50 KB of LDS, three workgroups per CU).
template <int TI, int TO, bool DB>
__device__ __forceinline__ void chain_layer(const f32x16 (&in)[TI], f32x16 (&out)[TO], const float* __restrict__ w, int ldw, int n_out,
                                            const int (&col0)[TI], const int (&cols)[TI], float* __restrict__ Ws /* 2 buffers */, uint32_t tid,
                                            uint32_t lane) {
    const uint32_t sc = tid & 31u, sr = tid >> 5;
    constexpr uint32_t BUF = 256u * MLP_PITCH;
    float pw[TO * 4];
    auto fetch = [&](int ti) {
#pragma unroll
        for (int j = 0; j < TO * 4; ++j) {                           // slab [TO * 32 rows][32 features of input tile ti]
            const uint32_t n = sr + 8u * j;
            pw[j] = (n < (uint32_t)n_out && (int)sc < cols[ti]) ? w[(size_t)n * ldw + col0[ti] + sc] : 0.0f;
        }
    };
    auto stash = [&](uint32_t buf) {
#pragma unroll
        for (int j = 0; j < TO * 4; ++j) Ws[buf * BUF + (sr + 8u * j) * MLP_PITCH + sc] = pw[j];
    };
    __syncthreads();                                                 // the previous layer's readers are done with both buffers
    fetch(0);
    stash(0);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
        const uint32_t buf = DB ? ((uint32_t)ti & 1u) : 0u;
        if (ti + 1 < TI) fetch(ti + 1);                              // in flight during the MFMAs below
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t kc = mlp_kmap((uint32_t)r, lane >> 5);
#pragma unroll
            for (int to = 0; to < TO; ++to) {
                const float av = Ws[buf * BUF + (to * 32u + (lane & 31u)) * MLP_PITCH + kc];
                out[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, in[ti][r], out[to], 0, 0, 0);
            }
        }
        if (ti + 1 < TI) {
            if (!DB) __syncthreads();                                // one buffer: this slab's readers first
            stash(DB ? buf ^ 1u : 0u);                               // (two buffers: nobody reads the other one now)
            __syncthreads();
        }
    }
}

template <int T>
__device__ __forceinline__ void chain_zero(f32x16 (&acc)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

// the chain's last activations -> y[m][n] through an LDS transpose (coalesced 128-byte row segments)
template <int T>
__device__ __forceinline__ void chain_store(const f32x16 (&acc)[T], float* __restrict__ y, int64_t y_stride, int n_valid, int M, uint32_t row0,
                                            float* __restrict__ Xs, uint32_t lane, uint32_t wave) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(wave * 32u + (lane & 31u)) * MLP_PITCH + mlp_kmap((uint32_t)r, lane >> 5)] = acc[t][r];
        __syncthreads();
        const uint32_t col = 32u * t + (lane & 31u);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t rr = 2u * j + (lane >> 5), row = row0 + wave * 32u + rr;
            if (col < (uint32_t)n_valid && row < (uint32_t)M) y[(size_t)row * y_stride + col] = Xs[(wave * 32u + rr) * MLP_PITCH + (lane & 31u)];
        }
    }
}

// T1..T4: 32-wide output tiles of the layers (0 = layer absent).  Layer-1 inputs are x[:, 0:K0]; the layers after the
// first see their predecessor's outputs, except that layer 1 may be SKIPPED (T1 = 0 is not allowed; see CONCAT below).
// CONCAT (the MLP of model.py:185-195): the input row is [p proprioceptive | f encoder-0 | f encoder-1 features] and K0 = p + 2 f
// is not a multiple of 32; it is simply streamed as layer 1's k range (zero-padded past K0), nothing special is needed.
template <int T1, int T2, int T3, int T4>
__global__ void __launch_bounds__(256) chain_kernel(ChainArgs a) {
    __shared__ float Xs[128 * MLP_PITCH];
    constexpr bool DB = T3 != 0;                                   // see chain_layer
    __shared__ float Ws[(DB ? 2 : 1) * 256 * MLP_PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t row0 = blockIdx.x * 128u;
    const uint32_t sc = tid & 31u, sr = tid >> 5;
    // ---- layer 1: B operand from the LDS slab of the input rows
    f32x16 a1[T1];
    chain_zero(a1);
    {
        float pa[16], pw[T1 * 4];
        auto fetch = [&](uint32_t k0) {
            const uint32_t gk = k0 + sc;
            const bool kin = gk < (uint32_t)a.K0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t gr = row0 + sr + 8u * j;
                pa[j] = (kin && gr < (uint32_t)a.M) ? a.x[(size_t)gr * a.x_stride + gk] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < T1 * 4; ++j) {
                const uint32_t n = sr + 8u * j;
                pw[j] = (kin && n < (uint32_t)a.n[0]) ? a.w[0][(size_t)n * a.K0 + gk] : 0.0f;
            }
        };
        fetch(0);
        for (uint32_t k0 = 0; k0 < (uint32_t)a.K0; k0 += MLP_BK) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) Xs[(sr + 8u * j) * MLP_PITCH + sc] = pa[j];
#pragma unroll
            for (int j = 0; j < T1 * 4; ++j) Ws[(sr + 8u * j) * MLP_PITCH + sc] = pw[j];
            __syncthreads();
            if (k0 + MLP_BK < (uint32_t)a.K0) fetch(k0 + MLP_BK);        // in flight during the MFMAs below
#pragma unroll 4
            for (uint32_t kk = 0; kk < MLP_BK; kk += 2) {
                const float bv = Xs[(wave * 32u + (lane & 31u)) * MLP_PITCH + kk + (lane >> 5)];
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    const float av = Ws[(t * 32u + (lane & 31u)) * MLP_PITCH + kk + (lane >> 5)];
                    a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a1[t], 0, 0, 0);
                }
            }
        }
    }
    if constexpr (T2 == 0) {
        chain_bias_act(a1, a.b[0], a.n[0], a.act[0], lane);
        chain_store(a1, a.y, a.y_stride, a.n[0], a.M, row0, Xs, lane, wave);
    } else {
        chain_bias_act(a1, a.b[0], a.n[0], a.act[0], lane);
        int c0[T1], cn[T1];
#pragma unroll
        for (int t = 0; t < T1; ++t) { c0[t] = 32 * t; cn[t] = a.n[0] - 32 * t < 32 ? (a.n[0] - 32 * t > 0 ? a.n[0] - 32 * t : 0) : 32; }
        f32x16 a2[T2];
        chain_zero(a2);
        chain_layer<T1, T2, DB>(a1, a2, a.w[1], a.n[0], a.n[1], c0, cn, Ws, tid, lane);
        chain_bias_act(a2, a.b[1], a.n[1], a.act[1], lane);
        if constexpr (T3 == 0) {
            chain_store(a2, a.y, a.y_stride, a.n[1], a.M, row0, Xs, lane, wave);
        } else {
            int d0[T2], dn[T2];
#pragma unroll
            for (int t = 0; t < T2; ++t) { d0[t] = 32 * t; dn[t] = a.n[1] - 32 * t < 32 ? (a.n[1] - 32 * t > 0 ? a.n[1] - 32 * t : 0) : 32; }
            f32x16 a3[T3];
            chain_zero(a3);
            chain_layer<T2, T3, DB>(a2, a3, a.w[2], a.n[1], a.n[2], d0, dn, Ws, tid, lane);
            chain_bias_act(a3, a.b[2], a.n[2], a.act[2], lane);
            if constexpr (T4 == 0) {
                chain_store(a3, a.y, a.y_stride, a.n[2], a.M, row0, Xs, lane, wave);
            } else {
                int e0[T3], en[T3];
#pragma unroll
                for (int t = 0; t < T3; ++t) { e0[t] = 32 * t; en[t] = a.n[2] - 32 * t < 32 ? (a.n[2] - 32 * t > 0 ? a.n[2] - 32 * t : 0) : 32; }
                f32x16 a4[T4];
                chain_zero(a4);
                chain_layer<T3, T4, DB>(a3, a4, a.w[3], a.n[2], a.n[3], e0, en, Ws, tid, lane);
                chain_bias_act(a4, a.b[3], a.n[3], a.act[3], lane);
                chain_store(a4, a.y, a.y_stride, a.n[3], a.M, row0, Xs, lane, wave);
            }
        }
    }
}

// tile shapes instantiated: the reference's encoder (<= 96 -> <= 64) and MLP (<= 256 -> <= 160 -> <= 128 -> <= 32)
hipError_t launch_chain(const ChainArgs& a, hipStream_t s) {
    const dim3 grid((uint32_t)((a.M + 127) / 128));
    auto tiles = [](int n) { return (n + 31) / 32; };
    if (a.n_layers == 2 && tiles(a.n[0]) <= 3 && tiles(a.n[1]) <= 2) {
        hipLaunchKernelGGL((chain_kernel<3, 2, 0, 0>), grid, dim3(256), 0, s, a);
    } else if (a.n_layers == 4 && tiles(a.n[0]) <= 8 && tiles(a.n[1]) <= 5 && tiles(a.n[2]) <= 4 && tiles(a.n[3]) <= 1) {
        hipLaunchKernelGGL((chain_kernel<8, 5, 4, 1>), grid, dim3(256), 0, s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rover
