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
    const int nt = (a.N + 31) / 32;
    switch (nt) {
        case 1: hipLaunchKernelGGL((linear_act_kernel<1, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
        case 2: hipLaunchKernelGGL((linear_act_kernel<2, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
        case 3: hipLaunchKernelGGL((linear_act_kernel<3, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
        case 4: hipLaunchKernelGGL((linear_act_kernel<4, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
        case 5: hipLaunchKernelGGL((linear_act_kernel<5, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
        case 6: hipLaunchKernelGGL((linear_act_kernel<6, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
        case 7: hipLaunchKernelGGL((linear_act_kernel<7, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
        case 8: hipLaunchKernelGGL((linear_act_kernel<8, NW>), dim3(nb), dim3(64 * NW), 0, s, a); break;
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

}  // namespace rover
