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
// staged through LDS with a 33-word pitch (conflict-free column reads); each wave keeps N/32 accumulator tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rover_internal.h"

namespace rover {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MLP_BM 128
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

template <int NT>
__global__ void __launch_bounds__(256) linear_act_kernel(LinearArgs a) {
    __shared__ float As[MLP_BM * MLP_PITCH];
    __shared__ float Ws[NT * 32 * MLP_PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t row0 = blockIdx.x * MLP_BM;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const uint32_t ar = lane & 31u, ak = lane >> 5;                  // A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
    for (uint32_t k0 = 0; k0 < (uint32_t)a.K; k0 += MLP_BK) {
        // stage the k-slab: rows x 32 contiguous floats (128 B per row, coalesced), zero-padded past M / N / K
        for (uint32_t e = tid; e < MLP_BM * MLP_BK; e += 256) {
            uint32_t r = e >> 5, c = e & 31u;
            uint32_t gr = row0 + r, gk = k0 + c;
            As[r * MLP_PITCH + c] = (gr < (uint32_t)a.M && gk < (uint32_t)a.K) ? a.x[(size_t)gr * a.x_stride + gk] : 0.0f;
        }
        for (uint32_t e = tid; e < NT * 32 * MLP_BK; e += 256) {
            uint32_t n = e >> 5, c = e & 31u;
            uint32_t gk = k0 + c;
            Ws[n * MLP_PITCH + c] = (n < (uint32_t)a.N && gk < (uint32_t)a.K) ? a.w[(size_t)n * a.K + gk] : 0.0f;
        }
        __syncthreads();
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
        const uint32_t col = t * 32u + (lane & 31u);
        if (col >= (uint32_t)a.N) continue;
        const float bias = a.b ? a.b[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = row0 + wave * 32u + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * (lane >> 5);
            if (row < (uint32_t)a.M) a.y[(size_t)row * a.y_stride + col] = mlp_act(acc[t][r] + bias, a.act);
        }
    }
}

hipError_t launch_linear_act(const LinearArgs& a, hipStream_t s) {
    const uint32_t nb = (uint32_t)((a.M + MLP_BM - 1) / MLP_BM);
    const int nt = (a.N + 31) / 32;
    switch (nt) {
        case 1: hipLaunchKernelGGL(linear_act_kernel<1>, dim3(nb), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(linear_act_kernel<2>, dim3(nb), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(linear_act_kernel<3>, dim3(nb), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(linear_act_kernel<4>, dim3(nb), dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL(linear_act_kernel<5>, dim3(nb), dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL(linear_act_kernel<6>, dim3(nb), dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL(linear_act_kernel<7>, dim3(nb), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL(linear_act_kernel<8>, dim3(nb), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace rover
