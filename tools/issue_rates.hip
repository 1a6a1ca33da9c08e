// issue_rates.hip — sustained instruction issue rates of one MI355X SIMD, measured (the denominators of bench.py's VALU roofline).
//
//   hipcc --offload-arch=gfx950 -O2 tools/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates > profiles/issue_rates.json
//
// For every instruction kind and W = 1, 2, 4, 8 waves per SIMD: 256 CUs x W workgroups of 4 waves (one wave per SIMD each), every wave
// runs ITERS iterations of an unrolled block of 64 independent instructions of that kind; the shader clock (s_memtime) is read at the
// start and at the end of every wave, the busiest SIMD's span decides:  cycles per instruction per SIMD = span / (W x ITERS x 64).
// "mix_*" rows interleave kinds the way cull_scan_kernel's scan loop does (vector and scalar instructions can issue in the same cycle
// from different waves).  The 100 MHz constant clock (s_memrealtime) next to it gives the shader clock the loop ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define ITERS 2000
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
struct Stamp { unsigned long long c0, c1, r0, r1; };

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

// 8 independent register sets a0..a7 so that back-to-back instructions never depend on each other
#define KERNEL(NAME, DECL, BODY, SINK)                                                                       \
    __global__ void __launch_bounds__(256) NAME(Stamp* out, float seed, unsigned iseed) {                                    \
        DECL                                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                        \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                      \
        const unsigned long long c0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < ITERS; ++it) { BODY }                                                          \
        const unsigned long long c1 = __builtin_readcyclecounter();                                          \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                      \
        SINK                                                                                                 \
        if ((threadIdx.x & 63u) == 0u) out[blockIdx.x * 4u + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};     \
    }

#define VDECL float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
              const float k = seed * 0.5f;
#define VSINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0].c0 = 1;
#define V8(OP) asm volatile(OP " %0, %0, %8, %0\n" OP " %1, %1, %8, %1\n" OP " %2, %2, %8, %2\n" OP " %3, %3, %8, %3\n"          \
                            OP " %4, %4, %8, %4\n" OP " %5, %5, %8, %5\n" OP " %6, %6, %8, %6\n" OP " %7, %7, %8, %7"          \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
KERNEL(k_v_fma_f32, VDECL, REP8(V8("v_fma_f32")), VSINK)
#define V8M(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n"                        \
                             OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8"                        \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
KERNEL(k_v_add_f32, VDECL, REP8(V8M("v_add_f32")), VSINK)
// v_fma_mix_f32 with an fp16 first source (low half), f32 second / third: what a lane-per-ray sphere test on fp16 records issues
#define V8X asm volatile("v_fma_mix_f32 %0, %8, -1.0, %0 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %1, %8, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"   \
                         "v_fma_mix_f32 %2, %8, -1.0, %2 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %3, %8, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"   \
                         "v_fma_mix_f32 %4, %8, -1.0, %4 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %5, %8, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"   \
                         "v_fma_mix_f32 %6, %8, -1.0, %6 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %7, %8, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
KERNEL(k_v_fma_mix_f32, VDECL, REP8(V8X), VSINK)
// v_addc_co_u32 x, vcc, x, x, vcc: shift a compare result into a per-lane bit mask
#define V8C asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\nv_addc_co_u32 %1, vcc, %1, %1, vcc\nv_addc_co_u32 %2, vcc, %2, %2, vcc\nv_addc_co_u32 %3, vcc, %3, %3, vcc\n" \
                         "v_addc_co_u32 %4, vcc, %4, %4, vcc\nv_addc_co_u32 %5, vcc, %5, %5, vcc\nv_addc_co_u32 %6, vcc, %6, %6, vcc\nv_addc_co_u32 %7, vcc, %7, %7, vcc"   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
KERNEL(k_v_addc, VDECL, REP8(V8C), VSINK)
// a compare writing VCC followed by the addc that consumes it (the pair as the mask build issues it)
#define V8CC asm volatile("v_cmp_gt_f32 vcc, %0, %8\nv_addc_co_u32 %1, vcc, %1, %1, vcc\nv_cmp_gt_f32 vcc, %2, %8\nv_addc_co_u32 %3, vcc, %3, %3, vcc\n"     \
                          "v_cmp_gt_f32 vcc, %4, %8\nv_addc_co_u32 %5, vcc, %5, %5, vcc\nv_cmp_gt_f32 vcc, %6, %8\nv_addc_co_u32 %7, vcc, %7, %7, vcc"       \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
KERNEL(k_cmp_addc, VDECL, REP8(V8CC), VSINK)
KERNEL(k_v_mul_f32, VDECL, REP8(V8M("v_mul_f32")), VSINK)
KERNEL(k_v_min_f32, VDECL, REP8(V8M("v_min_f32")), VSINK)
KERNEL(k_v_and_b32, VDECL, REP8(V8M("v_and_b32")), VSINK)
KERNEL(k_v_cndmask_b32, VDECL, REP8(V8M("v_cndmask_b32")), VSINK)
KERNEL(k_v_mbcnt_lo, VDECL, REP8(V8M("v_mbcnt_lo_u32_b32")), VSINK)
#define RL8 asm volatile("v_readlane_b32 s40, %0, 3\nv_readlane_b32 s41, %1, 5\nv_readlane_b32 s42, %2, 7\nv_readlane_b32 s43, %3, 9\n" \
                         "v_readlane_b32 s44, %4, 11\nv_readlane_b32 s45, %5, 13\nv_readlane_b32 s46, %6, 15\nv_readlane_b32 s47, %7, 17" \
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(k)                             \
                         : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
KERNEL(k_v_readlane, VDECL, REP8(RL8), VSINK)

#define PDECL f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
              const f2 k = {seed * 0.5f, seed * 0.25f};
#define PSINK if (a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y == 12345.678f) out[0].c0 = 1;
#define P8(OP) asm volatile(OP " %0, %0, %8, %0\n" OP " %1, %1, %8, %1\n" OP " %2, %2, %8, %2\n" OP " %3, %3, %8, %3\n"          \
                            OP " %4, %4, %8, %4\n" OP " %5, %5, %8, %5\n" OP " %6, %6, %8, %6\n" OP " %7, %7, %8, %7"          \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
KERNEL(k_v_pk_fma_f32, PDECL, REP8(P8("v_pk_fma_f32")), PSINK)
#define P8M(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n"                        \
                             OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8"                        \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
KERNEL(k_v_pk_mul_f32, PDECL, REP8(P8M("v_pk_mul_f32")), PSINK)
KERNEL(k_v_pk_add_f32, PDECL, REP8(P8M("v_pk_add_f32")), PSINK)

// v_cmp_gt_f32 writing an SGPR pair (what a ballot of a compare compiles to)
#define C8 asm volatile("v_cmp_gt_f32 s[40:41], %0, %8\nv_cmp_gt_f32 s[42:43], %1, %8\nv_cmp_gt_f32 s[44:45], %2, %8\nv_cmp_gt_f32 s[46:47], %3, %8\n" \
                        "v_cmp_gt_f32 s[48:49], %4, %8\nv_cmp_gt_f32 s[50:51], %5, %8\nv_cmp_gt_f32 s[52:53], %6, %8\nv_cmp_gt_f32 s[54:55], %7, %8"  \
                        :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(k)                                        \
                        : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
KERNEL(k_v_cmp_gt_f32, VDECL, REP8(C8), VSINK)

#define CV8 asm volatile("v_cvt_f32_f16 %0, %8\nv_cvt_f32_f16 %1, %8\nv_cvt_f32_f16 %2, %8\nv_cvt_f32_f16 %3, %8\n"             \
                         "v_cvt_f32_f16 %4, %8\nv_cvt_f32_f16 %5, %8\nv_cvt_f32_f16 %6, %8\nv_cvt_f32_f16 %7, %8"              \
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(k));
KERNEL(k_v_cvt_f32_f16, VDECL, REP8(CV8), VSINK)

// scalar ALU: independent 32-bit adds and 64-bit ands (the mask logic of the scan loop)
#define SDECL unsigned s0 = iseed, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
#define SSINK if (s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 == 0x12345678u) out[0].c0 = 1;
#define S8 asm volatile("s_add_u32 %0, %0, 3\ns_add_u32 %1, %1, 3\ns_add_u32 %2, %2, 3\ns_add_u32 %3, %3, 3\n"                 \
                        "s_add_u32 %4, %4, 3\ns_add_u32 %5, %5, 3\ns_add_u32 %6, %6, 3\ns_add_u32 %7, %7, 3"                   \
                        : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
KERNEL(k_s_add_u32, SDECL, REP8(S8), SSINK)

// the scan loop's mix: 2 packed vector instructions for every scalar one, and 1 compare in 6
#define MIXDECL PDECL SDECL
#define MIXSINK PSINK SSINK
#define MIX8 asm volatile("v_pk_fma_f32 %0, %0, %16, %0\ns_add_u32 %8, %8, 3\nv_pk_fma_f32 %1, %1, %16, %1\nv_pk_mul_f32 %2, %2, %16\ns_add_u32 %9, %9, 3\n"  \
                          "v_pk_fma_f32 %3, %3, %16, %3\nv_pk_add_f32 %4, %4, %16\ns_add_u32 %10, %10, 3\nv_pk_fma_f32 %5, %5, %16, %5\n"                   \
                          "v_pk_mul_f32 %6, %6, %16\ns_add_u32 %11, %11, 3\nv_pk_fma_f32 %7, %7, %16, %7"                                                   \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)             \
                          : "s"(s4), "s"(s5), "s"(s6), "s"(s7), "v"(k) : "scc");
KERNEL(k_mix_8pk_4s, MIXDECL, REP8(MIX8), MIXSINK)

struct Case { const char* name; void (*fn)(Stamp*, float, unsigned); int per_iter; const char* what; };

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;       // run one kind only (counter calibration under rocprofv3)
    Stamp* d = nullptr;
    const int max_blocks = 256 * 8;
    CHECK(hipMalloc((void**)&d, sizeof(Stamp) * max_blocks * 4));
    hipDeviceProp_t prop{};
    CHECK(hipGetDeviceProperties(&prop, 0));
    const Case cases[] = {
        {"v_fma_f32", k_v_fma_f32, 64, "wave64 f32 FMA"},
        {"v_add_f32", k_v_add_f32, 64, "wave64 f32 add"},
        {"v_fma_mix_f32", k_v_fma_mix_f32, 64, "mixed-precision FMA, fp16 first source"},
        {"v_addc_co_u32", k_v_addc, 64, "add with carry in / out through VCC"},
        {"cmp_addc", k_cmp_addc, 64, "v_cmp_gt_f32 vcc + v_addc_co_u32 pairs, counted as 2"},
        {"v_mul_f32", k_v_mul_f32, 64, "wave64 f32 multiply"},
        {"v_min_f32", k_v_min_f32, 64, "wave64 f32 min"},
        {"v_and_b32", k_v_and_b32, 64, "wave64 32-bit integer / logic"},
        {"v_cndmask_b32", k_v_cndmask_b32, 64, "select by vcc"},
        {"v_mbcnt_lo", k_v_mbcnt_lo, 64, "v_mbcnt_lo_u32_b32"},
        {"v_readlane", k_v_readlane, 64, "v_readlane_b32 into an SGPR"},
        {"v_pk_fma_f32", k_v_pk_fma_f32, 64, "packed f32 FMA (2 results per lane)"},
        {"v_pk_mul_f32", k_v_pk_mul_f32, 64, "packed f32 multiply"},
        {"v_pk_add_f32", k_v_pk_add_f32, 64, "packed f32 add"},
        {"v_cmp_gt_f32", k_v_cmp_gt_f32, 64, "f32 compare writing an SGPR pair"},
        {"v_cvt_f32_f16", k_v_cvt_f32_f16, 64, "fp16 -> f32 conversion"},
        {"s_add_u32", k_s_add_u32, 64, "scalar ALU"},
        {"mix_8pk_4s", k_mix_8pk_4s, 96, "8 packed vector + 4 scalar instructions interleaved, counted as 12"},
    };
    printf("{\n \"device\": \"%s\", \"cus\": %d, \"iters\": %d,\n \"method\": \"tools/issue_rates.hip: 256 CUs x W workgroups of 4 waves (one per SIMD); "
           "every wave issues iters x 64 independent instructions between two s_memtime reads; cycles_per_inst = the 95th-percentile wave span / (W x count): "
           "the cycles one SIMD needs per instruction when W waves share it\",\n \"rates\": {\n", prop.name, prop.multiProcessorCount, ITERS);
    bool first = true;
    for (const Case& c : cases) {
        if (only && std::string(only) != c.name) continue;
        for (int W : {1, 2, 4, 8}) {
            if (only && W != 8) continue;
            const int blocks = prop.multiProcessorCount * W;
            std::vector<Stamp> h((size_t)blocks * 4);
            for (int rep = 0; rep < 3; ++rep) {          // the last repetition counts (clocks ramped)
                hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f + rep, 7u + rep);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h.data(), d, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
            std::vector<double> span, mhz;
            for (const Stamp& s : h) {
                span.push_back((double)(s.c1 - s.c0));
                if (s.r1 > s.r0) mhz.push_back((double)(s.c1 - s.c0) / ((double)(s.r1 - s.r0) / 100.0));
            }
            std::sort(span.begin(), span.end());
            std::sort(mhz.begin(), mhz.end());
            const double med = span[span.size() / 2], hi = span[span.size() * 95 / 100], mx = span.back();
            const double n = (double)ITERS * c.per_iter;
            // sustained = the 95th percentile span: when the slowest waves of a SIMD end, its work is done (the fastest waves of a SIMD
            // end up to a third earlier: the arbiter is not fair, so the median under-states the time the SIMD was busy)
            printf("%s  \"%s@%d\": {\"cycles_per_inst\": %.4f, \"median\": %.4f, \"max\": %.4f, \"shader_mhz\": %.0f, \"what\": \"%s\"}",
                   first ? "" : ",\n", c.name, W, hi / (W * n), med / (W * n), mx / (W * n), mhz.empty() ? 0.0 : mhz[mhz.size() / 2], c.what);
            first = false;
        }
    }
    printf("\n }\n}\n");
    (void)hipFree(d);
    return 0;
}
