// Does a gfx950 SIMD overlap a v_mfma stream with plain VALU work - of ANOTHER wave on the same SIMD, or of the same wave?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o tools/microbench/ovl && tools/microbench/ovl
// One workgroup of 8 waves on one CU: waves 0-3 (one per SIMD) issue back-to-back MFMAs on 8 independent accumulators,
// waves 4-7 (the second wave of each SIMD) issue independent v_fma_f32 (64 per loop trip).  Shader cycles (s_memtime)
// of wave 0 / wave 4 for: MFMA alone, VALU alone, both; with the VALU waves at s_setprio 0 or 3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool F32>
__device__ __forceinline__ void mfma8(f32x4 (&acc)[8], const bf16x8& a, const bf16x8& b, float af, float bf) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if constexpr (F32) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[q], 0, 0, 0);
        else acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
    }
}

template <bool F32, int SAME>   // SAME: VALU instructions interleaved behind every MFMA of the SAME wave (0 = two-wave test)
__global__ __launch_bounds__(512) void k(int mode, int iters, int valu_per_mfma, int prio, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4, valu_wave = wave >= 4;
    f32x4 acc[8] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    const float af = threadIdx.x * 0.5f, bf = 1.25f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    if (valu_wave && prio) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const long long t0 = clock64();
    if (SAME > 0) {
        if (mfma_wave) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (F32) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[q], 0, 0, 0);
                    else acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < SAME; ++r) v[(q * SAME + r) & 15] = __builtin_fmaf(v[(q * SAME + r) & 15], 1.0001f, 0.5f);
                }
            }
        }
    } else {
        if (mfma_wave && (mode & 1))
            for (int it = 0; it < iters; ++it) mfma8<F32>(acc, a, b, af, bf);
        if (valu_wave && (mode & 2)) {
            for (int it = 0; it < (iters * valu_per_mfma) / 8; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
            }
        }
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    for (int q = 0; q < 16; ++q) s += v[q];
    out[threadIdx.x] = s;
}

template <bool F32>
void run(const char* name, float* out, long long* cyc) {
    long long h[8];
    const int iters = 4096;   // 8 MFMAs per trip
    printf("---- %s ----\n", name);
    for (int prio : {0, 1})
        for (int vpm : {1, 3}) {
            double t[4];
            for (int mode : {1, 2, 3}) {
                hipLaunchKernelGGL((k<F32, 0>), dim3(1), dim3(512), 0, 0, mode, iters, vpm, prio, out, cyc);
                (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
                t[mode] = (double)(mode == 1 ? h[0] : (mode == 2 ? h[4] : (h[0] > h[4] ? h[0] : h[4]))) / (iters * 8.0);
                if (mode == 3) printf("  (both: mfma wave done after %.1f, valu wave after %.1f)", h[0] / (iters * 8.0), h[4] / (iters * 8.0));
            }
            printf("\n%d VALU per MFMA, VALU wave prio %d: per MFMA slot  mfma alone %.1f  valu alone %.1f  both %.1f  (sum %.1f, max %.1f)\n",
                   vpm, prio ? 3 : 0, t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
        }
    hipLaunchKernelGGL((k<F32, 2>), dim3(1), dim3(512), 0, 0, 3, iters, 0, 0, out, cyc);
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("same wave, 2 VALU behind every MFMA: %.1f cycles per MFMA\n", (double)h[0] / (iters * 8.0));
    hipLaunchKernelGGL((k<F32, 6>), dim3(1), dim3(512), 0, 0, 3, iters, 0, 0, out, cyc);
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("same wave, 6 VALU behind every MFMA: %.1f cycles per MFMA\n", (double)h[0] / (iters * 8.0));
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
    run<false>("v_mfma_f32_16x16x32_bf16 (16 cycles)", out, cyc);
    run<true>("v_mfma_f32_16x16x4_f32 (32 cycles)", out, cyc);
    return 0;
}
