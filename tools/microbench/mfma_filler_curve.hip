// How many plain VALU instructions fit "for free" behind each MFMA of ONE wave on a gfx950 SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_filler_curve.hip -o tools/microbench/curve && tools/microbench/curve
// Every instruction of the measured loop is written as inline asm (hipcc neither reorders nor pads it): 8 MFMAs on 8
// independent accumulators per trip, K independent v_fma_f32 behind each of them, K = 0 .. 8.  Four waves (one per
// SIMD) run the same stream; shader cycles (s_memtime) per MFMA slot are printed for v_mfma_f32_16x16x32_bf16,
// v_mfma_f32_16x16x32_f16 and v_mfma_f32_16x16x4_f32, and for a second wave per SIMD running the SAME stream (8 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// one MFMA and its K fillers as ONE asm statement: nothing (no s_nop, no scalar loop bookkeeping) gets between them
#define F1 "v_fma_f32 %[v0], %[v0], %[c1], %[c2]\n\t"
#define F2 F1 "v_fma_f32 %[v1], %[v1], %[c1], %[c2]\n\t"
#define F3 F2 "v_fma_f32 %[v2], %[v2], %[c1], %[c2]\n\t"
#define F4 F3 "v_fma_f32 %[v3], %[v3], %[c1], %[c2]\n\t"
#define F5 F4 "v_fma_f32 %[v4], %[v4], %[c1], %[c2]\n\t"
#define F6 F5 "v_fma_f32 %[v5], %[v5], %[c1], %[c2]\n\t"
#define F7 F6 "v_fma_f32 %[v6], %[v6], %[c1], %[c2]\n\t"
#define F8 F7 "v_fma_f32 %[v7], %[v7], %[c1], %[c2]\n\t"
#define M0 "v_mfma_f32_16x16x32_bf16 %[acc], %[a], %[b], %[acc]\n\t"
#define M1 "v_mfma_f32_16x16x32_f16 %[acc], %[a], %[b], %[acc]\n\t"
#define M2 "v_mfma_f32_16x16x4_f32 %[acc], %[af], %[bf], %[acc]\n\t"
#define SLOT(M, F)                                                                                                         \
    asm volatile(M F : [acc] "+v"(acc), [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), \
                 [v5] "+v"(v[5]), [v6] "+v"(v[6]), [v7] "+v"(v[7])                                                          \
                 : [a] "v"(a), [b] "v"(b), [af] "v"(af), [bf] "v"(bf), [c1] "v"(c1), [c2] "v"(c2))
#define SLOTS(M)                                \
    if constexpr (K == 0) SLOT(M, "");          \
    else if constexpr (K == 1) SLOT(M, F1);     \
    else if constexpr (K == 2) SLOT(M, F2);     \
    else if constexpr (K == 3) SLOT(M, F3);     \
    else if constexpr (K == 4) SLOT(M, F4);     \
    else if constexpr (K == 5) SLOT(M, F5);     \
    else if constexpr (K == 6) SLOT(M, F6);     \
    else if constexpr (K == 7) SLOT(M, F7);     \
    else SLOT(M, F8);

template <int KIND, int K>
__device__ __forceinline__ void slot(f32x4& acc, float (&v)[8], const bf16x8& a, const bf16x8& b, float af, float bf, float c1, float c2) {
    if constexpr (KIND == 0) { SLOTS(M0) } else if constexpr (KIND == 1) { SLOTS(M1) } else { SLOTS(M2) }
}

template <int KIND, int K>
__global__ __launch_bounds__(512) void k(int iters, float* out, long long* cyc) {
    f32x4 acc[8] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((threadIdx.x & 7) + i); b[i] = (__bf16)(float)(i + 1); }
    const float af = threadIdx.x * 0.5f, bf = 1.25f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) slot<KIND, K>(acc[q], v, a, b, af, bf, c1, c2);
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    for (int q = 0; q < 8; ++q) s += v[q];
    out[threadIdx.x] = s;
}

template <int KIND, int K>
void one(int waves, float* out, long long* cyc, double* res) {
    const int iters = 2048;
    long long h[8];
    hipLaunchKernelGGL((k<KIND, K>), dim3(1), dim3(64 * waves), 0, 0, iters, out, cyc);
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    res[K] = (double)mx / (iters * 8.0);
}

template <int KIND>
void curve(const char* name, float* out, long long* cyc) {
    for (int waves : {4, 8}) {
        double r[9];
        one<KIND, 0>(waves, out, cyc, r); one<KIND, 1>(waves, out, cyc, r); one<KIND, 2>(waves, out, cyc, r);
        one<KIND, 3>(waves, out, cyc, r); one<KIND, 4>(waves, out, cyc, r); one<KIND, 5>(waves, out, cyc, r);
        one<KIND, 6>(waves, out, cyc, r); one<KIND, 7>(waves, out, cyc, r); one<KIND, 8>(waves, out, cyc, r);
        printf("%-28s %d wave(s)/SIMD, cycles per MFMA slot of one wave with K = 0..8 v_fma_f32 behind each MFMA:", name, waves / 4);
        for (int i = 0; i <= 8; ++i) printf(" %6.1f", r[i]);
        printf("\n");
    }
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
    curve<0>("v_mfma_f32_16x16x32_bf16", out, cyc);
    curve<1>("v_mfma_f32_16x16x32_f16", out, cyc);
    curve<2>("v_mfma_f32_16x16x4_f32", out, cyc);
    return 0;
}
