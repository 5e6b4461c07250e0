// Stand-alone reproducer for the run-to-run non-determinism of the streamed plane-split kernels (DESIGN_LOG.md, round 4,
// root-caused in round 4): a packed-fp32 VALU instruction (v_pk_fma_f32) that consumes the result of another packed-fp32
// instruction (v_pk_mul_f32) two issue slots earlier - hipcc (ROCm 7.2, gfx950) leaves ONE instruction between them - sees
// ZEROS instead of the producer's result in lanes 48..63 when a second wave shares the SIMD.
//
// The sequence is copied from rollout_fwd_kernel<GOPS_ENV_VEH3DOFCONTI, ..., SS> (veh_f_xu: hipcc's SLP vectoriser packs the
// scalar pose update; `hipcc -S`):
//      v_div_fmas_f32 v4, v4, v7, v10
//      v_pk_mul_f32 v[10:11], v[6:7], v[2:3] op_sel:[0,1] op_sel_hi:[0,0]              (sphi v, sphi u)
//      v_div_fixup_f32 v24, v4, v24, v28
//      v_pk_fma_f32 v[28:29], v[66:67], v[2:3], v[10:11] neg_lo:[0,0,1] neg_hi:[0,0,1]   cphi u - (sphi v)   <- v10 read as 0
//      v_pk_fma_f32 v[10:11], v[66:67], v[2:3], v[10:11] op_sel_hi:[0,1,1]
//      v_mov_b32_e32 v29, v11
//      v_pk_fma_f32 v[10:11], v[28:29], s[0:1], v[18:19] op_sel_hi:[1,0,1]               x' = x + 0.1 (.), y' = ...
// Every wave of the launch runs it in a loop (fixed registers, inline asm) between short MFMA bursts and checks x' against
// the same arithmetic on unpacked instructions; mismatches are counted per 16-lane quarter, together with how many of them
// equal "the subtrahend read as zero".  NOPS > 0 puts `s_nop NOPS-1` in front of the consuming v_pk_fma_f32.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/pk_hazard.hip -o tools/microbench/pk_hazard && tools/microbench/pk_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SEQ_HEAD                                                                                                              \
    "v_mov_b32 v124, %3\n\tv_mov_b32 v128, %4\n\tv_mov_b32 v106, %5\n\tv_mov_b32 v166, %6\n\tv_mov_b32 v167, %6\n\t"            \
    "v_mov_b32 v102, %7\n\tv_mov_b32 v103, %8\n\tv_mov_b32 v118, %9\n\tv_mov_b32 v119, %10\n\t"                                 \
    "s_mov_b32 s40, 0x3dcccccd\n\t"                                                                                           \
    "v_div_scale_f32 v104, s[44:45], v124, v124, v128\n\t"                                                                    \
    "v_rcp_f32_e32 v107, v104\n\t"                                                                                            \
    "s_nop 0\n\t"                                                                                                             \
    "v_fma_f32 v109, -v104, v107, 1.0\n\t"                                                                                    \
    "v_fmac_f32_e32 v107, v109, v107\n\t"                                                                                     \
    "v_div_scale_f32 v109, vcc, v128, v124, v128\n\t"                                                                         \
    "v_mul_f32_e32 v110, v109, v107\n\t"                                                                                      \
    "v_fma_f32 v111, -v104, v110, v109\n\t"                                                                                   \
    "v_fmac_f32_e32 v110, v111, v107\n\t"                                                                                     \
    "v_fma_f32 v104, -v104, v110, v109\n\t"                                                                                   \
    "v_div_fmas_f32 v104, v104, v107, v110\n\t"                                                                               \
    "v_pk_mul_f32 v[110:111], v[106:107], v[102:103] op_sel:[0,1] op_sel_hi:[0,0]\n\t"                                        \
    "v_div_fixup_f32 v124, v104, v124, v128\n\t"
#define SEQ_TAIL                                                                                                              \
    "v_pk_fma_f32 v[128:129], v[166:167], v[102:103], v[110:111] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"                           \
    "v_pk_fma_f32 v[110:111], v[166:167], v[102:103], v[110:111] op_sel_hi:[0,1,1]\n\t"                                       \
    "v_mov_b32_e32 v129, v111\n\t"                                                                                            \
    "v_pk_fma_f32 v[110:111], v[128:129], s[40:41], v[118:119] op_sel_hi:[1,0,1]\n\t"                                         \
    "s_nop 4\n\t"                                                                                                             \
    "v_mov_b32 %0, v124\n\tv_mov_b32 %1, v110\n\tv_mov_b32 %2, v111"
#define SEQ_IO                                                                                                                \
    : "=&v"(quot), "=&v"(sn0), "=&v"(sn1)                                                                                      \
    : "v"(den), "v"(num), "v"(sphi), "v"(cphi), "v"(u), "v"(v), "v"(x), "v"(y)                                                \
    : "v102", "v103", "v104", "v106", "v107", "v109", "v110", "v111", "v118", "v119", "v124", "v128", "v129", "v166", "v167", "s40", "s41",  \
      "s44", "s45", "vcc"

template <int NOPS>
__device__ __forceinline__ void veh_seq(float den, float num, float sphi, float cphi, float u, float v, float x, float y,
                                        float& quot, float& sn0, float& sn1) {
    if constexpr (NOPS == 0) asm volatile(SEQ_HEAD SEQ_TAIL SEQ_IO);
    else if constexpr (NOPS == 1) asm volatile(SEQ_HEAD "s_nop 0\n\t" SEQ_TAIL SEQ_IO);
    else if constexpr (NOPS == 2) asm volatile(SEQ_HEAD "s_nop 1\n\t" SEQ_TAIL SEQ_IO);
    else asm volatile(SEQ_HEAD "s_nop 3\n\t" SEQ_TAIL SEQ_IO);
}

template <int NOPS>
__global__ __launch_bounds__(256, 2) void pk_kernel(unsigned* __restrict__ bad, unsigned* __restrict__ zero, int iters, int nmfma, int stagger,
                                                    float* __restrict__ out) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * (lane + i)); B[i] = (__bf16)(0.02f * (lane - i)); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    unsigned nbad = 0, nzero = 0;
    float sink = 0.f;
    // odd workgroups start later: the two waves of a SIMD run the same stream a few instructions apart
    if (blockIdx.x & 1)
        for (int k = 0; k < stagger; ++k) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(sink));
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < nmfma; k += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc1, 0, 0, 0);
        }
        const float s = 1.f + 0.001f * (float)((it * 37 + lane * 11) & 1023);
        const float den = 1412.f * (4.f + s) + 21485.91f, num = 3000.f * s, sphi = 0.05f * s, cphi = 1.f - 0.001f * s;
        const float u = 4.f + s, v = 0.3f * s, x = 40.f * s, y = 2.f * s;
        float quot, sn0, sn1;
        veh_seq<NOPS>(den, num, sphi, cphi, u, v, x, y, quot, sn0, sn1);
        const float p = sphi * v;
        const float t0 = __builtin_fmaf(cphi, u, -p), t0z = cphi * u;
        const float want = __builtin_fmaf(t0, 0.1f, x), wantz = __builtin_fmaf(t0z, 0.1f, x);
        if (sn0 != want) { ++nbad; if (sn0 == wantz) ++nzero; }
        sink += quot + sn1;
    }
    if (nbad) { atomicAdd(&bad[lane], nbad); atomicAdd(&zero[lane], nzero); }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc0[0] + acc1[0] + sink + smem[0];
}

template <int NOPS>
static unsigned long long run(const char* what, int nmfma, int stagger, int wgs, size_t lds, int reps = 1) {
    unsigned *bad, *zero, h[64], hz[64];
    float* out;
    (void)hipMalloc(&bad, 256); (void)hipMalloc(&zero, 256); (void)hipMalloc(&out, 4);
    (void)hipMemset(bad, 0, 256); (void)hipMemset(zero, 0, 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pk_kernel<NOPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(pk_kernel<NOPS>, dim3(wgs), dim3(256), lds, 0, bad, zero, 20000, nmfma, stagger, out);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, bad, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(hz, zero, 256, hipMemcpyDeviceToHost);
    unsigned long long q[4] = {0, 0, 0, 0}, zq[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { q[l >> 4] += h[l]; zq[l >> 4] += hz[l]; }
    printf("%-26s mfma %2d stagger %3d x%d : wrong x' in lanes 0-15 / 16-31 / 32-47 / 48-63 = %llu / %llu / %llu / %llu   of which 'subtrahend read as 0': %llu / %llu / %llu / %llu\n",
           what, nmfma, stagger, reps, q[0], q[1], q[2], q[3], zq[0], zq[1], zq[2], zq[3]);
    (void)hipFree(bad); (void)hipFree(zero); (void)hipFree(out);
    return q[0] + q[1] + q[2] + q[3];
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; every wave runs the sequence 20000 times per launch (%d or %d waves x 64 lanes x 20000 results)\n", prop.gcnArchName, cus,
           8 * cus, 4 * cus);
    int best_nm = 16, best_st = 0;
    unsigned long long best = 0;
    for (int nm : {8, 16, 32, 64})
        for (int st = 0; st < 12; ++st) {
            const unsigned long long n = run<0>("2 WG/CU as compiled", nm, st, 2 * cus, 70 * 1024, 4);
            if (n > best) { best = n; best_nm = nm; best_st = st; }
        }
    printf("---- variants at mfma %d stagger %d, 20 launches each\n", best_nm, best_st);
    run<0>("2 WG/CU as compiled", best_nm, best_st, 2 * cus, 70 * 1024, 20);
    run<1>("2 WG/CU + s_nop 0", best_nm, best_st, 2 * cus, 70 * 1024, 20);
    run<2>("2 WG/CU + s_nop 1", best_nm, best_st, 2 * cus, 70 * 1024, 20);
    run<3>("2 WG/CU + s_nop 3", best_nm, best_st, 2 * cus, 70 * 1024, 20);
    run<0>("1 WG/CU as compiled", best_nm, best_st, cus, 100 * 1024, 40);
    run<0>("2 WG/CU, no MFMA", 0, best_st, 2 * cus, 70 * 1024, 20);
    return 0;
}
