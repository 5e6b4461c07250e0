// Support kernels around the rollout: MFMA-fragment weight packing, the veh3dofconti reference
// table, the weight-gradient GEMMs (dW = delta^T * input over all B*H samples), partial-sum
// reduction, and the single-step env-model entry point.
#include <algorithm>
#include "common.h"
#include "env_models.h"

// ---------------------------------------------------------------------------------------------
// Weight packing.  For Linear layer W [N][K] (torch layout):
//   fwd:  element ((nt*kch + c)*64 + lane)*4 + i  =  W[16nt + (lane&15)][16c + 4(lane>>4) + i]
//   bwd:  element ((nt*kch + c)*64 + lane)*4 + i  =  W[16c + 4(lane>>4) + i][16nt + (lane&15)]
// (zero where the input index is in the padding), so that lane `lane` fetches with one dwordx4 the
// B operands of four consecutive v_mfma_f32_16x16x4_f32.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_element(const float* __restrict__ W, int N, int K, int Kp,
                                             float* __restrict__ wp, float* __restrict__ wpt, int e) {
    if (e >= N * Kp) return;
    const int i = e & 3, lane = (e >> 2) & 63, blk = e >> 8;
    {   // forward: n tiles over N, chunks over Kp
        const int kch = Kp >> 4, nt = blk / kch, c = blk - nt * kch;
        const int n = 16 * nt + (lane & 15), k = 16 * c + 4 * (lane >> 4) + i;
        wp[e] = (k < K) ? W[(size_t)n * K + k] : 0.f;
    }
    {   // backward: n tiles over Kp (input index), chunks over N (output index)
        const int kch = N >> 4, nt = blk / kch, c = blk - nt * kch;
        const int kin = 16 * nt + (lane & 15), nout = 16 * c + 4 * (lane >> 4) + i;
        wpt[e] = (kin < K) ? W[(size_t)nout * K + kin] : 0.f;
    }
}

// Half-precision packings (rollout_f16.h: the weights are the A operand of v_mfma_f32_16x16x32_f16, 8 halfs
// per lane).  Element e = ((blk * 64 + lane) * 8 + i8):
//   wph[j]  (forward)   blk = (4q + jt) * kch + c, kch = kp32 / 32:  W[64q + 16(r>>2) + 4jt + (r&3)][32c + 8(lane>>4) + i8], r = lane & 15
//   wpth[j] (backward, j >= 1) same with roles swapped: rows = inputs of the layer (quads over dims[j]),
//           chunks over its outputs:  W[32c + 8(lane>>4) + i8][64q + 16(r>>2) + 4jt + (r&3)]
//   wpth[0] (backward, input adjoint) plain 16-row tiles over the 16-padded inputs:  W[32c + 8(lane>>4) + i8][16nt + r]
__device__ __forceinline__ void pack_element_h(const float* __restrict__ W, int N, int K, int Kp16, int Kp32, int layer,
                                               _Float16* __restrict__ wph, _Float16* __restrict__ wpth, int e) {
    const int i8 = e & 7, lane = (e >> 3) & 63, blk = e >> 9, r = lane & 15, kg = lane >> 4;
    if (e < N * Kp32) {   // forward
        const int kch = Kp32 >> 5, c = blk % kch, t = blk / kch, jt = t & 3, q = t >> 2;
        const int n = 64 * q + 16 * (r >> 2) + 4 * jt + (r & 3), k = 32 * c + 8 * kg + i8;
        wph[e] = (_Float16)((k < K) ? W[(size_t)n * K + k] : 0.f);
    }
    const int kchN = N >> 5;   // outputs are a multiple of 64
    if (layer >= 1) {
        if (e < N * K) {      // K = width of the previous hidden layer, a multiple of 64
            const int c = blk % kchN, t = blk / kchN, jt = t & 3, q = t >> 2;
            const int kin = 64 * q + 16 * (r >> 2) + 4 * jt + (r & 3), nout = 32 * c + 8 * kg + i8;
            wpth[e] = (_Float16)W[(size_t)nout * K + kin];
        }
    } else if (e < N * Kp16) {
        const int c = blk % kchN, nt = blk / kchN;
        const int kin = 16 * nt + r, nout = 32 * c + 8 * kg + i8;
        wpth[e] = (_Float16)((kin < K) ? W[(size_t)nout * K + kin] : 0.f);
    }
}

// Plane-split operands (common.h SplitDev).  One block packs one n-tile: 16 columns x 32 kc contraction slots.
//   forward,  layer j: column = OUTPUT feature 16 nt + col, slot s <-> input feature (j == 0 ? s : split_perm(s))
//   backward, layer j: column = INPUT feature 16 nt + col,  slot s <-> output feature split_perm(s)
// element ((c * 64 + lane) * 8 + i) of the tile = (column lane & 15, slot 32 c + 8 (lane >> 4) + i): the B operand of
// v_mfma_f32_16x16x32_{bf16,f16} for chunk c.  w1 = bf16(w) (round to nearest even), rf = f16((w - w1) * s_r) with the
// power of two s_r that brings the tile's largest residual into [2^13, 2^14); inv[nt] = 1 / s_r.
__device__ __forceinline__ float bf16_round(float w) {
    unsigned u = __float_as_uint(w);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ void pack_split_tile(const float* __restrict__ W, int N, int K, bool transposed, bool perm_slots,
                                                int kc, int nt, unsigned short* __restrict__ w1, _Float16* __restrict__ rf,
                                                float* __restrict__ inv) {
    __shared__ float red[256];
    const int tid = threadIdx.x, total = 16 * 32 * kc;
    auto wval = [&](int e) -> float {
        const int i = e & 7, lane = (e >> 3) & 63, c = e >> 9;
        const int col = 16 * nt + (lane & 15), slot = 32 * c + 8 * (lane >> 4) + i;
        const int f = perm_slots ? split_perm(slot) : slot;
        if (!transposed) return (col < N && f < K) ? W[(size_t)col * K + f] : 0.f;   // W[out = col][in = f]
        return (f < N && col < K) ? W[(size_t)f * K + col] : 0.f;                     // W[out = f][in = col]
    };
#if GOPS_SPLIT_F16X2
    // two half planes of w s_w: wh = f16(w s_w) (round to nearest), wl = f16((w s_w - wh) 2^11); s_w = the power of two that brings the
    // tile's largest |w| into [2^13, 2^14) - fp32's exponent range for the weights, and small-weight tiles use the half's normal range.
    // The tile's <= 16 elements per thread are loaded ONCE, all loads in flight together (a load - use - load loop costs one L2 round
    // trip per element: this kernel sits on the critical path of every update, round 5: 11 -> 5 us).
    constexpr int PER = 16;   // kc <= 8: 16 * 32 * 8 / 256
    float v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + 256 * q;
        v[q] = e < total ? wval(e) : 0.f;
    }
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q) mx = fmaxf(mx, fabsf(v[q]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s_w = 1.f;
    if (mx > 0.f && mx < 3.0e38f) {
        int ex;
        (void)frexpf(mx, &ex);   // mx = f * 2^ex, f in [0.5, 1)
        s_w = ldexpf(1.f, 14 - ex);
    }
    const size_t off = (size_t)nt * total;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + 256 * q;
        if (e < total) {
            const float w = v[q] * s_w;   // (exact: a power of two)
            const _Float16 h = (_Float16)w;
            w1[off + e] = __builtin_bit_cast(unsigned short, h);
            rf[off + e] = (_Float16)((w - (float)h) * SPLIT_LO_SCALE);
        }
    }
    if (tid == 0) inv[nt] = 1.f / s_w;
}
#else
    float mx = 0.f;
    for (int e = tid; e < total; e += 256) {
        const float w = wval(e);
        mx = fmaxf(mx, fabsf(w - bf16_round(w)));
    }
    red[tid] = mx;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] = fmaxf(red[tid], red[tid + h]);
        __syncthreads();
    }
    mx = red[0];
    float s_r = 1.f;
    if (mx > 0.f && mx < 3.0e38f) {
        int ex;
        (void)frexpf(mx, &ex);   // mx = f * 2^ex, f in [0.5, 1)
        s_r = ldexpf(1.f, 14 - ex);
    }
    const size_t off = (size_t)nt * total;
    for (int e = tid; e < total; e += 256) {
        const float w = wval(e), h = bf16_round(w);
        w1[off + e] = (unsigned short)(__float_as_uint(h) >> 16);
        rf[off + e] = (_Float16)((w - h) * s_r);
    }
    if (tid == 0) inv[nt] = 1.f / s_r;
}
#endif
// The fp32 MFMA-fragment packings (wp / wpt) of the POLICY are read by the exact-fp32 kernels only.  A veh3dofconti launch on the
// register-stationary plane-split kernels never reaches one - its sweep is the stationary plane-split sweep, and the calls that
// would divert a backward to the fp32 kernels after the forward has planned (terminal adjoints / gops_rollout_backward_adj,
// ActionRepeat, open loop: the EXT / GEN instantiations) do not exist for the vehicle models - so the 384 blocks of that packing
// (obs -> 256 -> 256) would be written for nobody.  Every other launch keeps it (an lq / idpendulum rollout may get an `_adj`
// backward, a value / MLP batch an `ext_delta` one, decided only at backward time).  The tail value net always keeps its packing
// (exact-fp32 tail evaluation).  Returns the first net to pack.
__host__ __device__ inline int fp32_pack_first_net(const RolloutParams& p) {
    return (!p.f16 && p.sp.on && p.env.kind == GOPS_ENV_VEH3DOFCONTI) ? 1 : 0;
}

__host__ __device__ inline int split_pack_blocks(const RolloutParams& p) {
    if (p.ss || p.ssb) {   // streamed-split forward: one block per n-tile of every hidden layer of the policy (and the tail value net)
        int nb = 0;
        for (int m = 0; m < (p.tail ? 2 : 1); ++m) {
            const MlpDev& d = m ? p.val : p.pol;
            for (int j = 0; j < d.nl - 1; ++j) {
                if (p.ss) nb += d.dims[j + 1] >> 4;
                if (p.ssb) nb += ((j == 0) ? d.kp[0] : d.dims[j]) >> 4;   // transposed planes of the sweep
            }
        }
        return nb;
    }
    if (!p.sp.on) return 0;
    return (p.pol.dims[1] >> 4) + (p.pol.dims[2] >> 4) + (p.pol.dims[1] >> 4) + (p.pol.kp[0] >> 4);
}

// Copies the by-value parameter block into device memory (stream ordered, no host staging): the
// rollout kernels then read it with uniform scalar loads instead of a per-lane scratch copy.
// F16 launches: every block also folds its slice of grad_v into max|g| (atomicMax on the bit pattern - for
// non-negative floats the unsigned order is the numeric order) in p.gscale[0], which the forward prologue
// zeroed.  The sweep and the reduce kernel derive the power-of-two scale from it (f16_grad_scale, common.h).
__global__ __launch_bounds__(256) void upload_params_kernel(const RolloutParams p, RolloutParams* dst) {
    if (blockIdx.x == 0) {
        const unsigned* src = reinterpret_cast<const unsigned*>(&p);
        unsigned* d = reinterpret_cast<unsigned*>(dst);
        for (unsigned i = threadIdx.x; i < sizeof(RolloutParams) / 4; i += blockDim.x) d[i] = src[i];
    }
    if (p.gscale != nullptr && p.grad_v != nullptr) {
        __shared__ float red[256];
        float mx = 0.f;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.B; i += gridDim.x * blockDim.x) mx = fmaxf(mx, fabsf(p.grad_v[i]));
        red[threadIdx.x] = mx;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
            __syncthreads();
        }
        if (threadIdx.x == 0 && red[0] < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(p.gscale), __float_as_uint(red[0]));
    }
}

// Streams that are read ONCE (stash operands of the weight-gradient GEMMs, split-K partials) are loaded with the non-temporal
// policy: MI355X_MICROARCH.md measures LDS-DMA fills 18 % earlier with `nt` and 6.5 - 6.8 instead of 6.4 TB/s chip-wide; here the
// wave-specialised GEMM went from 4.2 to 4.9 TB/s (target: weight-gradient group 140 -> 126 us) - the data does not displace
// the other kernels' working set in L2 / MALL either.  GOPS_DW_NT=0: default policy (A/B).
#ifndef GOPS_DW_NT
#define GOPS_DW_NT 1
#endif
#if GOPS_DW_NT
#define DW_STREAM_LOAD(ptr) __builtin_nontemporal_load(ptr)
#define DW_NT_SUFFIX " nt"
#else
#define DW_STREAM_LOAD(ptr) (*(ptr))
#define DW_NT_SUFFIX ""
#endif

// common.h async_copy16_to_lds with the stream policy above (the weight-gradient GEMMs' stash operands)
__device__ __forceinline__ void dw_async_copy16_to_lds(const float* gsrc, const float* lds_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off" DW_NT_SUFFIX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr(gsrc)), "s"(m0v)
                 : "memory");
}

hipError_t launch_upload_params(const RolloutParams& p, RolloutParams* dst, hipStream_t s) {
    static_assert(sizeof(RolloutParams) % 4 == 0, "parameter block must be dword sized");
    int nb = (p.B + 4095) / 4096;
    nb = nb < 1 ? 1 : (nb > 128 ? 128 : nb);   // >= 16 elements per thread
    hipLaunchKernelGGL(upload_params_kernel, dim3(nb), dim3(256), 0, s, p, dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Reference trajectories of pyth_veh3dofconti (ref_traj_model.py:26-232).  Every operation is
// rounded separately in fp32, in the reference's order (no FMA contraction): the heading is a
// 1 ms finite difference whose cancellation noise is part of the reference's result.
// ---------------------------------------------------------------------------------------------
// Transcendentals are evaluated in double and rounded once, i.e. correctly rounded in fp32: the CPU
// reference (Sleef u10) is correctly rounded for almost every argument, so this minimises the
// number of points where a 1-ulp difference in x(t) is amplified ~1e3x by the finite difference.
#define SINF_CR(x) ((float)sin((double)(x)))
#define COSF_CR(x) ((float)cos((double)(x)))
// One rounding per operation, never fused: HIP's __fmul_rn / __fadd_rn are plain `a * b` / `a + b`, which hipcc's default
// -ffp-contract=fast-honor-pragmas may contract into an fma with a neighbour - whether it did depended on unrelated codegen
// (round 4: building without packed-fp32 instructions moved 4 more of the 48 appended headings of the step fixture by ~1e-3).
// The reference (torch CPU) rounds every product and sum separately.
__device__ __forceinline__ float rn_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float rn_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float rn_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
#define RMUL(a, b) rn_mul((a), (b))
#define RADD(a, b) rn_add((a), (b))
#define RSUB(a, b) rn_sub((a), (b))

// `c` = GopsEnv.ref_c (include/gops_hip.h): the reference's path / speed parameters, folded on the host where the
// reference folds Python scalars.  With the default set every expression below rounds exactly like the constants the
// first version of this file had spelled out (x + 0.0f and 1.0f * x are exact).
__device__ __forceinline__ float ref_arc(const float* __restrict__ c, float t, int u_num) {
    if (u_num == 0) return RADD(RADD(RMUL(c[0], COSF_CR(RADD(RMUL(c[1], t), c[2]))), RMUL(c[3], t)), c[4]);
    if (u_num == 1) return RMUL(c[6], t);
    return 0.f;   // a speed id outside the registered set selects no profile: every mask of the reference's sum is false
}

__device__ __forceinline__ void ref_xy(const float* __restrict__ c, float t, int path, int u_num, float& x, float& y) {
    const float s = ref_arc(c, t, u_num);
    if (path == 0) {
        x = s;
        y = RMUL(c[7], SINF_CR(RADD(RMUL(c[8], t), c[9])));
    } else if (path == 1) {
        x = s;
        if (t <= c[10]) y = c[14];
        else if (t <= c[11]) y = RADD(RMUL(c[16], RSUB(t, c[10])), c[14]);
        else if (t <= c[12]) y = c[15];
        else if (t <= c[13]) y = RADD(RMUL(c[17], RSUB(t, c[12])), c[15]);
        else y = c[14];
    } else if (path == 2) {
        x = s;
        float sm = fmodf(t, c[18]);
        if (sm != 0.f && ((c[18] < 0.f) != (sm < 0.f))) sm += c[18];   // torch.remainder: sign of the divisor
        if (sm <= c[21]) y = RMUL(c[19], sm);
        else if (sm < c[18]) y = RMUL(c[20], RSUB(sm, c[18]));
        else y = 0.f;
    } else {
        const float q = s / c[22];
        x = RMUL(c[22], SINF_CR(q));
        y = RMUL(c[22], RSUB(COSF_CR(q), 1.0f));
    }
}

__device__ __forceinline__ f32x4 ref_point(const float* __restrict__ c, float t, int path, int u_num) {
    float x0, y0, x1, y1;
    ref_xy(c, t, path, u_num, x0, y0);
    ref_xy(c, RADD(t, 0.001f), path, u_num, x1, y1);
    const float phi = (float)atan2((double)RSUB(y1, y0), (double)RSUB(x1, x0));
    const float u = (u_num == 0) ? RADD(RMUL(c[5], SINF_CR(RADD(RMUL(c[1], t), c[2]))), c[3]) : (u_num == 1) ? c[6] : 0.f;
    f32x4 r = {x0, y0, phi, u};
    return r;
}

// The point (x, y, phi, u)(t) for the float ids the batches carry (info["path_num"], info["u_num"]).  Ids outside the registered
// sets behave like the reference's masked sums `sum_i (id == i) * f_i(t)` (ref_traj_model.py:54-84, 138-142): an unknown path
// selects nothing (zeros), an unknown speed profile leaves arc length and speed at zero under a known path.
__device__ __forceinline__ f32x4 ref_point_ids(const float* __restrict__ c, float t, float pn, float un) {
    const int path = (pn == 0.f) ? 0 : (pn == 1.f) ? 1 : (pn == 2.f) ? 2 : (pn == 3.f) ? 3 : -1;
    const int us = (un == 0.f) ? 0 : (un == 1.f) ? 1 : -1;
    if (path < 0) return f32x4{0.f, 0.f, 0.f, 0.f};
    return ref_point(c, t, path, us);
}

// table[b][i] : i <= P copies info["ref_points"][b][i]; i = P + s (s >= 1) is the point the model
// appends at rollout step s-1: evaluated at (t0 + s*dt accumulated in fp32) + P*dt.
__device__ __forceinline__ void ref_table_element(int B, int P, int H, const float* __restrict__ ref_points,
                                                  const float* __restrict__ path_num,
                                                  const float* __restrict__ u_num,
                                                  const float* __restrict__ ref_time, float pdt,
                                                  float* __restrict__ table, int idx, bool yphi_only,
                                                  const float* __restrict__ refc, const float* __restrict__ appended) {
    const int TL = P + 1 + H;
    if (idx >= B * TL) return;
    const int b = idx / TL, i = idx - b * TL;
    f32x4 v;
    if (i <= P) {
        if (yphi_only) {   // pyth_veh2dofconti: info["ref_points"] [B, P+1, 2] = (y, phi)
            const float* rp = ref_points + ((size_t)b * (P + 1) + i) * 2;
            v = f32x4{0.f, rp[0], rp[1], 0.f};
        } else {
            v = reinterpret_cast<const f32x4*>(ref_points)[(size_t)b * (P + 1) + i];
        }
    } else if (appended != nullptr) {   // bit-parity mode: the caller's points [B][H][4]
        v = reinterpret_cast<const f32x4*>(appended)[(size_t)b * H + (i - P - 1)];
    } else {
        // the 24 path constants, the three per-trajectory scalars: loaded together up front (read where they are used, inside the
        // branches of ref_xy, each costs a memory round trip of its own)
        float c[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) c[q] = refc[q];
        const float pn = path_num[b], un = u_num[b];
        float t = ref_time[b];
        for (int s = 0; s < i - P; ++s) t = RADD(t, 0.1f);
        v = ref_point_ids(c, RADD(t, pdt), pn, un);
    }
    reinterpret_cast<f32x4*>(table)[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// Forward prologue, ONE launch: block 0 uploads the parameter block, the next blocks pack the
// hidden-layer weights of the policy (and of the tail value net), the rest fill the reference table.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int pack_blocks(const MlpDev& d) {
    int nb = 0;
    for (int j = 0; j < d.nl - 1; ++j) nb += (d.dims[j + 1] * d.kp[j] + 255) / 256;
    return nb;
}
__host__ __device__ inline int pack_blocks_h(const MlpDev& d, int j) {   // covers wph[j] (N x kp32) and wpth[j]
    return (d.dims[j + 1] * d.kp32[j] + 255) / 256;
}

__global__ __launch_bounds__(256) void prologue_kernel(const RolloutParams p, RolloutParams* dst, int P, float pdt) {
    int b = blockIdx.x;
    if (b == 0) {
        const unsigned* src = reinterpret_cast<const unsigned*>(&p);
        unsigned* d = reinterpret_cast<unsigned*>(dst);
        constexpr unsigned NW = sizeof(RolloutParams) / 4, PER = (NW + 255) / 256;
        unsigned w[PER];   // all loads first, then the stores (a load - store loop: one round trip per 1 KB)
#pragma unroll
        for (unsigned q = 0; q < PER; ++q) {
            const unsigned i = threadIdx.x + 256 * q;
            w[q] = i < NW ? src[i] : 0u;
        }
#pragma unroll
        for (unsigned q = 0; q < PER; ++q) {
            const unsigned i = threadIdx.x + 256 * q;
            if (i < NW) d[i] = w[q];
        }
        if (p.gscale != nullptr && threadIdx.x == 0) p.gscale[0] = p.gscale[1] = p.gscale[3] = 0.f;   // max|grad_v| / max|delta_y| of the coming backward; [3]: the overflow mark of forward + sweep
        return;
    }
    b -= 1;
    for (int m = fp32_pack_first_net(p); m < (p.tail ? 2 : 1); ++m) {
        const MlpDev& d = m ? p.val : p.pol;
        for (int j = 0; j < d.nl - 1; ++j) {
            if (p.f16) {
                const int nb = pack_blocks_h(d, j);
                if (b < nb) {
                    pack_element_h(d.w[j], d.dims[j + 1], d.dims[j], d.kp[j], d.kp32[j], j,
                                   const_cast<_Float16*>(reinterpret_cast<const _Float16*>(d.wph[j])),
                                   const_cast<_Float16*>(reinterpret_cast<const _Float16*>(d.wpth[j])), b * 256 + threadIdx.x);
                    return;
                }
                b -= nb;
                continue;
            }
            const int nb = (d.dims[j + 1] * d.kp[j] + 255) / 256;
            if (b < nb) {
                pack_element(d.w[j], d.dims[j + 1], d.dims[j], d.kp[j],
                             const_cast<float*>(reinterpret_cast<const float*>(d.wp[j])),
                             const_cast<float*>(reinterpret_cast<const float*>(d.wpt[j])), b * 256 + threadIdx.x);
                return;
            }
            b -= nb;
        }
    }
    if (p.ss || p.ssb) {   // plane-split operands of the streamed-split forward kernels (p.ss) and of the streamed-split sweep (p.ssb)
        auto us = [](const bf16x8* q) { return const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(q)); };
        auto hf = [](const f16x8* q) { return const_cast<_Float16*>(reinterpret_cast<const _Float16*>(q)); };
        for (int m = 0; m < (p.tail ? 2 : 1); ++m) {
            const MlpDev& d = m ? p.val : p.pol;
            const SplitNetDev& sn = m ? p.ssv : p.ssp;
            for (int j = 0; j < d.nl - 1; ++j) {
                const int nt = p.ss ? d.dims[j + 1] >> 4 : 0;
                if (b < nt) {   // layer 0: natural input order; deeper layers read the plane image of the previous activation
                    pack_split_tile(d.w[j], d.dims[j + 1], d.dims[j], false, j > 0, sn.kc[j], b, us(sn.w1[j]), hf(sn.r[j]), const_cast<float*>(sn.inv[j]));
                    return;
                }
                b -= nt;
                if (p.ssb) {   // sweep: delta_{j+1} -> delta_j (j = 0: g_x) through W_j, tiles over its inputs, slots over its outputs
                    const SplitNetDev& st = m ? p.ssvt : p.sspt;
                    const int ntt = ((j == 0) ? d.kp[0] : d.dims[j]) >> 4;
                    if (b < ntt) {
                        pack_split_tile(d.w[j], d.dims[j + 1], d.dims[j], true, true, st.kc[j], b, us(st.w1[j]), hf(st.r[j]), const_cast<float*>(st.inv[j]));
                        return;
                    }
                    b -= ntt;
                }
            }
        }
    }
    if (p.sp.on) {   // plane-split operands of the stationary kernels: one block per n-tile
        const MlpDev& d = p.pol;
        const int n1 = d.dims[1] >> 4, n2 = d.dims[2] >> 4, k0 = d.kp[0] >> 4;
        auto us = [](const bf16x8* q) { return const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(q)); };
        auto hf = [](const f16x8* q) { return const_cast<_Float16*>(reinterpret_cast<const _Float16*>(q)); };
        if (b < n1) {            // forward layer 0: natural input order
            pack_split_tile(d.w[0], d.dims[1], d.dims[0], false, false, p.sp.kc[0], b, us(p.sp.w1[0]), hf(p.sp.r[0]), const_cast<float*>(p.sp.inv[0]));
            return;
        }
        b -= n1;
        if (b < n2) {            // forward layer 1: its input is the plane image of H_1
            pack_split_tile(d.w[1], d.dims[2], d.dims[1], false, true, p.sp.kc[1], b, us(p.sp.w1[1]), hf(p.sp.r[1]), const_cast<float*>(p.sp.inv[1]));
            return;
        }
        b -= n2;
        if (b < n1) {            // backward delta_2 -> delta_1 through W_1: tiles over its inputs, slots over its outputs
            pack_split_tile(d.w[1], d.dims[2], d.dims[1], true, true, d.dims[2] >> 5, b, us(p.sp.w1t[1]), hf(p.sp.rt[1]), const_cast<float*>(p.sp.invt[1]));
            return;
        }
        b -= n1;
        if (b < k0) {            // backward delta_1 -> g_x through W_0
            pack_split_tile(d.w[0], d.dims[1], d.dims[0], true, true, d.dims[1] >> 5, b, us(p.sp.w1t[0]), hf(p.sp.rt[0]), const_cast<float*>(p.sp.invt[0]));
            return;
        }
        b -= k0;
    }
    if (env_has_ref_table(p.env.kind)) {
        const int nrt = (p.B * (P + 1 + p.H) + 255) / 256;
        if (b < nrt) {
            ref_table_element(p.B, P, p.H, p.in.ref_points, p.in.path_num, p.in.u_num, p.in.ref_time, pdt,
                              const_cast<float*>(p.ref_table), b * 256 + threadIdx.x, p.env.kind == GOPS_ENV_VEH2DOF, p.env.ref_c,
                              p.in.ref_appended);
            return;
        }
        b -= nrt;
    }
    if (p.env.kind == GOPS_ENV_VEH3DOF_SURR) {
        // surrounding vehicles move independently of the policy: their (x, y, phi, u) after t = 0 .. H steps, one
        // thread per (trajectory, vehicle)
        const int ns = p.env.n_surr, idx = b * 256 + threadIdx.x;
        if (idx < p.B * ns) {
            const int bb = idx / ns, i = idx - bb * ns;
            const float* s5 = p.in.surr_state + ((size_t)bb * ns + i) * 5;
            f32x4 cur = {s5[0], s5[1], s5[2], s5[3]};
            const float delta = s5[4];
            f32x4* tab = const_cast<f32x4*>(p.surr_table) + (size_t)bb * (p.H + 1) * ns + i;
            for (int t = 0; t <= p.H; ++t) {
                tab[(size_t)t * ns] = cur;
                cur = surr_next(cur, delta);
            }
        }
    }
}

hipError_t launch_prologue(const RolloutParams& p, RolloutParams* dst, int P, float pdt, hipStream_t s) {
    int nb = 1;
    for (int m = fp32_pack_first_net(p); m < (p.tail ? 2 : 1); ++m) {
        const MlpDev& d = m ? p.val : p.pol;
        if (!p.f16) { nb += pack_blocks(d); continue; }
        for (int j = 0; j < d.nl - 1; ++j) nb += pack_blocks_h(d, j);
    }
    nb += split_pack_blocks(p);
    if (env_has_ref_table(p.env.kind)) nb += (p.B * (P + 1 + p.H) + 255) / 256;
    if (p.env.kind == GOPS_ENV_VEH3DOF_SURR) nb += (p.B * p.env.n_surr + 255) / 256;
    hipLaunchKernelGGL(prologue_kernel, dim3(nb), dim3(256), 0, s, p, dst, P, pdt);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// dW GEMM:  part[split][n][k] = sum_{s in split} D[s][n] * X[s][k]      (n < N, k < Kp)
//
// Both operands are feature-major stash tensors (common.h StashDev: element (tile q, feature f, row m) at
// (q * F + f) * 16 + m), and the contraction index IS the sample: an MFMA operand fragment - consecutive
// samples of one feature - is therefore contiguous in memory and is loaded straight from global into the
// registers of the lane that feeds it to the matrix core.  No LDS, no transposes, no barriers.
//
// Workgroup tile 32R x 32R outputs, 2 x 2 waves of 16R x 16R (R x R MFMA tiles).  A K-block is 32 samples = two
// sample tiles q0, q0 + 1; lane (f = lane & 15, g = lane >> 4) of a fragment holds samples 4g .. 4g+3 of both
// tiles: two dwordx4 loads, each of which covers a fully contiguous KiB per wave (16 features x 64 B).  (The order
// of the 32 samples inside the contraction is free as long as both operands use the same one.)  The next
// block's fragments are in flight while the current ones are multiplied.
//
// BF3: "bf16x3" products on v_mfma_f32_16x16x32_bf16 with fp32 results - every fp32 operand a is split EXACTLY
// into three bf16 planes a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 3 x 8
// significant bits cover the 24 of fp32) and a*b is accumulated in fp32 from the six plane products a1b1, a1b2,
// a2b1, a1b3, a2b2, a3b1; the three dropped ones are below 2^-25 |ab|, i.e. under fp32 rounding.  Six of these
// MFMAs cost 96 cycles against 256 for the eight v_mfma_f32_16x16x4_f32 of the same 32-deep block.
// !BF3: the fp32 matrix-core instruction itself (A/B knob GOPS_DW_F32, and the reference for the split).
//
// Workgroups whose k-tile is 0 also accumulate the bias gradient (column sums of D).
// ---------------------------------------------------------------------------------------------
// (split3: common.h)

// 8 fp32 samples of one feature -> the three bf16x8 plane fragments
__device__ __forceinline__ void split_frag(const f32x4& lo, const f32x4& hi, bf16x8 (&pl)[3]) {
#ifdef GOPS_EXP_NOSPLIT
    const u32x4 v0 = {__float_as_uint(lo[0]), __float_as_uint(lo[1]), __float_as_uint(lo[2]), __float_as_uint(lo[3])};
    const u32x4 v1 = {__float_as_uint(hi[0]), __float_as_uint(hi[1]), __float_as_uint(hi[2]), __float_as_uint(hi[3])};
    pl[0] = __builtin_bit_cast(bf16x8, v0); pl[1] = __builtin_bit_cast(bf16x8, v1); pl[2] = __builtin_bit_cast(bf16x8, v0 ^ v1);
    return;
#endif
    unsigned p0[3], p1[3], p2[3], p3[3];
    split3(lo[0], lo[1], p0);
    split3(lo[2], lo[3], p1);
    split3(hi[0], hi[1], p2);
    split3(hi[2], hi[3], p3);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const u32x4 v = {p0[q], p1[q], p2[q], p3[q]};
        pl[q] = __builtin_bit_cast(bf16x8, v);
    }
}

template <int R, bool BF3>
__global__ __launch_bounds__(NTHREADS, 2) void dw_gemm_fm_kernel(const float* __restrict__ D, int N,
                                                                  const float* __restrict__ X, int Kp,
                                                                  long long Q, int splits, int chunks_per_split,
                                                                  float* __restrict__ part,
                                                                  float* __restrict__ part_b) {
    constexpr int T = 32 * R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_k = (Kp + T - 1) / T, tiles = tiles_k * ((N + T - 1) / T);
    // XCD-aware order: workgroup b runs on XCD b % 8.  All output tiles of one sample split read the
    // same D / X blocks, so they get the same XCD and adjacent slots: the second reader of a block
    // hits that XCD's L2 instead of HBM.
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = local % tiles, split = (local / tiles) * 8 + xcd;
    if (split >= splits) return;
    const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
    const int wn = wave >> 1, wk = wave & 1;
    const int nb = tile_n * T + wn * 16 * R, kb = tile_k * T + wk * 16 * R;   // first feature of this wave's rows / columns
    const int f = lane & 15, g = lane >> 4;
    const bool want_bias = part_b != nullptr && tile_k == 0 && wk == 0;

    // Per-fragment element offsets inside a sample tile.  Features past the matrix edge are CLAMPED to the last one
    // (their products land in accumulator rows / columns that are never stored) and the block index of a prefetch is
    // clamped to the split's last block, so that every load is unconditional: the loop body is one basic block that
    // the scheduler can interleave freely (loads, bf16 splits and MFMAs of different fragments).
    int doff[R], xoff[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        doff[i] = min(nb + 16 * i + f, N - 1) * 16 + 4 * g;
        xoff[i] = min(kb + 16 * i + f, Kp - 1) * 16 + 4 * g;
    }
    const GLOBAL_AS float* Dg = gptr(D);
    const GLOBAL_AS float* Xg = gptr(X);
    const size_t dtile = (size_t)N * 16, xtile = (size_t)Kp * 16;
    // Split s owns the 32-sample blocks s, s + splits, s + 2 splits, ...: at any moment the resident workgroups read
    // one contiguous window of the stash (like a streaming kernel).  Contiguous per-split ranges would walk `splits`
    // streams in lockstep at a fixed large stride, which lands them on the same few HBM channels.
    const long long nblk_all = (Q + 1) >> 1;
    const int nblk = (int)((nblk_all - split + splits - 1) / splits);         // >= 1: splits <= nblk_all
    const bool odd_tail = (Q & 1) && ((nblk_all - 1) % splits == split);      // the very last block has one sample tile only
    auto blk_of = [&](int c) { return (long long)split + (long long)min(c, nblk - 1) * splits; };

    f32x4 acc[R][R] = {};
    float bsum[R];
#pragma unroll
    for (int i = 0; i < R; ++i) bsum[i] = 0.f;
    // Raw fragments: ONE register set.  A fragment's registers are refilled with the same fragment of the NEXT block
    // right after this block's copy has been split / consumed, so every load has a whole block of work to land
    // and no second buffer adds to the register pressure (spilling here would put scratch traffic - and its
    // in-order vmcnt waits - in front of the prefetches).
    f32x4 dra[R][2], xra[R][2];
    auto load_d = [&](long long blk, int i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) dra[i][h] = ld4(Dg + (size_t)min(2 * blk + h, Q - 1) * dtile + doff[i]);
    };
    auto load_x = [&](long long blk, int j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) xra[j][h] = ld4(Xg + (size_t)min(2 * blk + h, Q - 1) * xtile + xoff[j]);
    };
    // one 32-sample block; LAST_HALF_EMPTY: the second sample tile does not exist (odd tile count): its (clamped,
    // i.e. duplicate) fragments are replaced by zeros
    auto block = [&]<bool LAST_HALF_EMPTY>(long long next_blk) {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF3) {
            bf16x8 b[R][3];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                split_frag(xra[j][0], LAST_HALF_EMPTY ? zero4 : xra[j][1], b[j]);
                load_x(next_blk, j);
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const f32x4 d0 = dra[i][0], d1 = LAST_HALF_EMPTY ? zero4 : dra[i][1];
                bf16x8 a[3];
                split_frag(d0, d1, a);
                bsum[i] += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));   // (every wave: no branch in the block)
                load_d(next_blk, i);
                // six plane products, smallest terms first; consecutive MFMAs go to different accumulators
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int j = 0; j < R; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
            }
        } else {
            f32x4 xc[R][2];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                xc[j][0] = xra[j][0]; xc[j][1] = LAST_HALF_EMPTY ? zero4 : xra[j][1];
                load_x(next_blk, j);
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const f32x4 d0 = dra[i][0], d1 = LAST_HALF_EMPTY ? zero4 : dra[i][1];
                bsum[i] += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));   // (every wave: no branch in the block)
                load_d(next_blk, i);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < R; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u < 4 ? d0[u & 3] : d1[u & 3], xc[j][u >> 2][u & 3], acc[i][j], 0, 0, 0);
            }
        }
    };

#pragma unroll
    for (int j = 0; j < R; ++j) load_x(blk_of(0), j);
#pragma unroll
    for (int i = 0; i < R; ++i) load_d(blk_of(0), i);
    const int nfull = odd_tail ? nblk - 1 : nblk;
    for (int c = 0; c < nfull; ++c) block.template operator()<false>(blk_of(c + 1));
    if (odd_tail) block.template operator()<true>(blk_of(nblk));
    float* pbase = part + (size_t)split * N * Kp;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int k = kb + 16 * j + f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + 16 * i + 4 * g + r;
                if (n < N && k < Kp) pbase[(size_t)n * Kp + k] = acc[i][j][r];
            }
        }
    if (want_bias) {   // column sums of D: the four sample groups of a feature sit in lanes f, f+16, f+32, f+48
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float t = bsum[i];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (g == 0 && nb + 16 * i + f < N) part_b[(size_t)split * N + nb + 16 * i + f] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same GEMM for the large layers (N % 128 == 0, Kp % 128 == 0): the operand tiles of a
// 32-sample block (128 features x 2 sample tiles of each operand = 4 x 8 KiB of CONTIGUOUS stash memory) travel
// HBM -> LDS with global_load_lds_dwordx4 - no register holds them - into a ring of DWR_STAGES stages, DWR_STAGES - 1
// blocks ahead of the block being multiplied; wave w copies region w of a stage (eight 1-KiB copies).  Two stages and
// two workgroups per CU measured best (layer-2 GEMM of the target: 103 us; four stages with one workgroup per CU 141 us:
// with a single wave per SIMD nothing overlaps the bf16 split on the VALU with the MFMAs).
// Every element is fetched from L2 / HBM ONCE per workgroup (the register-direct kernel above fetches it once per
// wave that needs it).  A lane's fragment (feature f = lane & 15, samples 4 g .. 4 g + 3 of both sample tiles) is two
// conflict-free ds_read_b128 of the LDS image; it is split into the three bf16 planes in registers and multiplied.
// One barrier per block: it publishes block c (each wave first waits for its own copies of that block) and frees the
// stage that block c - 1 was read from for the copies of block c + DWR_STAGES - 1.
// ---------------------------------------------------------------------------------------------
// Two-half-plane form of an fp32 operand fragment for the H2 variant of the ring GEMM: with v = x * s,
//     hi = f16(v)  (round toward zero: a finite v never becomes inf),   lo = f16((v - hi) * 2^11)  (likewise),
// so v = hi + lo * 2^-11 to 2^-21 |v| (22 significant bits), and the product of two such operands is
//     d * a = dh * ah + (dh * al + dl * ah) * 2^-11 + O(2^-21 |d a|)
// - three v_mfma_f32_16x16x32_f16 per 32-sample block instead of the six bf16 MFMAs of the exact three-plane split, and 3-4
// VALU instructions per element instead of 5.5.  Mode 2 (adopted) keeps lo UNSCALED (lo = f16(v - hi), one accumulator):
// lo is a normal half for |v| >= 0.06 and a subnormal one below (absolute error <= 6e-8 in scaled units), so the scales
// put typical magnitudes near 1..30: activations as they are, deltas times the power of two that brings max|grad_v| into
// [16, 32).  Range: |x * s| >= 65504 saturates (round toward zero never produces inf); the converting threads watch for it
// and a workgroup whose operands saturated repeats ITS blocks with the exact three-plane split before it stores anything -
// a diverged rollout costs time, never a wrong gradient.
#ifndef GOPS_DW_H2_MODE
#define GOPS_DW_H2_MODE 2   // 1: lo planes scaled by 2^11, two accumulators, one workgroup per CU (r03: 251 us at the target, slower than the
                            // exact split); 2: unscaled lo planes, ONE accumulator, two workgroups per CU (177 us; the matrix core takes
                            // subnormal half inputs as they are - measured: same 2e-7 distance to the exact GEMM as mode 1)
#endif
#if GOPS_DW_H2_MODE == 2
#define DW_H2_SA 1.0f      // (unscaled lo planes: keep typical magnitudes near 1 so that lo stays a normal half)
#define DW_H2_LO 1.0f
#else
#define DW_H2_SA 0.0625f   // activations / observations: up to 1.05e6 before the main term saturates
#define DW_H2_LO 2048.f
#endif
__device__ __forceinline__ void split2h(const f32x4& lo4, const f32x4& hi4, float s, f16x8& ph, f16x8& pl) {
    const float s2 = s * DW_H2_LO;
    u32x4 uh, ul;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = e < 2 ? lo4[2 * e] : hi4[2 * e - 4], x1 = e < 2 ? lo4[2 * e + 1] : hi4[2 * e - 3];
        const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0 * s, x1 * s));
        const float r0 = fmaf((float)h[0], -DW_H2_LO, x0 * s2), r1 = fmaf((float)h[1], -DW_H2_LO, x1 * s2);   // (v - hi) * 2^11, exact
        uh[e] = __builtin_bit_cast(unsigned, h);
        ul[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
    ph = __builtin_bit_cast(f16x8, uh);
    pl = __builtin_bit_cast(f16x8, ul);
}

// The same split for the wave-specialised kernel's converting waves, spelled in the instructions it should cost: v_fma_mixlo / mixhi_f16
// form f16(s x) and f16(s x - hi) in ONE instruction each (fp32 product-sum, one rounding to half, written to one half of the
// destination) - 4 VALU per element pair instead of the 6 hipcc emits for the expression above (v_mul x 2, v_cvt_pkrtz, v_fma_mix x 2,
// v_cvt_pkrtz).  Rounding is to nearest here (toward zero there): lo then needs one bit less, the pair carries the same 22 bits;
// |s x| >= 65520 becomes inf instead of saturating - the callers' range watch (vmax < 65504) is on the fp32 values and unchanged.
#if GOPS_DW_H2_MODE == 2
__device__ __forceinline__ void split2h_mix(const f32x4& lo4, const f32x4& hi4, float s, f16x8& ph, f16x8& pl) {
    u32x4 uh, ul;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = e < 2 ? lo4[2 * e] : hi4[2 * e - 4], x1 = e < 2 ? lo4[2 * e + 1] : hi4[2 * e - 3];
        unsigned h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(s), "v"(x0));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(s), "v"(x1));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(s), "v"(x0), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(s), "v"(x1), "v"(h));
        uh[e] = h;
        ul[e] = l;
    }
    ph = __builtin_bit_cast(f16x8, uh);
    pl = __builtin_bit_cast(f16x8, ul);
}
#else
#define split2h_mix split2h
#endif
// running maximum of |a|, |b| in one instruction (hipcc spells fmaxf(m, fmaxf(fabsf(a), fabsf(b))) as two canonicalising v_max + v_max3)
__device__ __forceinline__ void absmax2(float& m, float a, float b) { asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a), "v"(b)); }

// What the hidden deltas of a backward call are measured against when they are scaled for the half planes (dscale =
// RolloutParams::gscale): the largest |delta_y| of the sweep (slot 1: the plane-split sweeps track it, rollout_bwd.hip) where there
// is one, else max|grad_v| (slot 0).  Whatever the choice, a block that still saturates is redone exactly.
__device__ __forceinline__ float dw_delta_yardstick(const float* dscale) {
    const float dy = gptr(dscale)[1], gv = gptr(dscale)[0];
    return dy > 0.f ? dy : gv;
}

#define DWR_STAGES 2
#define DWR_STAGE_FLOATS (4 * 2048)   // [D tile q0][D tile q0+1][X tile q0][X tile q0+1], each [128 features][16 rows]

// H2: the two-half-plane products (split2h) instead of the exact three-plane bf16 split; dscale -> max|grad_v| of the launch
// (RolloutParams::gscale): the deltas are multiplied by f16_grad_scale(max|grad_v|) / 64 before they are halved.  The
// staged fp32 tiles of a block are converted IN PLACE, once per workgroup: work item (operand, feature f, sample group g)
// reads the two 16-byte vectors that are lane (f, g)'s fragment (rows 4g..4g+3 of both sample tiles), forms the hi / lo
// half planes of those 8 samples and writes hi where the first tile's vector was, lo where the second's was - 1024 items
// per block, 4 per thread, the same items every block (so the bias column sums accumulate in the converting thread).  The
// MFMA phase then reads ready-made operands (two ds_read_b128 per fragment) and issues no VALU work at all.
template <bool H2>
__global__ __launch_bounds__(NTHREADS, (H2 && GOPS_DW_H2_MODE == 1) ? 1 : 2) void dw_gemm_ring_kernel(const float* __restrict__ D, int N,
                                                                    const float* __restrict__ X, int Kp,
                                                                    long long Q, int splits, int chunks_per_split,
                                                                    float* __restrict__ part,
                                                                    float* __restrict__ part_b, const float* __restrict__ dscale, int redo_guard) {
    extern __shared__ __attribute__((aligned(16))) float ring[];   // [DWR_STAGES][DWR_STAGE_FLOATS]
    constexpr int T = 128, R = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_k = (Kp + T - 1) / T, tiles = tiles_k * (N / T);   // (the last K-tile may be partial: Kp is a multiple of 16)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;   // XCD-aware order, as above
    const int tile = local % tiles, split = (local / tiles) * 8 + xcd;
    if (split >= splits) return;
    const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
    const int wn = wave >> 1, wk = wave & 1;
    const int f = lane & 15, g = lane >> 4;
    const bool want_bias = part_b != nullptr && tile_k == 0 && wk == 0;
    const long long nblk_all = (Q + 1) >> 1;                                  // block c of this split is block split + c * splits
    const int nblk = (int)((nblk_all - split + splits - 1) / splits);
    const bool odd_tail = (Q & 1) && ((nblk_all - 1) % splits == split);

    // this wave's copy job: region `wave` of a stage = sample tile (wave & 1) of operand (wave >> 1)
    const float* csrc = (wave < 2) ? D + (size_t)tile_n * T * 16 : X + (size_t)tile_k * T * 16;
    const size_t ctile = (size_t)((wave < 2) ? N : Kp) * 16;
    // features of a partial last K-tile that do not exist (>= Kp) are fetched from the tile's first features instead: finite
    // values whose products land in columns that are never stored
    const int kvalid = (wave < 2) ? T : min(T, Kp - tile_k * T);   // valid features of this wave's operand tile
    auto copy_block = [&](int c, int stage) {   // block index clamped: the copy count per block is constant
        const long long bq = min(((long long)split + (long long)min(c, nblk - 1) * splits) * 2 + (wave & 1), Q - 1);
        const float* src = csrc + (size_t)bq * ctile + 4 * lane;
        const float* dst = ring + stage * DWR_STAGE_FLOATS + wave * 2048;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int feat = u * 16 + (lane >> 2);
            dw_async_copy16_to_lds(feat < kvalid ? src + u * 256 : src + u * 256 - (size_t)(feat - (feat % kvalid)) * 16, dst + u * 256);
        }
    };

    f32x4 acc[R][R] = {};
    constexpr bool TWO_ACC = H2 && GOPS_DW_H2_MODE == 1;
    f32x4 accx[TWO_ACC ? R : 1][TWO_ACC ? R : 1] = {};   // cross terms (dh * al + dl * ah), scaled by 2^11
    float bsum[R] = {0.f, 0.f, 0.f, 0.f};
    float sd = 1.f;
    if constexpr (H2) sd = f16_grad_scale(dw_delta_yardstick(dscale)) * (GOPS_DW_H2_MODE == 2 ? 16.f : 0.015625f);
    auto block = [&]<bool LAST_HALF_EMPTY, bool M2>(int stage) {
        const float* st = ring + stage * DWR_STAGE_FLOATS;
        const float* da = st + (wn * 64 + f) * 16 + 4 * g;          // D fragments of row-tile i: + 256 i  (+ 2048: second tile)
        const float* xa = st + 4096 + (wk * 64 + f) * 16 + 4 * g;   // X fragments of column-tile j
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (M2) {   // operands were converted in place by convert_stage(): hi plane in the first tile's slot, lo in the second's
            f16x8 bh[R], bl[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(xa + 256 * j);
                bl[j] = *reinterpret_cast<const f16x8*>(xa + 256 * j + 2048);
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(da + 256 * i), al = *reinterpret_cast<const f16x8*>(da + 256 * i + 2048);
                if constexpr (TWO_ACC) {
#pragma unroll
                    for (int j = 0; j < R; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], accx[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < R; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], accx[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < R; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < R; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < R; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc[i][j], 0, 0, 0);
            }
            return;
        }
#ifdef GOPS_EXP_NOMFMA
        acc[0][0] += *reinterpret_cast<const f32x4*>(xa) + *reinterpret_cast<const f32x4*>(da);
        return;
#endif
        bf16x8 b[R][3];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xa + 256 * j);
            const f32x4 x1 = LAST_HALF_EMPTY ? zero4 : *reinterpret_cast<const f32x4*>(xa + 256 * j + 2048);
            split_frag(x0, x1, b[j]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(da + 256 * i);
            const f32x4 d1 = LAST_HALF_EMPTY ? zero4 : *reinterpret_cast<const f32x4*>(da + 256 * i + 2048);
            bf16x8 a[3];
            split_frag(d0, d1, a);
            bsum[i] += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < R; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        }
    };

    // H2: in-place conversion of a landed stage; item id = tid + 256 k: operand id >> 9, feature (id >> 2) & 127, group id & 3
    float vmax = 0.f;             // largest scaled magnitude this thread converted (saturation watch)
    float csum[2] = {0.f, 0.f};   // column sums of D over the two D items of this thread (features tid >> 2 and 64 + (tid >> 2), group tid & 3)
    auto convert_stage = [&]<bool LAST_HALF_EMPTY>(int stage) {
        float* st = ring + stage * DWR_STAGE_FLOATS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = tid + NTHREADS * k, op = id >> 9, fe = (id >> 2) & 127, gg = id & 3;
            float* at = st + op * 4096 + fe * 16 + 4 * gg;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(at);
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v1 = LAST_HALF_EMPTY ? zero4 : *reinterpret_cast<const f32x4*>(at + 2048);
            if (k < 2) csum[k] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
            const float sc = k < 2 ? sd : DW_H2_SA;
            const float m0 = fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3])));
            const float m1 = fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3])));
            vmax = fmaxf(vmax, fmaxf(m0, m1) * sc);   // (NaN-free inputs: a NaN in the stash is NaN in the result either way)
            f16x8 ph, pl;
            split2h(v0, v1, sc, ph, pl);
            *reinterpret_cast<f16x8*>(at) = ph;
            *reinterpret_cast<f16x8*>(at + 2048) = pl;
        }
    };
    const int nfull = odd_tail ? nblk - 1 : nblk;
    // one pass over this workgroup's blocks: M2 = two-half-plane products (with the in-place conversion), else the exact split
    auto run = [&]<bool M2>() {
#pragma unroll
        for (int d = 0; d < DWR_STAGES - 1; ++d) copy_block(d, d);
        for (int c = 0; c < nblk; ++c) {
            // this wave's copies of block c have landed once at most the 8 x (DWR_STAGES - 2) younger ones are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (DWR_STAGES - 2)) : "memory");
            __syncthreads();
            copy_block(c + DWR_STAGES - 1, (c + DWR_STAGES - 1) % DWR_STAGES);
            if constexpr (M2) {
                if (c < nfull) convert_stage.template operator()<false>(c % DWR_STAGES);
                else convert_stage.template operator()<true>(c % DWR_STAGES);
                __syncthreads();
            }
            if (c < nfull) block.template operator()<false, M2>(c % DWR_STAGES);
            else block.template operator()<true, M2>(c % DWR_STAGES);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail copies must not outlive the workgroup's LDS / the pass
    };
    bool redone = false;
    if constexpr (H2) {
        if (tid == 0) ring[DWR_STAGES * DWR_STAGE_FLOATS] = 0.f;
        run.template operator()<true>();
        // A half plane of THIS workgroup's operands saturated (|x * s| >= 65504: a diverged rollout): its products are wrong,
        // so the workgroup repeats its blocks with the exact three-plane split - time, never a wrong gradient.  (The
        // column sums were formed from the fp32 values and stand.)
        // (the flag word sits behind the ring in DYNAMIC LDS: __syncthreads_or would add static LDS to a 64-KiB dynamic
        // allocation, and hipFuncSetAttribute then refuses the 160-KiB dynamic limit the launch needs)
        unsigned* sat = reinterpret_cast<unsigned*>(ring + DWR_STAGES * DWR_STAGE_FLOATS);
        if (!(vmax < 65504.f)) *sat = 1u;   // (zeroed by thread 0 before the first barrier of the pass above)
        __syncthreads();
        if (*sat != 0u && redo_guard) {
            redone = true;
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            run.template operator()<false>();
        }
    } else {
        run.template operator()<false>();
    }

    const int nb = tile_n * T + wn * 64, kb = tile_k * T + wk * 64;
    float* pbase = part + (size_t)split * N * Kp;
    const float unscale = (H2 && !redone) ? 1.f / (sd * DW_H2_SA) : 1.f;   // (powers of two: exact)
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int k = kb + 16 * j + f;
            if (k >= Kp) continue;   // (partial last K-tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if constexpr (TWO_ACC) v = fmaf(accx[i][j][r], 1.f / 2048.f, v);
                if constexpr (H2) v *= unscale;
                pbase[(size_t)(nb + 16 * i + 4 * g + r) * Kp + k] = v;
            }
        }
    if constexpr (H2) {   // the converting threads hold the column sums: groups gg = tid & 3 of a feature sit in 4 adjacent lanes
        if (part_b != nullptr && tile_k == 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float t = csum[k];
                t += __shfl_xor(t, 1);
                t += __shfl_xor(t, 2);
                if ((tid & 3) == 0) part_b[(size_t)split * N + tile_n * T + 64 * k + (tid >> 2)] = t;
            }
        }
    } else
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float t = bsum[i];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (g == 0) part_b[(size_t)split * N + nb + 16 * i + f] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised ring GEMM (two-half-plane products) for the layers with N % 256 == 0.  Measured on the 4-wave ring
// kernel at the target (two GEMMs per update, 165 us): without the conversion 128 us, without the MFMAs 123 us - stash
// stream (~85 us at 6 TB/s), conversion VALU and MFMAs ADD, because all waves of a workgroup convert, then all multiply, and
// the co-resident workgroup falls into the same rhythm.  Here the two kinds of work run side by side on every SIMD:
//   * 512 threads, one workgroup per CU, 256 x 128 outputs (every delta element is fetched and converted by ONE workgroup
//     per K-tile), three 48-KiB stages [D q0][D q0+1][X q0][X q0+1] of 32 samples;
//   * waves 0-3 CONVERT block c + 1 in place (fp32 -> hi / lo half planes, 6 items per thread) while waves 4-7 MULTIPLY
//     block c (128 x 64 outputs per wave, 96 MFMAs); all eight waves issue the LDS-DMA copies of block c + 2 (6 KiB each);
//   * ONE barrier per block: converted block c + 1 is published, block c's stage is free for the copies of block c + 3,
//     and each wave has waited for its own copies of block c + 2.
// Saturated half planes (a diverged rollout): the workgroup repeats its blocks with the exact three-plane split on the same
// four multiplying waves, from unconverted stages.
// ---------------------------------------------------------------------------------------------
#ifndef DW_SPEC_REGDIRECT
#define DW_SPEC_REGDIRECT 1
#endif
#ifndef DW_SPEC_RAW3
#define DW_SPEC_RAW3 0   // three (1) or two (0) blocks of raw fragments in the converting waves' registers
#endif
struct DwSpec {
    static constexpr int NW = 8, NT = 512, TN = 256, T = 128;
    static constexpr int DT = TN * 16;                       // floats of one D sample tile
    static constexpr int STAGES = 3;
    static constexpr int STAGE_FLOATS = 2 * DT + 4096;       // 48 KiB
    static constexpr int PIECES = STAGE_FLOATS / 256 / NW;   // 1-KiB copies per wave and block
    static constexpr size_t lds_bytes() { return (size_t)STAGES * STAGE_FLOATS * sizeof(float) + 16; }
};

struct DwSpecGeo {
    const float* dbase;   // D + tile_n * 256 * 16
    const float* xbase;   // X + tile_k * 128 * 16
    float* pbase;         // part + split * N * Kp
    long long Q;
    int N, Kp, split, splits, nblk, nfull, xgroups, tile_n, tile_k;
};
// copy job of a wave: pieces wave * PIECES .. + PIECES - 1 of a stage; a piece = 16 features x 16 rows (1 KiB, contiguous in the
// stash) of one sample tile of one operand, in the stage's order.  Feature groups of a partial last K-tile that do not exist
// (>= Kp) are fetched from the tile's first groups instead: finite values whose columns are never stored.
__device__ __forceinline__ void dw_spec_copy(const DwSpecGeo& G, float* ring, int c, int stage, int wave, int lane) {
    constexpr int T = DwSpec::T, TN = DwSpec::TN, PC = DwSpec::PIECES;
    const long long b0 = ((long long)G.split + (long long)c * G.splits) * 2;
    float* dst = ring + stage * DwSpec::STAGE_FLOATS + wave * PC * 256;
#pragma unroll
    for (int u = 0; u < PC; ++u) {
        const int pc = wave * PC + u;   // (wave-uniform)
        const bool isx = pc >= 2 * (TN / 16);
        const int q = isx ? pc - 2 * (TN / 16) : pc, per = isx ? T / 16 : TN / 16;
        const int st_tile = q / per;
        int grp = q - st_tile * per;
        if (isx && grp >= G.xgroups) grp %= G.xgroups;
        const size_t bq = (size_t)min(b0 + st_tile, G.Q - 1);
        const float* src = (isx ? G.xbase + bq * ((size_t)G.Kp * 16) : G.dbase + bq * ((size_t)G.N * 16)) + grp * 256 + 4 * lane;
        dw_async_copy16_to_lds(src, dst + u * 256);
    }
}
// The same copies with their address arithmetic taken out of the loop.  A piece is wave-uniform (only the 16 bytes per lane inside
// it differ), its source moves by a constant number of bytes from block to block: the base lives in an SGPR pair that one
// s_add_u32 / s_addc_u32 advances, the lane's offset in ONE VGPR, and the copy is `global_load_lds_dwordx4 voff, s[base]`.
// dw_spec_copy above recomputes every piece's 64-bit address with ~25 VALU / SALU instructions - 150 per wave and block in all
// eight waves, next to the 96 MFMAs of a multiplying wave and the ~150 VALU of a converting one, and VALU and MFMA issue time ADD
// on a SIMD.  Not for the last block of a split whose second sample tile does not exist (clamped there: dw_spec_copy).
struct DwSpecSrc {
    unsigned long long p[DwSpec::PIECES];   // source of this wave's piece u in the NEXT block to copy (wave-uniform, without the lane offset)
    unsigned long long stride_d, stride_x;  // bytes from block c to block c + 1 of a split
};
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v) |   // (the builtin returns int: no sign extension)
           ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32);
}
__device__ __forceinline__ void dw_spec_src_init(const DwSpecGeo& G, int wave, DwSpecSrc& S) {
    constexpr int T = DwSpec::T, TN = DwSpec::TN, PC = DwSpec::PIECES;
    const long long b0 = (long long)G.split * 2;
#pragma unroll
    for (int u = 0; u < PC; ++u) {
        const int pc = wave * PC + u;
        const bool isx = pc >= 2 * (TN / 16);
        const int q = isx ? pc - 2 * (TN / 16) : pc, per = isx ? T / 16 : TN / 16;
        const int st_tile = q / per;
        int grp = q - st_tile * per;
        if (isx && grp >= G.xgroups) grp %= G.xgroups;
        const size_t bq = (size_t)(b0 + st_tile);
        const float* src = (isx ? G.xbase + bq * ((size_t)G.Kp * 16) : G.dbase + bq * ((size_t)G.N * 16)) + grp * 256;
        S.p[u] = uniform64((unsigned long long)(size_t)src);
    }
    S.stride_d = uniform64((unsigned long long)G.splits * 2ull * (unsigned long long)G.N * 64ull);
    S.stride_x = uniform64((unsigned long long)G.splits * 2ull * (unsigned long long)G.Kp * 64ull);
}
__device__ __forceinline__ void dw_spec_copy_next(DwSpecSrc& S, float* ring, int stage, int wave, int lane) {
    constexpr int TN = DwSpec::TN, PC = DwSpec::PIECES;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(const __attribute__((address_space(3))) void*)(ring + stage * DwSpec::STAGE_FLOATS + wave * PC * 256));
    const unsigned voff = 16u * (unsigned)lane;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
    for (int u = 0; u < PC; ++u) {
        const unsigned long long src = uniform64(S.p[u]);   // (loop-carried: hipcc may keep it in VGPRs - two v_readfirstlane then)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" DW_NT_SUFFIX : : "v"(voff), "s"(src), "s"(lds0 + 1024u * u) : "memory");
        S.p[u] += (wave * PC + u >= 2 * (TN / 16)) ? S.stride_x : S.stride_d;
    }
    asm volatile("s_mov_b32 m0, %0" : : "s"(keep));
}
__device__ __forceinline__ void dw_spec_skip(DwSpecSrc& S, int wave) {   // a block copied by dw_spec_copy: keep the running sources in step
#pragma unroll
    for (int u = 0; u < DwSpec::PIECES; ++u) S.p[u] += (wave * DwSpec::PIECES + u >= 2 * (DwSpec::TN / 16)) ? S.stride_x : S.stride_d;
}
__device__ __forceinline__ void dw_spec_store(const DwSpecGeo& G, const f32x4 (&acc)[8][4], int wn, int wk, int f, int g, float unscale) {
    const int nb = G.tile_n * DwSpec::TN + wn * 128, kb = G.tile_k * DwSpec::T + wk * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kb + 16 * j + f;
            if (k >= G.Kp) continue;   // (partial last K-tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) G.pbase[(size_t)(nb + 16 * i + 4 * g + r) * G.Kp + k] = acc[i][j][r] * unscale;
        }
}
// Exact pass of a workgroup whose half planes saturated: the same blocks from unconverted stages, three-plane bf16 split on
// the four multiplying waves, results stored here.  A separate (non-inlined) function: its 128 accumulator + 60 operand
// registers would otherwise shape the register allocation of the kernel's main loop.
__device__ __attribute__((noinline)) void dw_spec_exact_pass(const DwSpecGeo G, float* ring) {
    constexpr int DT = DwSpec::DT, SF = DwSpec::STAGE_FLOATS, NST = DwSpec::STAGES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mul_wave = wave >= 4;
    const int mw = wave & 3, wn = mw >> 1, wk = mw & 1, f = lane & 15, g = lane >> 4;
    f32x4 acc[8][4] = {};
    dw_spec_copy(G, ring, 0, 0, wave, lane);
    if (G.nblk > 1) dw_spec_copy(G, ring, 1, 1, wave, lane);
    for (int c = 0; c < G.nblk; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // block c (and c + 1) landed; every wave is past block c - 1
        if (c + 2 < G.nblk) dw_spec_copy(G, ring, c + 2, (c + 2) % NST, wave, lane);
        if (mul_wave) {
            const bool half_empty = c >= G.nfull;
            const float* st = ring + (c % NST) * SF;
            const float* da = st + (wn * 128 + f) * 16 + 4 * g;
            const float* xa = st + 2 * DT + (wk * 64 + f) * 16 + 4 * g;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            bf16x8 b[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xa + 256 * j);
                const f32x4 x1 = half_empty ? zero4 : *reinterpret_cast<const f32x4*>(xa + 256 * j + 2048);
                split_frag(x0, x1, b[j]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(da + 256 * i);
                const f32x4 d1 = half_empty ? zero4 : *reinterpret_cast<const f32x4*>(da + 256 * i + DT);
                bf16x8 a[3];
                split_frag(d0, d1, a);
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (mul_wave) dw_spec_store(G, acc, wn, wk, f, g, 1.f);
}

__global__ __launch_bounds__(512, 1) void dw_gemm_spec_kernel(const float* __restrict__ D, int N, const float* __restrict__ X, int Kp, long long Q,
                                                              int splits, float* __restrict__ part, float* __restrict__ part_b,
                                                              const float* __restrict__ dscale, int guard) {
    extern __shared__ __attribute__((aligned(16))) float ring[];   // [STAGES][STAGE_FLOATS] + flag word
    constexpr int T = DwSpec::T, TN = DwSpec::TN, DT = DwSpec::DT, SF = DwSpec::STAGE_FLOATS, NST = DwSpec::STAGES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mul_wave = wave >= 4;   // (waves w and w + 4 share a SIMD: one converting, one multiplying)
    const int tiles_k = (Kp + T - 1) / T, tiles = tiles_k * (N / TN);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;   // XCD-aware order, as above
    const int tile = local % tiles, split = (local / tiles) * 8 + xcd;
    if (split >= splits) return;
    const int f = lane & 15, g = lane >> 4;
    DwSpecGeo G;
    G.tile_n = tile / tiles_k;
    G.tile_k = tile - G.tile_n * tiles_k;
    G.N = N; G.Kp = Kp; G.Q = Q; G.split = split; G.splits = splits;
    const long long nblk_all = (Q + 1) >> 1;                                  // block c of this split is block split + c * splits
    G.nblk = (int)((nblk_all - split + splits - 1) / splits);
    const bool odd_tail = (Q & 1) && ((nblk_all - 1) % splits == split);
    G.nfull = odd_tail ? G.nblk - 1 : G.nblk;
    G.xgroups = min(T, Kp - G.tile_k * T) >> 4;
    G.dbase = D + (size_t)G.tile_n * TN * 16;
    G.xbase = X + (size_t)G.tile_k * T * 16;
    G.pbase = part + (size_t)split * N * Kp;
    const int nblk = G.nblk, nfull = G.nfull;

#if !DW_SPEC_REGDIRECT
    // multiplying waves: 128 x 64 outputs each
    const int mw = wave & 3, wn = mw >> 1, wk = mw & 1;
    f32x4 acc[8][4] = {};
    const float sd = f16_grad_scale(dw_delta_yardstick(dscale)) * 16.f;
    auto block_h2 = [&](int stage) {
        const float* st = ring + stage * SF;
        const float* da = st + (wn * 128 + f) * 16 + 4 * g;            // D fragments of row-tile i: + 256 i  (lo plane: + DT)
        const float* xa = st + 2 * DT + (wk * 64 + f) * 16 + 4 * g;    // X fragments of column-tile j: + 256 j  (lo plane: + 2048)
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(xa + 256 * j);
            bl[j] = *reinterpret_cast<const f16x8*>(xa + 256 * j + 2048);
        }
        // D fragments double-buffered behind scheduling barriers (left alone, hipcc hoists all sixteen reads: 64 registers, spills)
        f16x8 ah = *reinterpret_cast<const f16x8*>(da), al = *reinterpret_cast<const f16x8*>(da + DT);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f16x8 nh = ah, nl = al;
            if (i + 1 < 8) {
                nh = *reinterpret_cast<const f16x8*>(da + 256 * (i + 1));
                nl = *reinterpret_cast<const f16x8*>(da + 256 * (i + 1) + DT);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ah = nh; al = nl;
        }
    };

#else
    const int mw = wave & 3, wn = mw >> 1, wk = mw & 1;
    const float sd = f16_grad_scale(dw_delta_yardstick(dscale)) * 16.f;
#endif
    // converting waves (threads 0 .. 255): items k < 4: D feature (tid >> 2) + 64 k, k = 4, 5: X feature (tid >> 2) + 64 (k - 4);
    // sample group tid & 3.  The same items every block: the bias column sums accumulate in the converting thread.
    float vmax = 0.f, vmax_d = 0.f, vmax_x = 0.f;   // largest |delta| / |activation| this thread converted (unscaled; folded into vmax at the end)
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_csum = part_b != nullptr && G.tile_k == 0;   // (the bias gradient = column sums of D: once per row of output tiles)
    auto convert_stage = [&]<bool LAST_HALF_EMPTY>(int stage) {
        float* st = ring + stage * SF;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool isd = k < 4;
            const int fe = (tid >> 2) + 64 * (isd ? k : k - 4), gg = tid & 3, second = isd ? DT : 2048;
            float* at = st + (isd ? 0 : 2 * DT) + fe * 16 + 4 * gg;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(at);
            const f32x4 v1 = LAST_HALF_EMPTY ? zero4 : *reinterpret_cast<const f32x4*>(at + second);
            if (isd && want_csum) csum[k & 3] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
            float& vm = isd ? vmax_d : vmax_x;
            absmax2(vm, v0[0], v0[1]); absmax2(vm, v0[2], v0[3]);
            absmax2(vm, v1[0], v1[1]); absmax2(vm, v1[2], v1[3]);
            f16x8 ph, pl;
            split2h_mix(v0, v1, isd ? sd : DW_H2_SA, ph, pl);
            *reinterpret_cast<f16x8*>(at) = ph;
            *reinterpret_cast<f16x8*>(at + second) = pl;
        }
    };
    auto landed_and_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    (void)convert_stage; (void)landed_and_sync;   // (used by the LDS-DMA ring form only: DW_SPEC_REGDIRECT 0)

    unsigned* sat = reinterpret_cast<unsigned*>(ring + NST * SF);
    if (tid == 0) *sat = 0u;
#if DW_SPEC_REGDIRECT
    // Register-direct feed (round 6).  The LDS-DMA ring gave a block's copies ONE iteration to land and kept <= 48 KiB per CU in
    // flight - knock-out builds: without the MFMAs -10 us, without the conversion -1 us, i.e. the copy path set the pace (4.2 TB/s).
    // Here the four converting waves LOAD the fp32 fragments themselves (non-temporal 16-byte loads: a wave's instruction covers
    // 1 KiB of contiguous stash), hold two blocks in registers (2 x 48 of their 256), convert from registers and store only the
    // half planes: block c + 2 is converted while block c is multiplied, blocks c + 3 and c + 4 are in flight - two full
    // iterations of latency cover, ~96 KiB per CU in flight, and 144 instead of 240 KiB of LDS traffic per block.
    struct Raw { f32x4 v[6][2]; };
    Raw r0, r1;
    const int gg = tid & 3;
    long long lofs[6];   // float offset of item k inside a sample tile of its operand (+ operand / tile_k base)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const bool isd = k < 4;
        int fe = (tid >> 2) + 64 * (isd ? k : k - 4);
        if (!isd) { int grp = fe >> 4; if (grp >= G.xgroups) grp %= G.xgroups; fe = (grp << 4) | (fe & 15); }   // partial last K-tile: finite stand-ins
        lofs[k] = (long long)fe * 16 + 4 * gg;
    }
    const size_t dtile = (size_t)N * 16, xtile = (size_t)Kp * 16;
    auto load_block = [&](int c, Raw& r) {
        const size_t q0 = (size_t)(((long long)split + (long long)c * splits) * 2);
        const bool second = c < nfull;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float* base = (k < 4 ? G.dbase + q0 * dtile : G.xbase + q0 * xtile) + lofs[k];
            r.v[k][0] = DW_STREAM_LOAD(reinterpret_cast<const f32x4*>(base));
            r.v[k][1] = second ? DW_STREAM_LOAD(reinterpret_cast<const f32x4*>(base + (k < 4 ? dtile : xtile))) : zero4;
        }
    };
    auto convert_block = [&](const Raw& r, int stage) {
        float* st = ring + stage * SF;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool isd = k < 4;
            const int fe = (tid >> 2) + 64 * (isd ? k : k - 4), second = isd ? DT : 2048;
            float* at = st + (isd ? 0 : 2 * DT) + fe * 16 + 4 * gg;
            const f32x4 v0 = r.v[k][0], v1 = r.v[k][1];
            if (isd && want_csum) csum[k & 3] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
            float& vm = isd ? vmax_d : vmax_x;
            absmax2(vm, v0[0], v0[1]); absmax2(vm, v0[2], v0[3]);
            absmax2(vm, v1[0], v1[1]); absmax2(vm, v1[2], v1[3]);
            f16x8 ph, pl;
            split2h_mix(v0, v1, isd ? sd : DW_H2_SA, ph, pl);
            *reinterpret_cast<f16x8*>(at) = ph;
            *reinterpret_cast<f16x8*>(at + second) = pl;
        }
    };
    // The two kinds of waves run SEPARATE loops with the same number of barriers: the multiplying waves' 128 accumulator
    // registers and the converting waves' 96 raw-fragment registers are then live in different regions of the control-flow graph
    // and share the 256-register budget (one loop with a role branch inside keeps both live across it: spills).
    if (mul_wave) {
        f32x4 acc[8][4] = {};
        auto block_h2 = [&](int stage) {
            const float* st = ring + stage * SF;
            const float* da = st + (wn * 128 + f) * 16 + 4 * g;            // D fragments of row-tile i: + 256 i  (lo plane: + DT)
            const float* xa = st + 2 * DT + (wk * 64 + f) * 16 + 4 * g;    // X fragments of column-tile j: + 256 j  (lo plane: + 2048)
            f16x8 bh[4], bl[4];
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(xa + 256 * j);
                bl[j] = *reinterpret_cast<const f16x8*>(xa + 256 * j + 2048);
            }
            // D fragments double-buffered behind scheduling barriers (left alone, hipcc hoists all sixteen reads: 64 registers, spills)
            f16x8 ah = *reinterpret_cast<const f16x8*>(da), al = *reinterpret_cast<const f16x8*>(da + DT);
    #pragma unroll
            for (int i = 0; i < 8; ++i) {
                f16x8 nh = ah, nl = al;
                if (i + 1 < 8) {
                    nh = *reinterpret_cast<const f16x8*>(da + 256 * (i + 1));
                    nl = *reinterpret_cast<const f16x8*>(da + 256 * (i + 1) + DT);
                }
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], acc[i][j], 0, 0, 0);
    #pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], acc[i][j], 0, 0, 0);
    #pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ah = nh; al = nl;
            }
        };

        __syncthreads();                                  // blocks 0, 1 converted
        for (int c = 0; c < nblk; ++c) {
#ifndef DW_KO_MFMA   // (knock-out builds: what does each kind of work cost?)
            block_h2(c % NST);
#endif
            __syncthreads();
        }
        __syncthreads();                                  // the converting waves' verdict on the half range
        if (*sat != 0u && guard) {   // a diverged rollout: time, never a wrong gradient
            dw_spec_exact_pass(G, ring);
            return;
        }
        dw_spec_store(G, acc, wn, wk, f, g, 1.f / (sd * DW_H2_SA));   // (powers of two: exact)
        return;
    }
#if DW_SPEC_RAW3
    Raw r2;
    load_block(0, r0);   // blocks 0, 1 converted, blocks 2, 3, 4 in flight
    if (nblk > 1) load_block(1, r1);
    if (nblk > 2) load_block(2, r2);
    convert_block(r0, 0);
    if (nblk > 3) load_block(3, r0);
    if (nblk > 1) convert_block(r1, 1);
    if (nblk > 4) load_block(4, r1);
    __syncthreads();
    auto iteration = [&](int c, Raw& r) {   // r holds block c + 2
        if (c + 2 < nblk) {
            convert_block(r, (c + 2) % NST);
            if (c + 5 < nblk) load_block(c + 5, r);
        }
        __syncthreads();
    };
    for (int c = 0; c < nblk; c += 3) {   // block c + 2 sits in r2, r0, r1, r2, ...
        iteration(c, r2);
        if (c + 1 < nblk) iteration(c + 1, r0);
        if (c + 2 < nblk) iteration(c + 2, r1);
    }
#else
    load_block(0, r0);   // blocks 0, 1 converted, blocks 2, 3 in flight
    if (nblk > 1) load_block(1, r1);
    convert_block(r0, 0);
    if (nblk > 2) load_block(2, r0);
    if (nblk > 1) convert_block(r1, 1);
    if (nblk > 3) load_block(3, r1);
    __syncthreads();
    auto iteration = [&](int c, Raw& r) {   // r holds block c + 2
        if (c + 2 < nblk) {
#ifndef DW_KO_CONV
            convert_block(r, (c + 2) % NST);   // (stage of block c - 1: its multiplication ended before the last barrier)
#endif
#ifndef DW_KO_COPY
            if (c + 4 < nblk) load_block(c + 4, r);
#endif
        }
        __syncthreads();
    };
    for (int c = 0; c < nblk; c += 2) {
        iteration(c, r0);
        if (c + 1 < nblk) iteration(c + 1, r1);
    }
#endif
#else
    DwSpecSrc SRC;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);   // (the compiler cannot see that tid >> 6 is wave-uniform: keeps the sources in SGPRs)
    dw_spec_src_init(G, uwave, SRC);
    auto copy_block = [&](int c) {   // blocks are copied in order 0, 1, 2, ...; only a split's last block may lack its second sample tile
        if (c < nfull) dw_spec_copy_next(SRC, ring, c % NST, uwave, lane);
        else { dw_spec_copy(G, ring, c, c % NST, wave, lane); dw_spec_skip(SRC, uwave); }
    };
    copy_block(0);
    if (nblk > 1) copy_block(1);
    landed_and_sync();
    if (!mul_wave) {
        if (0 < nfull) convert_stage.template operator()<false>(0);
        else convert_stage.template operator()<true>(0);
    }
    __syncthreads();
    for (int c = 0; c < nblk; ++c) {
#ifndef DW_KO_COPY   // (knock-out builds: tools/gpu/scratch - what does each kind of work cost?)
        if (c + 2 < nblk) copy_block(c + 2);
#endif
        if (mul_wave) {
#ifndef DW_KO_MFMA
            block_h2(c % NST);
#endif
        } else if (c + 1 < nblk) {
#ifndef DW_KO_CONV
            if (c + 1 < nfull) convert_stage.template operator()<false>((c + 1) % NST);
            else convert_stage.template operator()<true>((c + 1) % NST);
#endif
        }
        landed_and_sync();
    }
#endif
    if (!mul_wave) {
        vmax = fmaxf(vmax_d * sd, vmax_x * DW_H2_SA);
        if (!(vmax < 65504.f)) *sat = 1u;
        if (want_csum) {   // groups gg = tid & 3 of a feature sit in 4 adjacent lanes (formed from the fp32 values)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = csum[k];
                t += __shfl_xor(t, 1);
                t += __shfl_xor(t, 2);
                if ((tid & 3) == 0) part_b[(size_t)split * N + G.tile_n * TN + 64 * k + (tid >> 2)] = t;
            }
        }
    }
    __syncthreads();
    if (*sat != 0u && guard) {   // a diverged rollout: time, never a wrong gradient
        dw_spec_exact_pass(G, ring);
        return;
    }
#if !DW_SPEC_REGDIRECT
    if (mul_wave) dw_spec_store(G, acc, wn, wk, f, g, 1.f / (sd * DW_H2_SA));   // (powers of two: exact)
#endif
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of a layer with 16 (padded) inputs - layer 0 of the pyth_lq / pyth_idpendulum / gym / mobilerobot nets:
// dW[n][k] = sum_s D[s][n] X[s][k], k < 16.  Bound by the D stream (4 N bytes per sample; X adds 64): with 16 output columns
// the matrix core has next to nothing to do (4 MFMAs per 16 features and sample tile), so the products are EXACT fp32
// (`v_mfma_f32_16x16x4_f32`) - no operand split, no scale, no fallback; the 64 x 64-tile kernel above spends three quarters
// of its splits and MFMAs on padding columns here (cfg5: 138 us for 335 MB).  Lane (f, g) of a fragment holds samples
// 4g .. 4g+3 of feature f - ONE 16-byte load per 16 features x 16 samples, a contiguous KiB per wave; MFMA e of a tile
// contracts samples {e, 4 + e, 8 + e, 12 + e} (element e of every lane's vector, the same order for both operands).
// Workgroup = 4 waves x up to 4 blocks of 16 features (256 features); split s takes the sample tiles s, s + splits, ...;
// two batches of DWK_U tiles alternate in registers (one in flight while the other is multiplied).
// ---------------------------------------------------------------------------------------------
#define DWK_U 2
__global__ __launch_bounds__(NTHREADS, 3) void dw_skinny_kernel(const float* __restrict__ D, int N, const float* __restrict__ X, long long Q,
                                                                 int splits, float* __restrict__ part, float* __restrict__ part_b) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = lane & 15, g = lane >> 4;
    const int ngroups = (N + 255) / 256;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;   // XCD-aware order, as above
    const int grp = local % ngroups, split = (local / ngroups) * 8 + xcd;
    if (split >= splits) return;
    int doff[4], nfeat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        nfeat[i] = 256 * grp + 16 * (wave + 4 * i);                       // first feature of this wave's block i
        doff[i] = min(nfeat[i] + f, N - 1) * 16 + 4 * g;                  // (blocks past N: clamped, never stored)
    }
    const int xoff = f * 16 + 4 * g;
    const GLOBAL_AS float* Dg = gptr(D);
    const GLOBAL_AS float* Xg = gptr(X);
    const size_t dtile = (size_t)N * 16;
    struct Batch { f32x4 d[DWK_U][4]; f32x4 x[DWK_U]; };
    auto load = [&](long long q0, Batch& b) {   // tiles q0, q0 + splits, ...: past the end -> the last tile with X = 0
#pragma unroll
        for (int u = 0; u < DWK_U; ++u) {
            const long long q = q0 + (long long)u * splits;
            const size_t qq = (size_t)min(q, Q - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) b.d[u][i] = DW_STREAM_LOAD(reinterpret_cast<const GLOBAL_AS f32x4*>(Dg + qq * dtile + doff[i]));
            b.x[u] = DW_STREAM_LOAD(reinterpret_cast<const GLOBAL_AS f32x4*>(Xg + qq * 256 + xoff));
        }
    };
    f32x4 acc[4] = {};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto mul = [&](long long q0, const Batch& b) {
#pragma unroll
        for (int u = 0; u < DWK_U; ++u) {
            const bool live = q0 + (long long)u * splits < Q;
            f32x4 x = b.x[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = live ? x[e] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 d = b.d[u][i];
                bsum[i] += live ? (d[0] + d[1]) + (d[2] + d[3]) : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[e], x[e], acc[i], 0, 0, 0);
            }
        }
    };
    const long long stride = (long long)DWK_U * splits;
    Batch b0, b1;
    load(split, b0);
    for (long long q = split; q < Q; q += 2 * stride) {
        load(q + stride, b1);
        mul(q, b0);
        load(q + 2 * stride, b0);
        mul(q + stride, b1);
    }
    float* pbase = part + (size_t)split * N * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (nfeat[i] >= N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) pbase[(size_t)(nfeat[i] + 4 * g + r) * 16 + f] = acc[i][r];
        if (part_b != nullptr) {
            float t = bsum[i];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (g == 0) part_b[(size_t)split * N + nfeat[i] + f] = t;
        }
    }
}
// GOPS_VF_DW_NO_SKINNY: the 64 x 64-tile kernel for these layers too
bool dw_skinny_ok(int N, int Kp, unsigned vflags) {
    return !(vflags & GOPS_VF_DW_NO_SKINNY) && Kp == 16 && (N % 16) == 0;
}

// The layers the wave-specialised kernel takes (GOPS_VF_DW_NO_SPEC: the 4-wave ring kernel)
static bool dw_spec_ok(int N, int Kp, unsigned vflags) {
    return !(vflags & GOPS_VF_DW_NO_SPEC) && (N % 256) == 0 && ((Kp % 128) == 0 || (Kp > 128 && (Kp % 16) == 0));
}

// dscale: device pointer to max|grad_v| of the launch (the deltas' magnitude reference), or null: with it the large
// layers run the two-half-plane products (22 significant bits per operand), without it - or with GOPS_DW_EXACT set -
// the exact three-plane bf16 split.
hipError_t launch_dw_gemm(const float* D, int N, const float* X, int Kp, long long S, int splits,
                          int chunks_per_split, float* part, float* part_b, bool big, hipStream_t s, const float* dscale, unsigned vflags) {
    const bool force_f32 = (vflags & GOPS_VF_DW_F32) != 0;       // A/B knob: fp32 MFMA GEMM
    const bool force_exact = (vflags & GOPS_VF_DW_EXACT) != 0;
    const bool no_guard = (vflags & GOPS_VF_DW_NO_GUARD) != 0;   // test knob: skip the exact re-run behind a saturated launch
    const long long Q = (S + TB - 1) / TB;
    const int T = big ? 128 : 64, tiles = ((N + T - 1) / T) * ((Kp + T - 1) / T);
    const dim3 grid(tiles * ((splits + 7) / 8) * 8), block(NTHREADS);
    const bool no_ring = (vflags & GOPS_VF_DW_DIRECT) != 0;   // A/B knob: register-direct kernel for the large layers too
    if (big && !force_f32 && !no_ring && (N % 128) == 0 && ((Kp % 128) == 0 || (Kp > 128 && (Kp % 16) == 0))) {
        const float* none = nullptr;
        if (dscale != nullptr && !force_exact && dw_spec_ok(N, Kp, vflags)) {
            const dim3 grids(((N / 256) * ((Kp + 127) / 128)) * ((splits + 7) / 8) * 8);
            launch_with_lds(dw_gemm_spec_kernel, grids, dim3(512), DwSpec::lds_bytes(), s, D, N, X, Kp, Q, splits, part, part_b, dscale,
                            no_guard ? 0 : 1);
        } else if (dscale != nullptr && !force_exact) {
            launch_with_lds(dw_gemm_ring_kernel<true>, grid, block, (size_t)DWR_STAGES * DWR_STAGE_FLOATS * sizeof(float) + 16, s, D, N, X, Kp, Q,
                            splits, chunks_per_split, part, part_b, dscale, no_guard ? 0 : 1);
        } else {
            launch_with_lds(dw_gemm_ring_kernel<false>, grid, block, (size_t)DWR_STAGES * DWR_STAGE_FLOATS * sizeof(float), s, D, N, X, Kp, Q,
                            splits, chunks_per_split, part, part_b, none, 1);
        }
    }
    else if (!force_f32 && dw_skinny_ok(N, Kp, vflags))
        hipLaunchKernelGGL(dw_skinny_kernel, dim3(((N + 255) / 256) * ((splits + 7) / 8) * 8), block, 0, s, D, N, X, Q, splits, part, part_b);
    else if (big && !force_f32) hipLaunchKernelGGL((dw_gemm_fm_kernel<4, true>), grid, block, 0, s, D, N, X, Kp, Q, splits, chunks_per_split, part, part_b);
    else if (big) hipLaunchKernelGGL((dw_gemm_fm_kernel<4, false>), grid, block, 0, s, D, N, X, Kp, Q, splits, chunks_per_split, part, part_b);
    else if (!force_f32) hipLaunchKernelGGL((dw_gemm_fm_kernel<2, true>), grid, block, 0, s, D, N, X, Kp, Q, splits, chunks_per_split, part, part_b);
    else hipLaunchKernelGGL((dw_gemm_fm_kernel<2, false>), grid, block, 0, s, D, N, X, Kp, Q, splits, chunks_per_split, part, part_b);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// dW GEMM, half-precision operands (GOPS_DTYPE_F16):  part[split][n][k] = sum_s D[s][n] * X[s][k] with D, X
// row-major _Float16 stash tiles, fp32 accumulation on v_mfma_f32_16x16x32_f16.  Same decomposition as the
// bf16x3 kernel (128 x 128 outputs per workgroup, 2 x 2 waves of 64 x 64, split-K over samples, XCD-aware
// block order); a chunk is 64 samples = two MFMA K-steps.  The contraction index (samples) is the ROW
// index of both operands in memory, so the staging transposes: a thread loads 4 rows x 8 columns (one
// dwordx4 per row, 256-byte coalesced rows), regroups them with v_perm_b32 into 8 columns x 4 samples and
// writes [g = s/8][column][8 samples] fragments (ds_write_b64); a lane's MFMA fragment is one ds_read_b128.
// 16-byte units are XOR-swizzled (column ^ (column >> 3 & 7)) so that both directions are conflict-free.
// ---------------------------------------------------------------------------------------------
#define DWH_SC 64                       // samples per staged chunk
#define DWH_T 128                       // tile edge
#define DWH_PLANE (8 * DWH_T * 8)       // halfs per operand: [8 g][128 cols][8 samples]

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(NTHREADS, 2) void dw_gemm_f16_kernel(const _Float16* __restrict__ D, int N,
                                                                   const _Float16* __restrict__ X, int Kp,
                                                                   long long S, int splits, int chunks_per_split,
                                                                   float* __restrict__ part,
                                                                   float* __restrict__ part_b) {
    __shared__ __attribute__((aligned(16))) _Float16 As[DWH_PLANE];   // D^T
    __shared__ __attribute__((aligned(16))) _Float16 Bs[DWH_PLANE];   // X
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_k = (Kp + DWH_T - 1) / DWH_T, tiles = tiles_k * ((N + DWH_T - 1) / DWH_T);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;   // same XCD-aware order as dw_gemm_kernel
    const int tile = local % tiles, split = (local / tiles) * 8 + xcd;
    if (split >= splits) return;
    const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
    const int n0 = tile_n * DWH_T, k0 = tile_k * DWH_T;
    const long long s_begin = (long long)split * chunks_per_split * DWH_SC;
    const int wn = wave >> 1, wk = wave & 1;

    // Staging item of this thread: columns 8*c8 .. +7, samples 4*s4 .. +3 of the chunk.  Lane bits ->
    // (s4 bit 0, c8 bits 0..2, c8 bit 3, s4 bit 1), wave -> s4 bits 2..3: one load instruction covers four
    // full 256-byte rows, and the 16 lanes of a ds_write_b64 group hit 16 distinct 8-byte slots.
    const int c8 = ((lane >> 1) & 7) | (((lane >> 4) & 1) << 3);
    const int s4 = (lane & 1) | (((lane >> 5) & 1) << 1) | (wave << 2);
    int wofs[8];   // LDS half-offset of column 8*c8 + i
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = 8 * c8 + i, nsw = n ^ ((n >> 3) & 7);
        wofs[i] = (((s4 >> 1) * DWH_T + nsw) << 3) + ((s4 & 1) << 2);
    }
    const bool want_bias = part_b != nullptr && tile_k == 0;

    f32x4 acc[4][4] = {};
    u32x4 dreg[4], xreg[4];
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto gload = [&](long long s0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long srow = s0 + 4 * s4 + r;
            const int n = n0 + 8 * c8, k = k0 + 8 * c8;
            const u32x4 z = {0u, 0u, 0u, 0u};
            // (default cache policy: with 128 x 128 output tiles every operand slice is read by two workgroups, the second time from
            // L2 - the non-temporal policy measured 2 % slower here, cfg5 fp16)
            dreg[r] = (srow < S && n < N) ? *reinterpret_cast<const u32x4*>(D + srow * N + n) : z;
            xreg[r] = (srow < S && k < Kp) ? *reinterpret_cast<const u32x4*>(X + srow * Kp + k) : z;
        }
    };
    // 4 rows x 8 halfs -> for each column pair (2w, 2w+1): samples 0..3 as two dwords each
    auto lstore = [&]<bool SUM>(const u32x4& q0, const u32x4& q1, const u32x4& q2, const u32x4& q3, _Float16* base) {
        #pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned r0w = q0[w], r1w = q1[w], r2w = q2[w], r3w = q3[w];
            // (low halfs of rows 0,1), (rows 2,3) = column 2w; (high halfs) = column 2w+1  (hipcc folds these to v_perm_b32)
            const unsigned lo01 = (r0w & 0xffffu) | (r1w << 16), lo23 = (r2w & 0xffffu) | (r3w << 16);
            const unsigned hi01 = (r0w >> 16) | (r1w & 0xffff0000u), hi23 = (r2w >> 16) | (r3w & 0xffff0000u);
            { const u32x2 v = {lo01, lo23}; *reinterpret_cast<u32x2*>(base + wofs[2 * w]) = v; }
            { const u32x2 v = {hi01, hi23}; *reinterpret_cast<u32x2*>(base + wofs[2 * w + 1]) = v; }
            if (SUM && want_bias) {
                const f16x2 a0 = __builtin_bit_cast(f16x2, lo01), a1 = __builtin_bit_cast(f16x2, lo23);
                const f16x2 b0 = __builtin_bit_cast(f16x2, hi01), b1 = __builtin_bit_cast(f16x2, hi23);
                bsum[2 * w] += ((float)a0[0] + (float)a0[1]) + ((float)a1[0] + (float)a1[1]);
                bsum[2 * w + 1] += ((float)b0[0] + (float)b0[1]) + ((float)b1[0] + (float)b1[1]);
            }
        }
    };
    // fragment (8 consecutive samples of tile-local column `col`) of this lane in K-step ks
    auto frag = [&](const _Float16* base, int ks, int col) {
        const int nsw = col ^ ((col >> 3) & 7);
        return *reinterpret_cast<const f16x8*>(base + (((4 * ks + (lane >> 4)) * DWH_T + nsw) << 3));
    };

    gload(s_begin);
    for (int c = 0; c < chunks_per_split; ++c) {
        __syncthreads();
        lstore.template operator()<true>(dreg[0], dreg[1], dreg[2], dreg[3], As);
        lstore.template operator()<false>(xreg[0], xreg[1], xreg[2], xreg[3], Bs);
        __syncthreads();
        if (c + 1 < chunks_per_split) gload(s_begin + (long long)(c + 1) * DWH_SC);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = frag(Bs, ks, wk * 64 + 16 * j + (lane & 15));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 a = frag(As, ks, wn * 64 + 16 * i + (lane & 15));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* pbase = part + (size_t)split * N * Kp;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wk * 64 + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + 16 * i + 4 * (lane >> 4) + r;
                if (n < N && k < Kp) pbase[(size_t)n * Kp + k] = acc[i][j][r];
            }
        }
    if (want_bias) {   // column sums of D: 16 sample groups per column, combined in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(As);   // [16][128] floats = 8 KiB of the 16 KiB plane
#pragma unroll
        for (int i = 0; i < 8; ++i) red[s4 * DWH_T + 8 * c8 + i] = bsum[i];
        __syncthreads();
        if (tid < DWH_T && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q * DWH_T + tid];
            part_b[(size_t)split * N + n0 + tid] = t;
        }
    }
}

hipError_t launch_dw_gemm_f16(const void* D, int N, const void* X, int Kp, long long S, int splits,
                              int chunks_per_split, float* part, float* part_b, hipStream_t s) {
    const int tiles = ((N + DWH_T - 1) / DWH_T) * ((Kp + DWH_T - 1) / DWH_T);
    hipLaunchKernelGGL(dw_gemm_f16_kernel, dim3(tiles * ((splits + 7) / 8) * 8), dim3(NTHREADS), 0, s,
                       static_cast<const _Float16*>(D), N, static_cast<const _Float16*>(X), Kp, S, splits, chunks_per_split,
                       part, part_b);
    return hipGetLastError();
}

// Output layer (width A <= 4) on the VALU: part[split][a][k] = sum_s dy[s][a] * h[s][k].
// Row-major half h (GOPS_DTYPE_F16 stash): thread = (8 columns, sample lane): one 16-byte read of h per sample, four samples
// in flight per thread (a workgroup keeps 16 KiB of the stream in flight; the round-3 form - 8-byte reads, 8 KiB - ran at
// 2.6 TB/s); the two sample lanes of a wave are combined by a lane exchange, the four waves through LDS, in a fixed order.
__global__ __launch_bounds__(NTHREADS) void dw_out_h_kernel(const float* __restrict__ dy, const _Float16* __restrict__ h, int K, int A,
                                                            long long S, long long per_split, float* __restrict__ part,
                                                            float* __restrict__ part_b) {
    __shared__ __attribute__((aligned(16))) float red[4][GOPS_MAX_ACT][256];
    __shared__ float redb[4][GOPS_MAX_ACT];
    const int split = blockIdx.x, tid = threadIdx.x, c8 = tid & 31, sl = tid >> 5, wave = tid >> 6;
    const long long s0 = split * per_split, s1 = min(S, s0 + per_split);
    const int nblk = K >> 3;   // (half nets: K is a multiple of 64)
    for (int cb0 = 0; cb0 < nblk; cb0 += 32) {   // uniform trip count: every lane reaches the barriers
        const int cb = cb0 + c8;
        const bool col_ok = cb < nblk;
        float acc[GOPS_MAX_ACT][8] = {};
        float accb[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
        if (col_ok) {
            const GLOBAL_AS _Float16* hp = gptr(h) + 8 * cb;
#pragma unroll 4
            for (long long sidx = s0 + sl; sidx < s1; sidx += 8) {
                const f32x4 g = *gptr(reinterpret_cast<const f32x4*>(dy) + sidx);
                const f16x8 hv = *reinterpret_cast<const GLOBAL_AS f16x8*>(hp + sidx * K);
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[a][e] = fmaf(g[a], (float)hv[e], acc[a][e]);
                    accb[a] += g[a];
                }
            }
        }
#pragma unroll
        for (int a = 0; a < GOPS_MAX_ACT; ++a) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[a][e] += __shfl_xor(acc[a][e], 32);
            accb[a] += __shfl_xor(accb[a], 32);
        }
        if ((tid & 32) == 0) {
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a) {
                *reinterpret_cast<f32x4*>(&red[wave][a][8 * c8]) = f32x4{acc[a][0], acc[a][1], acc[a][2], acc[a][3]};
                *reinterpret_cast<f32x4*>(&red[wave][a][8 * c8 + 4]) = f32x4{acc[a][4], acc[a][5], acc[a][6], acc[a][7]};
            }
            if (c8 == 0 && cb0 == 0)
                for (int a = 0; a < GOPS_MAX_ACT; ++a) redb[wave][a] = accb[a];
        }
        __syncthreads();
        const int k = 8 * cb0 + tid;
        if (k < K)
            for (int a = 0; a < A; ++a)
                part[((size_t)split * A + a) * K + k] = (red[0][a][tid] + red[1][a][tid]) + (red[2][a][tid] + red[3][a][tid]);
        if (tid == 0 && cb0 == 0)
            for (int a = 0; a < A; ++a) part_b[(size_t)split * A + a] = (redb[0][a] + redb[1][a]) + (redb[2][a] + redb[3][a]);
        __syncthreads();
    }
}

// Feature-major fp32 h (common.h StashDev): a split is a run of whole sample tiles; thread k owns feature k - its 16
// rows of a tile are 64 contiguous bytes, a wave reads 4 contiguous KiB per tile - and needs no cross-thread reduction;
// the tile's 16 x 4 block of dy is the same for every thread (uniform loads through the scalar cache).
__global__ __launch_bounds__(NTHREADS) void dw_out_fm_kernel(const float* __restrict__ dy, const float* __restrict__ h, int K, int A,
                                                             long long Q, long long tiles_per_split,
                                                             float* __restrict__ part, float* __restrict__ part_b) {
    const int split = blockIdx.x;
    const long long q0 = split * tiles_per_split, q1 = min(Q, q0 + tiles_per_split);
    for (int k = threadIdx.x; k < ((K + NTHREADS - 1) / NTHREADS) * NTHREADS; k += NTHREADS) {
        const bool col_ok = k < K;
        const int kc = col_ok ? k : K - 1;
        float acc[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, accb[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
        for (long long qb = q0; qb < q1; qb += 2) {   // two tiles (8 vectors) in flight per thread
            f32x4 hv[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long long q = min(qb + u, q1 - 1);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) hv[u][rg] = ld4(gptr(h) + ((size_t)q * K + kc) * 16 + 4 * rg);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (qb + u < q1) {
                    const f32x4* gp = reinterpret_cast<const f32x4*>(dy) + (size_t)(qb + u) * 16;   // uniform: [16 rows][4]
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const f32x4 g = gp[m];
#pragma unroll
                        for (int a = 0; a < GOPS_MAX_ACT; ++a) { acc[a] += g[a] * hv[u][m >> 2][m & 3]; accb[a] += g[a]; }
                    }
                }
            }
        }
        if (col_ok) {
            for (int a = 0; a < A; ++a) part[((size_t)split * A + a) * K + k] = acc[a];
            if (k == 0)
                for (int a = 0; a < A; ++a) part_b[(size_t)split * A + a] = accb[a];
        }
    }
}

hipError_t launch_dw_out(const float* dy, const float* h, bool h_is_half, int K, int A, long long S, int splits,
                         float* part, float* part_b, hipStream_t s) {
    if (h_is_half) {
        const long long per = (S + splits - 1) / splits;
        hipLaunchKernelGGL(dw_out_h_kernel, dim3(splits), dim3(NTHREADS), 0, s, dy, reinterpret_cast<const _Float16*>(h), K, A, S, per, part,
                           part_b);
    } else {   // splits beyond the tile count produce zero slabs (their loops are empty)
        const long long Q = (S + TB - 1) / TB, per = (Q + splits - 1) / splits;
        hipLaunchKernelGGL(dw_out_fm_kernel, dim3(splits), dim3(NTHREADS), 0, s, dy, h, K, A, Q, per, part, part_b);
    }
    return hipGetLastError();
}

// out[r][c] = sum_split part[split][r][c]   for c < cols (row stride `ld` inside a split), for every
// job of the table in ONE launch.  Block = 64 outputs x 4 split lanes; each lane sums every 4th split
// with independent loads in flight, then the 4 lanes are combined through LDS (fixed order ->
// deterministic).
#define LOSS_BLOCKS 64
#define LOSS_ONE_BLOCK_MAX 8192   // up to here ONE block forms the sums (<= 32 elements per thread, no partials / ticket round)
// sum of (s0, s1) over the block in a fixed order: butterfly inside each wave, then the four waves in order; every thread returns the totals
__device__ __forceinline__ void block_sum2(double& s0, double& s1, double (*red)[4]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
    }
    __syncthreads();   // (red may still be read by a previous call)
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    s0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
}
// Sum of n values in double, one block, in batch_loss_kernel's single-block order (the two produce the same bits): mean_stats[0] =
// sc * mean(x), mean_stats[1] = mean(x).
__device__ __forceinline__ void mean_block(const float* __restrict__ a, int n, float sc, float* __restrict__ stats, double (*red2)[4]) {
    double s0 = 0.0, s1 = 0.0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
        float av[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + q * 256;
            av[q] = i < n ? a[i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i0 + q * 256 < n) { s0 += (double)av[q]; s1 += (double)av[q]; }
    }
    block_sum2(s0, s1, red2);
    if (threadIdx.x == 0) {
        stats[0] = (float)((double)sc * s0 / (double)n);
        stats[1] = (float)(s1 / (double)n);
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceJobs jobs) {
    __shared__ float red[4][64];
    __shared__ double red2[2][4];
    if (jobs.reset != nullptr && blockIdx.x == 0 && threadIdx.x == 0) jobs.reset[0] = jobs.reset[1] = 0.f;
    // fused Adam step (gops_rollout_backward_update): this step's scalar factors, left by the sweep kernel of the same call
    // (common.h adam_snapshot: adam_kernel's, formed once instead of per block)
    const bool adam = jobs.ad_snap != nullptr;
    float step_size = 0.f, bc2_sqrt = 1.f, omb1 = 0.f, omb2 = 0.f, b2 = 0.f, gsc = 1.f;
    if (adam) {
        step_size = jobs.ad_snap[0]; bc2_sqrt = jobs.ad_snap[1]; gsc = jobs.ad_snap[2];
        omb1 = (float)(1.0 - jobs.ad_b1); omb2 = (float)(1.0 - jobs.ad_b2); b2 = (float)jobs.ad_b2;
    }
    if ((int)blockIdx.x >= jobs.block0[jobs.n]) {   // the extra block: the loss mean
        mean_block(jobs.mean_x, jobs.mean_n, jobs.mean_sc, jobs.mean_stats, red2);
    } else {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.block0[j + 1]) ++j;
    const float* __restrict__ part = jobs.part[j];
    const int splits = jobs.splits[j], rows = jobs.rows[j], cols = jobs.cols[j], ld = jobs.ld[j];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int idx = (blockIdx.x - jobs.block0[j]) * 64 + o;
    const bool valid = idx < rows * cols;
    // the element's Adam operands travel with the partial sums (they are needed when the sums are)
    float pm = 0.f, pv = 0.f, pp = 0.f;
    const bool step_here = adam && jobs.ad_p[j] != nullptr && sl == 0 && valid;
    float pt = 0.f;
    const bool avg_here = step_here && jobs.pk_t[j] != nullptr;
    if (step_here) { pm = jobs.ad_m[j][idx]; pv = jobs.ad_v[j][idx]; pp = jobs.ad_p[j][idx]; }
    if (avg_here) pt = jobs.pk_t[j][idx];
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (valid) {
        const int r = idx / cols, c = idx - r * cols;
        const size_t stride = (size_t)jobs.slab_rows[j] * ld;
        const float* p = part + (size_t)r * ld + c;
        int s = sl;
        // 8 independent loads in flight per thread (the kernel is latency-bound: one 256-byte row segment per wave and
        // split); the pairs are added in a fixed order, so the result does not depend on the unrolling
        // (jobs with hundreds of splits - one slab per sweep workgroup, rollout_h64.hip's fused first-layer gradient: 1024 at cfg5 -
        // take 16 at a time; element i of a thread's sequence goes to accumulator i % 4 in every form of the loop, so the sums
        // do not depend on which form ran)
        for (; s + 60 < splits; s += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = DW_STREAM_LOAD(p + (size_t)(s + 4 * u) * stride);
#pragma unroll
            for (int u = 0; u < 16; u += 4) { acc0 += v[u]; acc1 += v[u + 1]; acc2 += v[u + 2]; acc3 += v[u + 3]; }
        }
        for (; s + 28 < splits; s += 32) {
            const float v0 = DW_STREAM_LOAD(p + (size_t)s * stride), v1 = DW_STREAM_LOAD(p + (size_t)(s + 4) * stride),
                        v2 = DW_STREAM_LOAD(p + (size_t)(s + 8) * stride), v3 = DW_STREAM_LOAD(p + (size_t)(s + 12) * stride),
                        v4 = DW_STREAM_LOAD(p + (size_t)(s + 16) * stride), v5 = DW_STREAM_LOAD(p + (size_t)(s + 20) * stride),
                        v6 = DW_STREAM_LOAD(p + (size_t)(s + 24) * stride), v7 = DW_STREAM_LOAD(p + (size_t)(s + 28) * stride);
            acc0 += v0; acc1 += v1; acc2 += v2; acc3 += v3;
            acc0 += v4; acc1 += v5; acc2 += v6; acc3 += v7;
        }
        for (; s + 12 < splits; s += 16) {
            acc0 += p[(size_t)s * stride];
            acc1 += p[(size_t)(s + 4) * stride];
            acc2 += p[(size_t)(s + 8) * stride];
            acc3 += p[(size_t)(s + 12) * stride];
        }
        for (; s < splits; s += 4) acc0 += p[(size_t)s * stride];
    }
    red[sl][o] = (acc0 + acc1) + (acc2 + acc3);
    __syncthreads();
    if (sl == 0 && valid) {
        float t = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
        if (jobs.unscale != nullptr) t *= 1.f / f16_grad_scale(*jobs.unscale);   // power of two: exact
        // a plane conversion of the forward or of the sweep left the half range (their tiles' returns / one delta are NaN already):
        // no element of this call's gradient is to be trusted - all of them leave as NaN, and none takes an optimizer step
        if (jobs.poison != nullptr && *jobs.poison != 0u) t = __builtin_nanf("");
        jobs.out[j][idx] = t;
        if (step_here && !grad_is_finite(t * gsc)) {
            atomicAdd(jobs.ad_skipped, 1u);   // (only on the failure path: no contention in a healthy update)
        } else if (step_here) {   // adam_kernel's arithmetic, operation for operation
            const float gi = t * gsc;
            const float mi = pm + (gi - pm) * omb1;
            const float vi = pv * b2 + omb2 * gi * gi;
            jobs.ad_m[j][idx] = mi;
            jobs.ad_v[j][idx] = vi;
            const float pn = pp - step_size * (mi / (sqrtf(vi) / bc2_sqrt + jobs.ad_eps));
            jobs.ad_p[j][idx] = pn;
            if (avg_here) jobs.pk_t[j][idx] = rn_add(rn_mul(pt, jobs.pk_omt), rn_mul(jobs.pk_tau, pn));   // polyak_kernel's roundings
        }
    }
    }
}

void reduce_jobs_add(ReduceJobs& jobs, const float* part, int splits, int rows, int cols, int ld, float* out, int slab_rows) {
    const int i = jobs.n++;
    jobs.part[i] = part; jobs.out[i] = out;
    jobs.splits[i] = splits; jobs.rows[i] = rows; jobs.cols[i] = cols; jobs.ld[i] = ld;
    jobs.slab_rows[i] = slab_rows > 0 ? slab_rows : rows;
    if (i == 0) jobs.block0[0] = 0;
    jobs.block0[i + 1] = jobs.block0[i] + (rows * cols + 63) / 64;
}

hipError_t launch_batch_loss(const float* a, const float* b, int n, float gsc, float sc0, float* grad, float* stats, hipStream_t s);
hipError_t launch_reduce(const ReduceJobs& jobs_in, hipStream_t s) {
    if (jobs_in.n == 0) return hipSuccess;
    ReduceJobs jobs = jobs_in;
    const bool own_mean = jobs.mean_x != nullptr && jobs.mean_n > LOSS_ONE_BLOCK_MAX;   // too long for one block: its own launch
    if (own_mean) jobs.mean_x = nullptr;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(jobs.block0[jobs.n] + (jobs.mean_x != nullptr ? 1 : 0)), dim3(256), 0, s, jobs);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && own_mean) e = launch_batch_loss(jobs_in.mean_x, nullptr, jobs_in.mean_n, 0.f, jobs_in.mean_sc, nullptr, jobs_in.mean_stats, s);
    return e;
}

// ---------------------------------------------------------------------------------------------
// Output layer of ANY width W (gops_mlp_forward / _backward: FiniteHorizonFullPolicy emits act_dim * pre_horizon
// values): y = h Wo^T + b over the stashed last hidden activation h [S][K] (tile-major rows == batch rows for H = 1),
// and its input adjoint g_h = g_y Wo together with a zero-padded copy g_yp [S][Wp] of g_y that the regular dW GEMM
// takes as its delta operand.  63 MFLOP at B = 4096, W = 60: plain VALU, 16 rows per block staged in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_out_fwd_kernel(const float* __restrict__ h, int K, const float* __restrict__ Wo,
                                                             const float* __restrict__ bo, int W, int B, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float hs[];   // [16][K + 4] row-major copy of the FM tile
    const int b0 = blockIdx.x * 16, rows = min(16, B - b0), ld = K + 4;
    for (int i = threadIdx.x; i < 4 * K; i += 256) {   // 16-byte unit i of the tile: feature i >> 2, rows 4 (i & 3) .. +3
        const f32x4 v = reinterpret_cast<const f32x4*>(h + (size_t)blockIdx.x * 16 * K)[i];
        const int k = i >> 2, m0 = (i & 3) << 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) hs[(m0 + r) * ld + k] = v[r];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 16 * W; o += 256) {   // o = m * W + w: consecutive threads -> consecutive y elements
        const int m = o / W, w = o - m * W;
        if (m >= rows) continue;
        const f32x4* wr = reinterpret_cast<const f32x4*>(Wo + (size_t)w * K);
        const f32x4* hr = reinterpret_cast<const f32x4*>(hs + m * ld);
        float acc = 0.f;
        for (int c = 0; c < (K >> 2); ++c) {
            const f32x4 a = hr[c], bq = wr[c];
            acc += a[0] * bq[0] + a[1] * bq[1] + a[2] * bq[2] + a[3] * bq[3];
        }
        y[(size_t)(b0 + m) * W + w] = acc + bo[w];
    }
}

// g_h stays row-major [S][K] (rollout_bwd reads it as `ext_delta`); g_yp is the FM delta operand [S/16][Wp][16]
__global__ __launch_bounds__(256) void linear_out_bwd_kernel(const float* __restrict__ gy, int W, int Wp, const float* __restrict__ Wo,
                                                             int K, int B, long long S, float* __restrict__ gh, float* __restrict__ gyp) {
    extern __shared__ __attribute__((aligned(16))) float gs[];   // [16][Wp]
    const long long s0 = (long long)blockIdx.x * 16;
    for (int i = threadIdx.x; i < 16 * Wp; i += 256) {
        const int m = i / Wp, w = i - m * Wp;
        gs[i] = (s0 + m < B && w < W) ? gy[(size_t)(s0 + m) * W + w] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * Wp; i += 256) {   // i = w * 16 + m: the FM tile in memory order
        const int w = i >> 4, m = i & 15;
        gyp[(size_t)blockIdx.x * 16 * Wp + i] = gs[m * Wp + w];
    }
    for (int o = threadIdx.x; o < 16 * K; o += 256) {   // o = m * K + k: coalesced reads of Wo rows and writes of g_h
        const int m = o / K, k = o - m * K;
        if (s0 + m >= S) continue;
        float acc = 0.f;
        for (int w = 0; w < W; ++w) acc += gs[m * Wp + w] * Wo[(size_t)w * K + k];
        gh[(size_t)(s0 + m) * K + k] = acc;
    }
}

hipError_t launch_linear_out_fwd(const float* h, int K, const float* Wo, const float* bo, int W, int B, float* y, hipStream_t s) {
    hipLaunchKernelGGL(linear_out_fwd_kernel, dim3((B + 15) / 16), dim3(256), 16 * (K + 4) * sizeof(float), s, h, K, Wo, bo, W, B, y);
    return hipGetLastError();
}

hipError_t launch_linear_out_bwd(const float* gy, int W, int Wp, const float* Wo, int K, int B, long long S, float* gh,
                                 float* gyp, hipStream_t s) {
    hipLaunchKernelGGL(linear_out_bwd_kernel, dim3((unsigned)((S + 15) / 16)), dim3(256), 16 * Wp * sizeof(float), s, gy, W, Wp, Wo,
                       K, B, S, gh, gyp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Adam over a table of tensors, one launch.  Same update as torch.optim.Adam (foreach path):
//   m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// lr and the step count are read from device memory (graph-replayable); the scalar factors are
// formed in double like torch's host code, then rounded to fp32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(const GopsAdamTensors T, GopsAdamState* st, double beta1,
                                                   double beta2, float eps) {
    const double b1p = st->beta1_pow * beta1, b2p = st->beta2_pow * beta2;   // beta^t, t = step + 1
    const float step_size = (float)(st->lr / (1.0 - b1p)), bc2_sqrt = (float)sqrt(1.0 - b2p);
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2), b2 = (float)beta2;
    const float gsc = (float)st->grad_scale;   // 1/N of the data-parallel mean (1 otherwise)
    const int ti = blockIdx.y;
    const long long n = T.numel[ti];
    float* __restrict__ p = T.param[ti];
    const float* __restrict__ g = T.grad[ti];
    float* __restrict__ m = T.exp_avg[ti];
    float* __restrict__ v = T.exp_avg_sq[ti];
    // 4 elements per thread with all 16 loads in flight together: the kernel is a chain of memory round trips (load g, m, v, p - store),
    // and a load - store loop pays one per element (65536-element layers, 64 blocks: 9.2 us; one element per thread on 256 blocks costs
    // more than it saves - every block takes the ticket below)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float gq[4], mq[4], vq[4], pq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long i = i0 + q * stride;
            const bool ok = i < n;
            gq[q] = ok ? g[i] : 0.f; mq[q] = ok ? m[i] : 0.f; vq[q] = ok ? v[i] : 0.f; pq[q] = ok ? p[i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long i = i0 + q * stride;
            if (i < n && !grad_is_finite(gq[q] * gsc)) {
                atomicAdd(&st->skipped_nonfinite, 1u);   // no step for this element (common.h grad_is_finite)
            } else if (i < n) {
                const float gi = gq[q] * gsc;
                const float mi = mq[q] + (gi - mq[q]) * omb1;
                const float vi = vq[q] * b2 + omb2 * gi * gi;
                m[i] = mi;
                v[i] = vi;
                p[i] = pq[q] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
            }
        }
    }
    // Every block has read the state (its values feed the arithmetic above) before it takes a ticket;
    // the last one publishes the advanced state.  No fence: the next reader is a later kernel.
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y;
        if (atomicAdd(&st->ticket, 1u) == total - 1) {
            st->ticket = 0;
            st->step += 1;
            st->beta1_pow = b1p;
            st->beta2_pow = b2p;
        }
    }
}

// Polyak averaging of target networks (gops/algorithm/infadp.py:124-133: p_targ.mul_(1 - tau); p_targ.add_(tau * p)), all
// tensors of a network in ONE launch; the same two roundings per element as the two torch passes it replaces.
// Table: param[] = target tensors, grad[] = online tensors (GopsAdamTensors reused; the moment slots are ignored).
__global__ __launch_bounds__(256) void polyak_kernel(const GopsAdamTensors T, float omt, float tau) {
    const int ti = blockIdx.y;
    const long long n = T.numel[ti];
    float* __restrict__ t = T.param[ti];
    const float* __restrict__ o = T.grad[ti];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        t[i] = rn_add(rn_mul(t[i], omt), rn_mul(tau, o[i]));
}
hipError_t launch_polyak(const GopsAdamTensors& T, float omt, float tau, hipStream_t s) {
    long long nmax = 1;
    for (int i = 0; i < T.n; ++i) nmax = std::max<long long>(nmax, T.numel[i]);
    const int bx = (int)std::min<long long>((nmax + 255) / 256, 64);
    hipLaunchKernelGGL(polyak_kernel, dim3(bx, T.n), dim3(256), 0, s, T, omt, tau);
    return hipGetLastError();
}

// Loss scalars of a batch in ONE launch (they were three torch reductions + three elementwise passes per INFADP update pair:
// ~10 % of a cfg5 fp16 update).  a, b: [n]; grad (nullable) <- gsc * (a - b);  stats[0] <- sum((a - b)^2) / n (b null: sc0 * sum(a) / n),
// stats[1] <- sum(a) / n.  Blocks add up their elements in double in a fixed order and park the partial sums behind the two results
// (stats[2 ..]: GOPS_LOSS_STATS_FLOATS floats in all, the last one a ticket that must be zero on entry and is left zero); the last block to
// finish adds the partials in block order - deterministic.
__global__ __launch_bounds__(256) void batch_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float gsc, float sc0,
                                                         float* __restrict__ grad, float* __restrict__ stats) {
    __shared__ double red[2][4];
    __shared__ bool last;
    double s0 = 0.0, s1 = 0.0;
    // 8 elements per thread in flight (a load - add loop pays one memory round trip per element); added in index order
    const int stride = gridDim.x * 256;
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 8 * stride) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + q * stride;
            av[q] = i < n ? a[i] : 0.f;
            bv[q] = (b != nullptr && i < n) ? b[i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + q * stride;
            if (i < n) {
                if (b != nullptr) {
                    const float d = av[q] - bv[q];
                    if (grad != nullptr) grad[i] = gsc * d;
                    s0 += (double)(d * d);
                } else {
                    s0 += (double)av[q];
                }
                s1 += (double)av[q];
            }
        }
    }
    block_sum2(s0, s1, red);
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) {
            stats[0] = (float)((b != nullptr ? 1.0 : (double)sc0) * s0 / (double)n);
            stats[1] = (float)(s1 / (double)n);
        }
        return;
    }
    double* part = reinterpret_cast<double*>(stats + 2);               // [LOSS_BLOCKS][2]
    unsigned* ticket = reinterpret_cast<unsigned*>(stats + 2 + 4 * LOSS_BLOCKS);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s0;
        part[2 * blockIdx.x + 1] = s1;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {   // the blocks' partial sums, one per thread, then the same fixed-order tree (a serial walk by one thread cost 15 us)
        __threadfence();
        const volatile double* vp = part;
        const bool has = threadIdx.x < gridDim.x;
        s0 = has ? vp[2 * threadIdx.x] : 0.0;
        s1 = has ? vp[2 * threadIdx.x + 1] : 0.0;
        block_sum2(s0, s1, red);
        if (threadIdx.x == 0) {
            stats[0] = (float)((b != nullptr ? 1.0 : (double)sc0) * s0 / (double)n);
            stats[1] = (float)(s1 / (double)n);
            *ticket = 0u;
        }
    }
}
hipError_t launch_batch_loss(const float* a, const float* b, int n, float gsc, float sc0, float* grad, float* stats, hipStream_t s) {
    hipLaunchKernelGGL(batch_loss_kernel, dim3(n <= LOSS_ONE_BLOCK_MAX ? 1 : LOSS_BLOCKS), dim3(256), 0, s, a, b, n, gsc, sc0, grad, stats);
    return hipGetLastError();
}

// Zero fill as a KERNEL on the caller's stream: inside a captured HIP graph a hipMemsetAsync becomes a memset node,
// which was measured to race with the neighbouring kernel nodes on replay (non-reproducible gradients); a kernel node
// is ordered like every other launch.
__global__ __launch_bounds__(256) void fill_zero_kernel(f32x4* __restrict__ p4, size_t n4, float* __restrict__ tail, int ntail) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p4[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}

hipError_t launch_fill_zero(float* p, size_t n, hipStream_t s) {   // p is 16-byte aligned (workspace carving)
    if (n == 0) return hipSuccess;
    const size_t n4 = n >> 2;
    const unsigned blocks = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (n4 + 255) / 256));
    hipLaunchKernelGGL(fill_zero_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<f32x4*>(p), n4, p + (n4 << 2), (int)(n & 3));
    return hipGetLastError();
}

hipError_t launch_adam(const GopsAdamTensors& T, GopsAdamState* st, double beta1, double beta2, float eps,
                       hipStream_t s) {
    long long nmax = 1;
    for (int i = 0; i < T.n; ++i) nmax = T.numel[i] > nmax ? T.numel[i] : nmax;
    int bx = (int)((nmax + 1023) / 1024);   // >= 4 elements per thread of the largest tensor
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(adam_kernel, dim3(bx, T.n), dim3(256), 0, s, T, st, beta1, beta2, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Single wrapped env-model step (pyth_base_model.py:59-67 contract), one thread per trajectory.
// ---------------------------------------------------------------------------------------------
__global__ void env_step_kernel(const GopsEnv env, int B, const GopsStepIO io, float pdt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int O = env.obs_dim, A = env.act_dim;
    const bool data = env.data_env != 0;   // the DATA environment's step (include/gops_hip.h: GopsEnv.data_env)
    float u[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < A; ++a) u[a] = wrap_action(env, a, io.action[(size_t)b * A + a]);
    const bool dn = !data && !env.no_mask_at_done && io.done != nullptr && io.done[b] != 0.f;   // (no MaskAtDoneModel: done flags ignored)
    const int nrep = (env.repeat_num > 1 && !data) ? env.repeat_num : 1;
    const bool last_only = nrep > 1 && env.repeat_last_reward != 0;
    float r = 0.f;
    bool done_m = false;
    const float* ob = io.obs + (size_t)b * O;
    float* nob = io.next_obs + (size_t)b * O;
    if (env.kind == GOPS_ENV_LQ) {
        float x[GOPS_MAX_LQ_STATE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, xn[GOPS_MAX_LQ_STATE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < O; ++i) x[i] = obs_unscale(env, i, ob[i]);
        float rs = 0.f;
        for (int rep = 0; rep < nrep; ++rep) {   // ActionRepeatModel: sub-steps with the initial done flag (nrep = 1 otherwise)
            if (rep > 0 && !dn)
                for (int i = 0; i < O; ++i) x[i] = xn[i];
            lq_forward(env, x, u, xn, r);
            rs = last_only ? r : rs + r;
        }
        r = rs;
        for (int i = 0; i < O; ++i) {
            const float v = obs_rescale(env, i, dn ? x[i] : xn[i]);
            nob[i] = (env.clip_obs && !data) ? clampf(v, env.obs_low[i], env.obs_high[i]) : v;
            // data env (lq_base.py:224-231, 236-239): done when the NEXT state leaves the state bounds
            if (data && env.clip_obs && (xn[i] > env.obs_high[i] || xn[i] < env.obs_low[i])) done_m = true;   // clip_obs: bounds are finite
        }
        if (data && done_m) r -= 100.f;
    } else if (env.kind == GOPS_ENV_CARTPOLE || env.kind == GOPS_ENV_PENDULUM) {
        const int NS = env.kind == GOPS_ENV_CARTPOLE ? 4 : 3;
        float x[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < NS; ++i) x[i] = obs_unscale(env, i, ob[i]);
        float rs = 0.f;
        for (int rep = 0; rep < nrep; ++rep) {
            if (rep > 0 && !dn)
                for (int i = 0; i < NS; ++i) x[i] = xn[i];
            if (env.kind == GOPS_ENV_CARTPOLE) {
                cart_forward(cart_const(), x, u[0], xn, r, done_m);
            } else {
                PendStep w;
                pend_forward(x, u[0], xn, r, w);
            }
            rs = last_only ? r : rs + r;
        }
        r = rs;
        // data env (env_gym/gym_cartpoleconti.py:102-137): the same physics, reward 1 also for the step that ends the
        // episode, no observation clip
        if (data && env.kind == GOPS_ENV_CARTPOLE) r = 1.f;
        for (int i = 0; i < NS; ++i) {
            const float v = obs_rescale(env, i, dn ? x[i] : xn[i]);
            nob[i] = (env.clip_obs && !data) ? clampf(v, env.obs_low[i], env.obs_high[i]) : v;
        }
    } else if (env.kind == GOPS_ENV_IDPENDULUM) {   // data env == model (pyth_idpendulum.py:71-87 calls the model's Dynamics)
        const IdpConst IC = idp_const();
        float s[6], sn[6], s0[6];
        for (int i = 0; i < 6; ++i) s0[i] = s[i] = obs_unscale(env, i, ob[i]);
        float rs = 0.f;
        for (int rep = 0; rep < nrep; ++rep) {
            IdpSub w;
            for (int k = 0; k < 5; ++k) {   // same arithmetic as the rollout kernels (one sincosf pair, then rotations)
                if (k == 0) idp_substep<true>(IC, s, 500.f * u[0], 0.002f, sn, w);
                else idp_substep<false>(IC, s, 500.f * u[0], 0.002f, sn, w);
                idp_advance_trig(s, 0.002f, w, w);
                for (int i = 0; i < 6; ++i) s[i] = sn[i];
            }
            r = idp_reward(s, u[0]);
            rs = last_only ? r : rs + r;
            done_m = idp_done(IC, s);
        }
        r = rs;
        for (int i = 0; i < 6; ++i) nob[i] = (dn && !env.scale_obs) ? ob[i] : obs_rescale(env, i, dn ? s0[i] : s[i]);
    } else if (env.kind == GOPS_ENV_MOBILEROBOT) {
        const MobConst MC = mob_const();
        float x[MOB_OBS], xn[MOB_OBS], c;
        for (int i = 0; i < MOB_OBS; ++i) x[i] = ob[i];
        const float nv = io.noise != nullptr ? io.noise[(size_t)b * 2 + 0] : 0.f;
        const float nw = io.noise != nullptr ? io.noise[(size_t)b * 2 + 1] : 0.f;
        MobStep w;
        if (data) mob_forward<true>(MC, x, u[0], u[1], nv, nw, xn, r, c, done_m, w);   // pyth_mobilerobot.py:108-152: headings clipped to +-pi
        else mob_forward(MC, x, u[0], u[1], nv, nw, xn, r, c, done_m, w);
        io.constraint[b] = c;   // of the model's new state, whatever `done` says
        for (int i = 0; i < MOB_OBS; ++i) {
            const float v = dn ? x[i] : xn[i];
            nob[i] = (env.clip_obs && !data) ? clampf(v, env.obs_low[i], env.obs_high[i]) : v;   // (the data env clips nothing)
        }
    } else if (env.kind == GOPS_ENV_VEH2DOF) {
        const Veh2Const C2 = veh2_const();
        const int P = env.pre_horizon;
        float s[4], sn[4], o4[4];
        for (int i = 0; i < 4; ++i) { s[i] = io.state[(size_t)b * 4 + i]; o4[i] = ob[i]; }
        r = veh2_reward(o4, u[0]);
        float sphi, cphi;
        sincosf(s[1], &sphi, &cphi);
        veh2_f_xu(C2, s, u[0], sphi, cphi, sn);
        const float nt = RADD(io.ref_time[b], 0.1f);
        const float pn = io.path_num[b], un = io.u_num[b];
        const f32x4 newp = io.ref_appended != nullptr ? reinterpret_cast<const f32x4*>(io.ref_appended)[b]
                                                      : ref_point_ids(env.ref_c, RADD(nt, pdt), pn, un);
        const float* rin = io.ref_points + (size_t)b * (P + 1) * 2;
        float* rout = io.next_ref_points + (size_t)b * (P + 1) * 2;
        for (int i = 0; i < P; ++i) { rout[2 * i] = rin[2 * (i + 1)]; rout[2 * i + 1] = rin[2 * (i + 1) + 1]; }
        rout[2 * P] = newp[1]; rout[2 * P + 1] = newp[2];
        const float o0 = sn[0] - rout[0], o1 = sn[1] - rout[1];
        done_m = (fabsf(o0) > 2.f) || (fabsf(o1) > 3.14159265358979323846f);
        if (data && done_m) r -= 100.f;   // data env (pyth_veh2dofconti.py:179-219): the model's step, -100 at done
        if (dn) {
            for (int i = 0; i < O; ++i) nob[i] = ob[i];
        } else {
            nob[0] = o0; nob[1] = o1; nob[2] = sn[2]; nob[3] = sn[3];
            for (int i = 1; i <= P; ++i) nob[3 + i] = sn[0] - rout[2 * i];
        }
        for (int i = 0; i < 4; ++i) io.next_state[(size_t)b * 4 + i] = sn[i];
        io.next_ref_time[b] = nt;
        if (env.cstr_err && io.constraint != nullptr) io.constraint[b] = fabsf(o4[0]) - env.err_tol[0];   // of the observation it was called with
    } else if (env.kind == GOPS_ENV_VEH3DOFCONTI || env.kind == GOPS_ENV_VEH3DOF_SURR) {
        const bool surr = env.kind == GOPS_ENV_VEH3DOF_SURR;
        const VehConst VC = veh_const();
        const int P = env.pre_horizon;
        float s[6], sn[6], o6[6];
        for (int i = 0; i < 6; ++i) { s[i] = io.state[(size_t)b * 6 + i]; o6[i] = ob[i]; }
        VehStep w;
        sincosf(s[2], &w.sphi, &w.cphi);
        veh_f_xu(VC, s, u[0], u[1], sn, w);
        r = surr ? veh_reward_w(env.reward_w, o6, u[0], u[1]) : veh_reward(o6, u[0], u[1]);
        float pen_c = 0.f;
        if (surr && env.surr_penalty) {   // collision penalty on the current pose / current surrounding vehicle
            const float* s5 = io.surr_state + (size_t)b * env.n_surr * 5;
            const f32x4 cur = {s5[0], s5[1], s5[2], s5[3]};
            SurrCstr sc0;
            surr_constraint<false>(env, s[0], s[1], w.sphi, w.cphi, &cur, sc0);
            float dummy;
            pen_c = sc0.c[0];
            r -= surr_penalty(pen_c, dummy);
        }
        const float nt = RADD(io.ref_time[b], 0.1f);
        const float pn = io.path_num[b], un = io.u_num[b];
        const f32x4 newp = io.ref_appended != nullptr ? reinterpret_cast<const f32x4*>(io.ref_appended)[b]
                                                      : ref_point_ids(env.ref_c, RADD(nt, pdt), pn, un);
        const f32x4* rin = reinterpret_cast<const f32x4*>(io.ref_points) + (size_t)b * (P + 1);
        f32x4* rout = reinterpret_cast<f32x4*>(io.next_ref_points) + (size_t)b * (P + 1);
        float cn, snn;
        sincosf(-sn[2], &snn, &cn);
        for (int j = 0; j <= P; ++j) {
            const f32x4 rp = (j < P) ? rin[j + 1] : newp;
            rout[j] = rp;
            const float dx = rp[0] - sn[0], dy = rp[1] - sn[1];
            const float xtf = dx * cn - dy * snn, ytf = dx * snn + dy * cn;
            const float ptf = angle_normalize(rp[2] - sn[2]), utf = rp[3] - sn[3];
            if (j == 0) {
                if (data)   // pyth_veh3dofconti.py:263-271: world-frame offsets to the first reference point, 5 / 2 / pi
                    done_m = (fabsf(dx) > 5.f) || (fabsf(dy) > 2.f) || (fabsf(ptf) > 3.14159265358979323846f);
                else
                    done_m = (fabsf(xtf) > 10.f) || (fabsf(ytf) > 10.f) || (fabsf(ptf) > 3.14159265358979323846f);
                if (!dn) { nob[0] = xtf; nob[1] = ytf; nob[2] = ptf; nob[3] = utf; nob[4] = sn[4]; nob[5] = sn[5]; }
            } else if (!dn) {
                float* d = nob + 6 + 4 * (j - 1);
                d[0] = xtf; d[1] = ytf; d[2] = ptf; d[3] = utf;
            }
        }
        if (data && done_m) r -= 100.f;   // :224-226
        if (surr) {   // pyth_veh3dofconti_surrcstr_model.py:84-95: surrounding vehicles step, relative obs, constraint (unmasked)
            f32x4 pts[GOPS_MAX_SURR];
            for (int i = 0; i < env.n_surr; ++i) {
                const float* s5 = io.surr_state + ((size_t)b * env.n_surr + i) * 5;
                const f32x4 cur = {s5[0], s5[1], s5[2], s5[3]};
                pts[i] = surr_next(cur, s5[4]);
                float* d5 = io.next_surr_state + ((size_t)b * env.n_surr + i) * 5;
                d5[0] = pts[i][0]; d5[1] = pts[i][1]; d5[2] = pts[i][2]; d5[3] = pts[i][3]; d5[4] = s5[4];
                if (!dn) {
                    float* d = nob + 6 + 4 * P + 4 * i;
                    if (env.surr_penalty) {   // ego frame of the CURRENT state
                        const float dx = pts[i][0] - s[0], dy = pts[i][1] - s[1];
                        d[0] = dx * w.cphi + dy * w.sphi; d[1] = -dx * w.sphi + dy * w.cphi;
                        d[2] = angle_normalize(pts[i][2] - s[2]); d[3] = pts[i][3] - s[3];
                    } else {
                        d[0] = pts[i][0] - sn[0]; d[1] = pts[i][1] - sn[1]; d[2] = pts[i][2] - sn[2]; d[3] = pts[i][3] - sn[3];
                    }
                }
            }
            SurrCstr sc;
            float sp, cp;
            sincosf(sn[2], &sp, &cp);
            surr_constraint<false>(env, sn[0], sn[1], sp, cp, pts, sc);
            if (env.surr_penalty) sc.c[0] = pen_c;   // info["constraint"] is filled before the info dict is updated (:131-139)
            if (env.cstr_err) { sc.c[0] = fabsf(o6[1]) - env.err_tol[0]; sc.c[1] = fabsf(o6[3]) - env.err_tol[1]; }   // current obs
            for (int k = 0; k < env.n_constraint; ++k) io.constraint[(size_t)b * env.n_constraint + k] = sc.c[k];
            if (env.surr_penalty) done_m = false;
        }
        if (dn) for (int i = 0; i < O; ++i) nob[i] = ob[i];
        for (int i = 0; i < 6; ++i) io.next_state[(size_t)b * 6 + i] = sn[i];
        io.next_ref_time[b] = nt;
    }
    float rr = dn ? 0.f : r;
    if (env.shaping) rr = (rr + env.reward_shift) * env.reward_scale;
    io.reward[b] = rr;
    io.next_done[b] = (dn || done_m) ? 1.f : 0.f;
}

// model.get_constraint(obs, info), one thread per row (gops_env_constraint)
__global__ void env_constraint_kernel(const GopsEnv env, int B, const GopsStepIO io) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nc = env.n_constraint;
    if (env.cstr_err) {
        const float* ob = io.obs + (size_t)b * env.obs_dim;
        if (env.kind == GOPS_ENV_VEH2DOF) {
            io.constraint[b] = fabsf(ob[0]) - env.err_tol[0];
        } else {
            io.constraint[(size_t)b * 2 + 0] = fabsf(ob[1]) - env.err_tol[0];
            io.constraint[(size_t)b * 2 + 1] = fabsf(ob[3]) - env.err_tol[1];
        }
        return;
    }
    const float* st = io.state + (size_t)b * 6;
    f32x4 pts[GOPS_MAX_SURR];
    for (int i = 0; i < GOPS_MAX_SURR; ++i) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pts[i] = z;
        if (i < env.n_surr) {
            const float* sv = io.surr_state + ((size_t)b * env.n_surr + i) * 5;
            pts[i][0] = sv[0]; pts[i][1] = sv[1]; pts[i][2] = sv[2]; pts[i][3] = sv[3];
        }
    }
    float sp, cp;
    sincosf(st[2], &sp, &cp);
    SurrCstr sc;
    surr_constraint<false>(env, st[0], st[1], sp, cp, pts, sc);
    for (int k = 0; k < nc; ++k) io.constraint[(size_t)b * nc + k] = sc.c[k];
}

hipError_t launch_env_constraint(const GopsEnv& env, int B, const GopsStepIO& io, hipStream_t s) {
    hipLaunchKernelGGL(env_constraint_kernel, dim3((B + 127) / 128), dim3(128), 0, s, env, B, io);
    return hipGetLastError();
}

hipError_t launch_env_step(const GopsEnv& env, int B, const GopsStepIO& io, float pdt, hipStream_t s) {
    GopsEnv padded = env;
    lq_pad_env(padded);
    hipLaunchKernelGGL(env_step_kernel, dim3((B + 127) / 128), dim3(128), 0, s, padded, B, io, pdt);
    return hipGetLastError();
}
