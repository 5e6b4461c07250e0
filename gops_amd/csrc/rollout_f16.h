// Half-precision MLP phases of the rollout kernels (GOPS_DTYPE_F16, BASELINE.json configs[4]
// "fp16 MFMA MLP path"): v_mfma_f32_16x16x32_f16 with fp32 accumulation.
//
// Every contraction is issued TRANSPOSED: the MFMA's M index is the output feature, its N index
// the trajectory of the tile, its K index the input feature:
//     A (weights)      lane l supplies W[feature(l & 15)][32c + 8(l >> 4) + 0..7]   one 16-byte load
//     B (activations)  lane l supplies act[m = l & 15][32c + 8(l >> 4) + 0..7]      one ds_read_b128
//     D                lane l holds out[m = l & 15][features 4(l >> 4) + 0..3]
// so a lane's results are consecutive features of ONE trajectory: they go to the row-major LDS
// tile / stash as 16-byte vectors and are directly the next layer's B operand.  A wave owns a "quad"
// of four n-tiles = 64 output features; the weight packing permutes the rows inside a quad (tile j,
// row 4g + r  <->  feature 64q + 16g + 4j + r) so that lane (m, g) holds the 16 CONSECUTIVE features
// 64q + 16g .. + 15 of trajectory m: two 16-byte stores per quad, 128 contiguous bytes per stash row
// and wave.
#pragma once
#include "common.h"

#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f16x8 ld8h(const _Float16* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ f16x8 ld8h(const GLOBAL_AS _Float16* p) { return *(const GLOBAL_AS f16x8*)p; }
// float -> half that saturates at +-65504 instead of producing inf (backward deltas of a diverging trajectory)
__device__ __forceinline__ _Float16 sat_h(float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
__device__ __forceinline__ f16x8 zero8h() {
    const f16x8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f,
                     (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    return z;
}

// (gelu_pair_h: common.h)

// acc[j] (n-tile j of quad q, see the header) += W_quad * act^T over kch chunks of 32 inputs.  The
// weight fragments stream from L2 through a ring of PF chunks (L2 latency >> the 4 MFMAs of a chunk).  Measured
// at B = 65536 (cfg5): a deeper ring (4) or register-stationary fragments (16 K0 + 128 registers per lane for a
// 256-256 policy) cost more in occupancy than they save - the step is bound by dependent VALU / LDS / scalar
// latency, which 4 resident workgroups per CU hide better than 2 or 3.
#ifndef GOPS_F16_PF
#define GOPS_F16_PF 2   // chunks of weight fragments in flight per wave (16 registers each)
#endif
__device__ __forceinline__ void gemm_quad_h(const _Float16* act, int ld, int kch, const f16x8* Wp, int q,
                                            int lane, f32x4 (&acc)[4]) {
    constexpr int PF = GOPS_F16_PF;
    const GLOBAL_AS f16x8* wb = gptr(Wp) + (size_t)q * 4 * kch * 64 + lane;
    const _Float16* brow = act + (lane & 15) * ld + 8 * (lane >> 4);
    f16x8 ring[PF][4];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int cd = d < kch ? d : kch - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) ring[d][j] = wb[((size_t)j * kch + cd) * 64];
    }
    for (int c0 = 0; c0 < kch; c0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int c = c0 + d;
            if (c < kch) {
                const f16x8 b = ld8h(brow + 32 * c);
                f16x8 a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = ring[d][j];
                const int cn = (c + PF < kch) ? c + PF : kch - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) ring[d][j] = wb[((size_t)j * kch + cn) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = MFMA_F16(a[j], b, acc[j]);
            }
        }
    }
}

// fp32 observation tile xs [TB][ldx] (columns < kp valid, the rest of a row zero) -> half tile x16
// [TB][ld16] with kp32 columns (zero padded) and, when g16 is non-null, the stash rows g16[(row0+m)*kp32 ..].
__device__ __forceinline__ void convert_x_h(const float* xs, int ldx, int kp, int kp32, _Float16* x16, int ld16,
                                            _Float16* g16, size_t row0, int tid) {
    const int upr = kp32 >> 3;   // 8-column units per row
    for (int idx = tid; idx < TB * upr; idx += NTHREADS) {
        const int m = idx / upr, c = (idx - m * upr) << 3;
        f16x8 v = zero8h();
        if (c < kp) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xs + m * ldx + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(xs + m * ldx + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (_Float16)a[e]; v[4 + e] = (_Float16)b[e]; }
        }
        *reinterpret_cast<f16x8*>(x16 + m * ld16 + c) = v;
        if (g16 != nullptr) __builtin_nontemporal_store(v, gptr(reinterpret_cast<f16x8*>(g16 + (row0 + m) * kp32 + c)));
    }
}

// Hidden layers of `M` on the half tile x16 (TB x kp32[0]).  Returns the LDS buffer holding the last
// hidden activation.  stash_h[j] (and, for GELU, stash_g[j] <- act'(z)) receive rows row0 .. of the
// tile for m < stash_rows when non-null.  s_bias: LDS fp32 biases, row j at s_bias + j * ldb.
__device__ __forceinline__ const _Float16* mlp_hidden_forward_h(const MlpDev& M, const _Float16* x16, int ldx16,
                                                                _Float16* ha, _Float16* hb, int ld16, int tid,
                                                                const float* s_bias, int ldb, float* const* stash_h,
                                                                float* const* stash_g, size_t row0, int stash_rows) {
    const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
    const int L = M.nl - 1;
    const _Float16* cur = x16;
    int ldc = ldx16;
    _Float16* out = ha;
    for (int j = 0; j < L; ++j) {
        const int N = M.dims[j + 1], kch = M.kp32[j] >> 5, nquads = N >> 6;
        const float* bias = s_bias + j * ldb;
        _Float16* hrow = (stash_h != nullptr && m < stash_rows) ? reinterpret_cast<_Float16*>(stash_h[j + 1]) + (row0 + m) * N : nullptr;
        _Float16* grow = (stash_g != nullptr && m < stash_rows && M.act == GOPS_ACT_GELU)
                             ? reinterpret_cast<_Float16*>(stash_g[j + 1]) + (row0 + m) * N : nullptr;
        for (int q = wave; q < nquads; q += 4) {
            f32x4 acc[4] = {};
            gemm_quad_h(cur, ldc, kch, M.wph[j], q, lane, acc);
            const int f0 = 64 * q + 16 * g;
            f32x4 bv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) bv[jj] = *reinterpret_cast<const f32x4*>(bias + f0 + 4 * jj);
            act_dispatch(M.act, [&]<int ACT>() {
                f16x8 o[2], gd[2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 4 * jj + r;
                        const float z = acc[jj][r] + bv[jj][r];
                        float h, dh = 0.f;
                        if (ACT == GOPS_ACT_GELU) gelu_pair_h(z, h, dh);
                        else h = act_fwd_t<ACT>(z);
                        o[e >> 3][e & 7] = (_Float16)h;
                        gd[e >> 3][e & 7] = (_Float16)dh;
                    }
                *reinterpret_cast<f16x8*>(out + m * ld16 + f0) = o[0];
                *reinterpret_cast<f16x8*>(out + m * ld16 + f0 + 8) = o[1];
                if (hrow != nullptr) {
                    __builtin_nontemporal_store(o[0], gptr(reinterpret_cast<f16x8*>(hrow + f0)));
                    __builtin_nontemporal_store(o[1], gptr(reinterpret_cast<f16x8*>(hrow + f0 + 8)));
                }
                if (ACT == GOPS_ACT_GELU && grow != nullptr) {
                    __builtin_nontemporal_store(gd[0], gptr(reinterpret_cast<f16x8*>(grow + f0)));
                    __builtin_nontemporal_store(gd[1], gptr(reinterpret_cast<f16x8*>(grow + f0 + 8)));
                }
            });
        }
        __syncthreads();
        cur = out;
        ldc = ld16;
        out = (out == ha) ? hb : ha;
    }
    return cur;
}

// Output layer (width A <= 4) on the VALU in fp32 from the half activations: thread (hm = tid >> 4,
// hp = tid & 15) strides over k; y[a] is valid in every lane of the 16-lane group.  Wo: [A][ldw] fp32.
// ZEROED: Wo / bo are the LDS copies whose rows a >= A are zero-filled - all four outputs are formed unconditionally so that
// the reads of a thread are issued back to back (see mlp_head in rollout_fwd.hip).
template <bool ZEROED, class WP, class BP>
__device__ __forceinline__ void mlp_head_h(WP Wo, int ldw, BP bo, int K, int A, const _Float16* hcur, int ld16,
                                           int tid, float (&y)[GOPS_MAX_ACT]) {
    const int hm = tid >> 4, hp = tid & 15;
#pragma unroll
    for (int a = 0; a < GOPS_MAX_ACT; ++a) y[a] = 0.f;
#pragma unroll 2
    for (int k = 8 * hp; k < K; k += 128) {
        const f16x8 hv = ld8h(hcur + hm * ld16 + k);
        if constexpr (ZEROED) {
            f32x4 w0[GOPS_MAX_ACT], w1[GOPS_MAX_ACT];
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a) { w0[a] = ld4(Wo + a * ldw + k); w1[a] = ld4(Wo + a * ldw + k + 4); }
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                y[a] += ((float)hv[0] * w0[a][0] + (float)hv[1] * w0[a][1] + (float)hv[2] * w0[a][2] + (float)hv[3] * w0[a][3]) +
                        ((float)hv[4] * w1[a][0] + (float)hv[5] * w1[a][1] + (float)hv[6] * w1[a][2] + (float)hv[7] * w1[a][3]);
        } else {
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                if (a < A) {
                    const f32x4 w0 = ld4(Wo + a * ldw + k), w1 = ld4(Wo + a * ldw + k + 4);
                    y[a] += ((float)hv[0] * w0[0] + (float)hv[1] * w0[1] + (float)hv[2] * w0[2] + (float)hv[3] * w0[3]) +
                            ((float)hv[4] * w1[0] + (float)hv[5] * w1[1] + (float)hv[6] * w1[2] + (float)hv[7] * w1[3]);
                }
        }
    }
    if constexpr (ZEROED) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&bo[0]);
#pragma unroll
        for (int a = 0; a < GOPS_MAX_ACT; ++a) y[a] = row16_sum(y[a]) + bv[a];
    } else {
#pragma unroll
        for (int a = 0; a < GOPS_MAX_ACT; ++a) {
            y[a] = row16_sum(y[a]);
            if (a < A) y[a] += bo[a];
        }
    }
}

// Backward through one MLP in half precision.  delta_y (fp32, in the launch's scaled units) in s_gy[TB][4]
// -> hidden deltas (half; stashed to st_d[j] when st_d is non-null) and, if want_gx, G[m][n] += (delta_1
// W_0)[m][n] for n < ncols in fp32.  act' comes from the stash: act'(z) itself for GELU (st_z), else derived
// from the stashed activation (st_h).
template <class WP, class Hook>
__device__ __forceinline__ void mlp_backward_h(const MlpDev& M, WP Wo, int ldw, const float* s_gy, _Float16* da,
                                               _Float16* db, int ld16, float* G, int ldg, int tid,
                                               float* const* st_h, float* const* st_z, float* const* st_d,
                                               float* stash_dy, size_t row0, int nvalid, bool want_gx, int ncols,
                                               Hook&& after_head) {
    const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
    const int L = M.nl - 1, A = M.dims[M.nl];
    const bool gelu = (M.act == GOPS_ACT_GELU);
    {   // head: delta_L[m][k] = (sum_a gy[m][a] Wo[a][k]) * act'_L[m][k]
        const int K = M.dims[L];
        const int hm = tid >> 4, hp = tid & 15;
        float gy[GOPS_MAX_ACT];
#pragma unroll
        for (int a = 0; a < GOPS_MAX_ACT; ++a) gy[a] = (a < A) ? s_gy[hm * 4 + a] : 0.f;
        const GLOBAL_AS _Float16* src = gptr(reinterpret_cast<const _Float16*>(gelu ? st_z[L] : st_h[L]) + (row0 + hm) * K);
        _Float16* dst = (st_d != nullptr) ? reinterpret_cast<_Float16*>(st_d[L]) + (row0 + hm) * K : nullptr;
        // the act' operands of the first 512 columns are requested before the first delta store (a load inside the loop waits for
        // its own HBM round trip: the compiler does not move it across a store it cannot prove disjoint)
        f16x8 hvs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 8 * hp + 128 * i;
            hvs[i] = zero8h();
            if (k < K && hm < nvalid) hvs[i] = ld8h(src + k);
        }
        act_dispatch(M.act, [&]<int ACT>() {
            auto column_block = [&](int k, const f16x8 hv) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a)
                    if (a < A) {
                        const f32x4 w0 = ld4(Wo + a * ldw + k), w1 = ld4(Wo + a * ldw + k + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc[e] += gy[a] * w0[e]; acc[4 + e] += gy[a] * w1[e]; }
                    }
                f16x8 dv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float hf = (float)hv[e];
                    const float d = (ACT == GOPS_ACT_GELU) ? hf : act_bwd_t<ACT>(hf, hf);
                    dv[e] = sat_h((hm < nvalid) ? acc[e] * d : 0.f);
                }
                *reinterpret_cast<f16x8*>(da + hm * ld16 + k) = dv;
                if (dst != nullptr) __builtin_nontemporal_store(dv, gptr(reinterpret_cast<f16x8*>(dst + k)));
            };
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (8 * hp + 128 * i < K) column_block(8 * hp + 128 * i, hvs[i]);
            for (int k = 8 * hp + 512; k < K; k += 128) column_block(k, (hm < nvalid) ? ld8h(src + k) : zero8h());
        });
        if (stash_dy != nullptr && tid < TB) {
            f32x4 v = {s_gy[tid * 4 + 0], s_gy[tid * 4 + 1], s_gy[tid * 4 + 2], s_gy[tid * 4 + 3]};
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                if (a >= A || tid >= nvalid) v[a] = 0.f;
            *gptr(reinterpret_cast<f32x4*>(stash_dy + (row0 + tid) * 4)) = v;
        }
    }
    __syncthreads();
    after_head();
    _Float16* cur = da;
    _Float16* out = db;
    for (int j = L - 1; j >= 1; --j) {   // delta_j = (delta_{j+1} W_j) * act'_j
        const int N = M.dims[j], kch = M.dims[j + 1] >> 5, nquads = N >> 6;
        const GLOBAL_AS _Float16* src = gptr(reinterpret_cast<const _Float16*>(gelu ? st_z[j] : st_h[j]) + (row0 + m) * N);
        _Float16* dst = (st_d != nullptr) ? reinterpret_cast<_Float16*>(st_d[j]) + (row0 + m) * N : nullptr;
        for (int q = wave; q < nquads; q += 4) {
            const int f0 = 64 * q + 16 * g;
            f16x8 hv[2] = {zero8h(), zero8h()};
            if (m < nvalid) { hv[0] = ld8h(src + f0); hv[1] = ld8h(src + f0 + 8); }   // in flight during the GEMM
            f32x4 acc[4] = {};
            gemm_quad_h(cur, ld16, kch, M.wpth[j], q, lane, acc);
            act_dispatch(M.act, [&]<int ACT>() {
                f16x8 o[2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 4 * jj + r;
                        const float hf = (float)hv[e >> 3][e & 7];
                        const float d = (ACT == GOPS_ACT_GELU) ? hf : act_bwd_t<ACT>(hf, hf);
                        o[e >> 3][e & 7] = sat_h((m < nvalid) ? acc[jj][r] * d : 0.f);
                    }
                *reinterpret_cast<f16x8*>(out + m * ld16 + f0) = o[0];
                *reinterpret_cast<f16x8*>(out + m * ld16 + f0 + 8) = o[1];
                if (dst != nullptr) {
                    __builtin_nontemporal_store(o[0], gptr(reinterpret_cast<f16x8*>(dst + f0)));
                    __builtin_nontemporal_store(o[1], gptr(reinterpret_cast<f16x8*>(dst + f0 + 8)));
                }
            });
        }
        __syncthreads();
        _Float16* tmp = cur; cur = out; out = tmp;
    }
    if (want_gx) {   // g_x = delta_1 W_0: plain 16-feature tiles over the (16-padded) inputs, fp32 into G
        const int kch = M.dims[1] >> 5, nt_tot = M.kp[0] >> 4;
        const _Float16* brow = cur + m * ld16 + 8 * g;
        for (int nt = wave; nt < nt_tot; nt += 4) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const GLOBAL_AS f16x8* wb = gptr(M.wpth[0]) + (size_t)nt * kch * 64 + lane;
            for (int c = 0; c < kch; c += 2) {   // kch is even (hidden widths are multiples of 64)
                const f16x8 a0 = wb[(size_t)c * 64], a1 = wb[(size_t)(c + 1) * 64];
                acc0 = MFMA_F16(a0, ld8h(brow + 32 * c), acc0);
                acc1 = MFMA_F16(a1, ld8h(brow + 32 * (c + 1)), acc1);
            }
            const int f = 16 * nt + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (f + r < ncols) G[m * ldg + f + r] += acc0[r] + acc1[r];
        }
    }
}
