// Forward horizon rollout: one workgroup (4 wavefronts) owns a tile of 16 trajectories and walks
// all H timesteps without leaving the CU.  Per step: policy MLP (hidden layers on
// v_mfma_f32_16x16x4_f32 with the activations staged in LDS and the weights streamed from L2 in
// MFMA-fragment order), tanh head + wrapper chain, env model step, masked/shaped reward into the
// discounted return.  Replaces the Python loop of fhadp.py:117-120 / infadp.py:171-180,198-208.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#define GOPS_STREAMB_EXACT_REFILL   // (this translation unit only: the backward kernels' register allocation degrades with it)
#include "common.h"
#ifndef GOPS_IDP_FWD_UNROLL
#define GOPS_IDP_FWD_UNROLL 1   // sub-steps 2 .. 5 of pyth_idpendulum unrolled (cfg2 forward 213 -> 208 us)
#endif
#include "env_models.h"
#include "rollout_f16.h"

// Hidden layers of `M` applied to the LDS tile `in` (TB x kp[0], leading dim ld_in).  Returns the
// LDS buffer that holds the last hidden activation.  When stash_h is non-null the activations
// (and GELU pre-activations) of the tile are written to stash_h[j] + row0 * dims[j].  Layers 0 / 1
// use the register-stationary fragments W0 / W1 when those are StatW, else stream from L2.
// narrow: LDS image of the packed weights of ALL hidden layers, layer after layer (common.h gemm_layer_lds), or null.
template <class W0T, class W1T>
__device__ __forceinline__ float* mlp_hidden_forward(const MlpDev& M, const W0T& W0, const W1T& W1,
                                                     const float* in, int ld_in, float* ha, float* hb,
                                                     int ldh, int tid, const float* s_bias,
                                                     float* const* stash_h, float* const* stash_z,
                                                     size_t row0, DbgClock& dbg, const f32x4* narrow = nullptr) {
    const int lane = tid & 63;
    const int L = M.nl - 1;
    const float* cur = in;
    int ldc = ld_in;
    float* out = ha;
    for (int j = 0; j < L; ++j) {
        const int N = M.dims[j + 1], kch = M.kp[j] >> 4, nt_tot = N >> 4;
        const float* bias = s_bias + j * ldh;   // LDS copy of the layer's bias
        const bool save_z = (stash_z != nullptr) && (M.act == GOPS_ACT_GELU);
        float* zrow = save_z ? stash_z[j + 1] + row0 * N : nullptr;
        // The activation tile goes to the FM stash straight from the epilogue registers: a lane's four rows of
        // feature n are one fire-and-forget 16-byte store, a wave's n-tile 1 KiB of contiguous memory.  All 16
        // rows are written (rows past the batch end hold the activations of a zero observation).
        float* hrow = (stash_h != nullptr) ? stash_h[j + 1] + row0 * N : nullptr;
        auto epi = [&]<int CNT>(const f32x4 (&acc)[4], int nt0) {
            DBG_TICK(14)
            act_dispatch(M.act, [&]<int ACT>() {
                float bn[CNT];
#pragma unroll
                for (int q = 0; q < CNT; ++q) bn[q] = bias[((nt0 + q) << 4) + (lane & 15)];
                const int m0 = (lane >> 4) << 2;
#pragma unroll
                for (int q = 0; q < CNT; ++q) {
                    const int n = ((nt0 + q) << 4) + (lane & 15);
                    f32x4 hv, zv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = acc[q][r] + bn[q];
                        float hr, dr = z;
                        if constexpr (ACT == GOPS_ACT_GELU) gelu_pair(z, hr, dr);   // dr: gelu'(z), what the sweep needs
                        else hr = act_fwd_t<ACT>(z);
                        hv[r] = hr; zv[r] = dr;
                        out[(m0 + r) * ldh + n] = hv[r];
                    }
                    if (hrow != nullptr) __builtin_nontemporal_store(hv, gptr(reinterpret_cast<f32x4*>(hrow + n * 16 + m0)));
                    if (ACT == GOPS_ACT_GELU && save_z) __builtin_nontemporal_store(zv, gptr(reinterpret_cast<f32x4*>(zrow + n * 16 + m0)));
                }
            });
        };
        bool done = false;
        if constexpr (!std::is_same<W0T, NoW>::value) {
            if (j == 0) { gemm_layer_stat(cur, ldc, W0, nt_tot, tid, epi, kch, M.wp[0]); done = true; }
        }
        if constexpr (!std::is_same<W1T, NoW>::value) {
            if (j == 1) { gemm_layer_stat(cur, ldc, W1, nt_tot, tid, epi); done = true; }
        }
        if (!done) {
            if (narrow != nullptr) {
                gemm_layer_lds(cur, ldc, kch, nt_tot, narrow, tid, epi);
                narrow += kch * nt_tot * 64;
            } else {
                gemm_layer(cur, ldc, kch, nt_tot, M.wp[j], tid, epi);
            }
        }
        DBG_TICK(8 + 3 * (j & 1))
        __syncthreads();
        DBG_TICK(9 + 3 * (j & 1))
        DBG_TICK(10 + 3 * (j & 1))
        cur = out;
        ldc = ldh;
        out = (out == ha) ? hb : ha;
    }
    return const_cast<float*>(cur);
}

// obs -> 64 -> 64 -> act policies (the shape of nearly every script under the reference's example_train/) on the narrow launches:
// the two hidden layers written out with their shapes as compile-time constants - one n-tile per wave, no layer loop, and nothing of
// the parameter block read inside the step loop (Hot64: pinned once per launch).  The generic layer loop spends ~1 k cycles per layer
// on its own set-up before the first MFMA (knock-outs of the narrow kernel); same products in the same order.
struct Hot64 {
    int kch0, act;              // 16-wide k-chunks of the (padded) policy input; activation kind
    float *h1, *h2, *z1, *z2;   // stash tensors of the two hidden layers (null: nothing is stashed)
    const f32x4 *w0, *w1;       // LDS images of the packed weights (narrow_fill)
};
__device__ __forceinline__ float* mlp_hidden_forward_n64(const Hot64& hn, const float* in, int ld_in, float* ha, float* hb, int ldh, int tid,
                                                         const float* s_bias, size_t row0, DbgClock& dbg) {
    const int lane = tid & 63, wave = tid >> 6;
    const int n = (wave << 4) + (lane & 15), m0 = (lane >> 4) << 2;
    auto layer = [&]<int J>(const float* A, int lda, float* out) {
        const int kch = (J == 0) ? hn.kch0 : 4;
        const float* arow = A + (lane & 15) * lda + 4 * (lane >> 4);
        const f32x4* wl = (J == 0 ? hn.w0 : hn.w1) + (wave * kch * 64 + lane);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (J == 1) {   // K = 64: the eight fragment reads up front, then the 16 MFMAs
            f32x4 a[4], b[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { a[c] = *reinterpret_cast<const f32x4*>(arow + 16 * c); b[c] = wl[c * 64]; }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], b[c][i], acc, 0, 0, 0);
        } else {
            for (int c = 0; c < kch; ++c) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * c), b = wl[c * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
            }
        }
        DBG_TICK(14)
        const float bn = s_bias[J * ldh + n];
        float* hs = (J == 0) ? hn.h1 : hn.h2;
        float* zs = (J == 0) ? hn.z1 : hn.z2;
        float* hrow = (hs != nullptr) ? hs + row0 * 64 : nullptr;
        float* zrow = (zs != nullptr) ? zs + row0 * 64 : nullptr;
        act_dispatch(hn.act, [&]<int ACT>() {
            f32x4 hv, zv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[r] + bn;
                float hr, dr = z;
                if constexpr (ACT == GOPS_ACT_GELU) gelu_pair(z, hr, dr);   // dr: gelu'(z), what the sweep needs
                else hr = act_fwd_t<ACT>(z);
                hv[r] = hr; zv[r] = dr;
                out[(m0 + r) * ldh + n] = hr;
            }
            if (hrow != nullptr) __builtin_nontemporal_store(hv, gptr(reinterpret_cast<f32x4*>(hrow + n * 16 + m0)));
            if (ACT == GOPS_ACT_GELU && zrow != nullptr) __builtin_nontemporal_store(zv, gptr(reinterpret_cast<f32x4*>(zrow + n * 16 + m0)));
        });
        DBG_TICK(8 + 3 * J)
        __syncthreads();
        DBG_TICK(9 + 3 * J)
    };
    layer.template operator()<0>(in, ld_in, ha);
    layer.template operator()<1>(ha, ldh, hb);
    return hb;
}

// Output layer (width A <= 4) on the VALU: thread (hm = tid>>4, hp = tid&15) strides over k.
// Wo is [4][ldw] in LDS with the rows a >= A ZERO-FILLED (staged once per launch) and bo likewise: all four
// outputs are formed unconditionally, so the 5 x (K / 64) vector reads of a thread are issued back to back.  (With an
// `if (a < A)` around each product hipcc built one basic block per product - every LDS read followed by its own wait,
// 2.3 k cycles per step for K = 256.)  The result y[a] is valid in every lane of the trajectory's 16-lane group.
// ZEROED = false: Wo / bo are the caller's global tensors with exactly A rows (tail value net: A = 1).
template <bool ZEROED, class WP, class BP>
__device__ __forceinline__ void mlp_head(WP Wo, int ldw, BP bo, int K, int A, const float* hcur, int ldh,
                                         int tid, float (&y)[GOPS_MAX_ACT]) {
    const int hm = tid >> 4, hp = tid & 15;
#pragma unroll
    for (int a = 0; a < GOPS_MAX_ACT; ++a) y[a] = 0.f;
    if ((K & 63) == 0 && (ldw & 3) == 0) {   // 16-byte LDS reads, conflict-free within a 16-lane group
#pragma unroll 4
        for (int k = 4 * hp; k < K; k += 64) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(hcur + hm * ldh + k);
            if constexpr (ZEROED) {
                f32x4 wv[GOPS_MAX_ACT];
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a) wv[a] = ld4(Wo + a * ldw + k);
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a)
                    y[a] += hv[0] * wv[a][0] + hv[1] * wv[a][1] + hv[2] * wv[a][2] + hv[3] * wv[a][3];
            } else {
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a)
                    if (a < A) {
                        const f32x4 wv = ld4(Wo + a * ldw + k);
                        y[a] += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
                    }
            }
        }
    } else {
        for (int k = hp; k < K; k += 16) {
            const float hv = hcur[hm * ldh + k];
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                if (a < A) y[a] += hv * Wo[a * ldw + k];
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

// Plane-split policy evaluation of the register-stationary kernels (common.h SplitDev): obs tile xs -> head
// pre-activation.  Per step: the observation tile becomes a plane image (natural column order), hidden layer 0 runs on
// it and leaves the plane image of H_1 (plane_store order), hidden layer 1 runs on that; H_1 / H_2 (and gelu' for GELU)
// go to the fp32 FM stash straight from the epilogue registers like in mlp_hidden_forward.  The output layer is formed
// from the H_2 registers: every lane multiplies its 4 x 4 values with its own four head-weight columns (registers,
// loaded once), 16-lane DPP sums give the per-wave partial of each (row, action), and the four waves' partials meet in
// s_part - H_2 never touches LDS.  Returns the head pre-activation of (trajectory tid >> 4, action tid & 15) for
// tid & 15 < A (after the barrier that publishes s_part).
// KC0 > SPLIT_MAX_RESIDENT_KC0: layer 0's planes stream from L2 (StreamQ, common.h)
#define SPLIT_MAX_RESIDENT_KC0 4
template <int KC0, int AMAX>   // AMAX: compile-time bound of the action dimension (registers of the head partials)
struct SplitPolicy {
    static constexpr bool STREAM0 = KC0 > SPLIT_MAX_RESIDENT_KC0;
    typename std::conditional<STREAM0, StreamQ<KC0, 4>, StatQ<KC0, 4, true>>::type Q0;   // layer 0: bf16 plane in registers, half residual plane in LDS - or both streamed
    // AGPR pinning of layer 1's planes (common.h pin_agpr): both planes for the veh3dofconti instantiations (AMAX == 2) - with the
    // launch-time loads batched (round 5) pinning costs nothing at launch any more, and the 22 unpinned residual fragments that
    // hipcc copied back to VGPRs in front of their MFMAs (88 x v_accvgpr_read per step) are gone: forward 181.7 -> 179.5 us; the
    // idpendulum / lq instantiations measured better with the bf16 plane only (cfg2: 234.9 vs 236.9 us)
    static constexpr int PIN1 = (AMAX == 2) ? 3 : GOPS_PIN_MODE;
    StatQ<8, 4, false, PIN1> Q1;       // layer 1: both planes in registers
    // r0_lds: LDS region for layer 0's residual plane (16 n-tiles x KC0 chunks x 1 KiB); the caller's barrier publishes it
    __device__ __forceinline__ void load(const RolloutParams& p, int tid, f16x8* r0_lds) {
        const MlpDev& M = p.pol;
        if constexpr (STREAM0) Q0.load(p.sp.w1[0], p.sp.r[0], p.sp.inv[0], M.dims[1] >> 4, tid);
        else Q0.load(p.sp.w1[0], p.sp.r[0], p.sp.inv[0], M.dims[1] >> 4, tid, r0_lds);
        Q1.load(p.sp.w1[1], p.sp.r[1], p.sp.inv[1], M.dims[2] >> 4, tid);
    }
    // s_bias: [.][ldb] hidden biases; s_wo4: [256][4] head weights, feature-major, rows a >= A zero; s_bo: [4] head bias
    // what run() reads of the parameter block, pinned to scalar registers by the caller (common.h keep_s)
    struct Hot {
        int kp0, act;
        float *h1, *h2, *z1, *z2;
    };
    unsigned ovf = 0;   // half-range overflow of a plane conversion (common.h split2h), tested at the end of a tile
    __device__ __forceinline__ float run(const Hot& hot, const float* xs, int ldx, char* xq, int rowb0, char* hq,
                                         float* s_part, const float* s_bias, int ldb, const float* s_wo4, const float* s_bo,
                                         int tid, bool stash, size_t row0, DbgClock& dbg, bool combine = true) {
        const int lane = tid & 63, wave = tid >> 6, m0 = (lane >> 4) << 2;
        constexpr int ROWB1 = 2 * 256 + 16;
        StreamRing<KC0> ring0;
        if constexpr (STREAM0) Q0.prime(ring0, 0);   // the first chunks of W_0's planes travel during the conversion pass
        plane_convert_x(xs, ldx, hot.kp0, 32 * KC0, xq, rowb0, tid, SPLIT_FWD_SA, ovf);
        __syncthreads();
        DBG_TICK(1)
        const bool gelu = hot.act == GOPS_ACT_GELU;
        // ---- hidden layer 0 ----
        {
            f32x4 acc[4] = {}, accr[4] = {};
            if constexpr (STREAM0) {   // two n-tiles at a time through the ring
                f32x4 pa[2] = {}, pr[2] = {};
                gemm_split_pair(xq, rowb0, Q0, ring0, 0, lane, pa, pr);
                Q0.prime(ring0, 1);
                acc[0] = pa[0]; acc[1] = pa[1]; accr[0] = pr[0]; accr[1] = pr[1];
                f32x4 pb[2] = {}, ps[2] = {};
                gemm_split_pair(xq, rowb0, Q0, ring0, 1, lane, pb, ps);
                acc[2] = pb[0]; acc[3] = pb[1]; accr[2] = ps[0]; accr[3] = ps[1];
            } else {
                gemm_split(xq, rowb0, Q0, lane, acc, accr);
            }
            DBG_TICK(14)
            float* hrow = stash ? hot.h1 + row0 * 256 : nullptr;
            float* zrow = (stash && gelu) ? hot.z1 + row0 * 256 : nullptr;
            f32x4 hv[4];
            act_dispatch(hot.act, [&]<int ACT>() {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = 64 * wave + 16 * q + (lane & 15);
                    const float sc = Q0.inv[q], bn = s_bias[n];
                    f32x4 zv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = split_preact(acc[q][r], accr[q][r], sc, bn);
                        float hr, dr = z;
                        if constexpr (ACT == GOPS_ACT_GELU) gelu_pair(z, hr, dr);
                        else hr = act_fwd_t<ACT>(z);
                        hv[q][r] = hr; zv[r] = dr;
                    }
                    if (hrow != nullptr) __builtin_nontemporal_store(hv[q], gptr(reinterpret_cast<f32x4*>(hrow + n * 16 + m0)));
                    if (ACT == GOPS_ACT_GELU && zrow != nullptr) __builtin_nontemporal_store(zv, gptr(reinterpret_cast<f32x4*>(zrow + n * 16 + m0)));
                }
            });
            plane_store(hq, ROWB1, wave, lane, hv, SPLIT_FWD_SA, ovf);
        }
        DBG_TICK(8)
        __syncthreads();
        DBG_TICK(9)
        // ---- hidden layer 1 + output layer partials ----
        {
            f32x4 acc[4] = {}, accr[4] = {};
            gemm_split(hq, ROWB1, Q1, lane, acc, accr);
            DBG_TICK(11)
            float* hrow = stash ? hot.h2 + row0 * 256 : nullptr;
            float* zrow = (stash && gelu) ? hot.z2 + row0 * 256 : nullptr;
            float part[4][AMAX] = {};
            act_dispatch(hot.act, [&]<int ACT>() {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = 64 * wave + 16 * q + (lane & 15);
                    const float sc = Q1.inv[q], bn = s_bias[ldb + n];
                    const f32x4 wq = *reinterpret_cast<const f32x4*>(s_wo4 + n * 4);
                    f32x4 h4, zv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = split_preact(acc[q][r], accr[q][r], sc, bn);
                        float hr, dr = z;
                        if constexpr (ACT == GOPS_ACT_GELU) gelu_pair(z, hr, dr);
                        else hr = act_fwd_t<ACT>(z);
                        h4[r] = hr; zv[r] = dr;
#pragma unroll
                        for (int a = 0; a < AMAX; ++a) part[r][a] = fmaf(hr, wq[a], part[r][a]);
                    }
                    if (hrow != nullptr) __builtin_nontemporal_store(h4, gptr(reinterpret_cast<f32x4*>(hrow + n * 16 + m0)));
                    if (ACT == GOPS_ACT_GELU && zrow != nullptr) __builtin_nontemporal_store(zv, gptr(reinterpret_cast<f32x4*>(zrow + n * 16 + m0)));
                }
            });
            DBG_TICK(12)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < AMAX; ++a) ps[a] = row16_sum(part[r][a]);
                if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(s_part + (wave * TB + m0 + r) * 4) = ps;
            }
        }
        DBG_TICK(2)
        __syncthreads();
        if (!combine) return 0.f;   // (the caller sums the four waves' partials itself, in the thread layout of its env phase)
        const int hm = tid >> 4, la = tid & 15;
        float ya = 0.f;
        if (la < GOPS_MAX_ACT)
            ya = ((s_part[(0 * TB + hm) * 4 + la] + s_part[(1 * TB + hm) * 4 + la]) +
                  (s_part[(2 * TB + hm) * 4 + la] + s_part[(3 * TB + hm) * 4 + la])) + s_bo[la];
        DBG_TICK(6)
        return ya;
    }
};
struct NoSplit {};

// Streamed-split forward (SS): the hidden stack of ANY depth (all layers 256 wide) on plane-split MFMAs with every layer's
// weight planes streamed from L2 (StreamQ, two n-tiles at a time) - for the launches the register-stationary kernels do not
// take (three hidden layers, tail value nets with many tiles per CU, ...).  ONE activation plane buffer `hq`, rewritten in
// place behind a barrier (two buffers would cost the second workgroup per CU); head partials from the last layer's
// registers as in SplitPolicy.  s_wo4: [256][4] head weights, feature-major, unused rows zero; returns the head output of
// (trajectory tid >> 4, output tid & 15) in lanes tid & 15 < 4.
template <int AMAX>
__device__ __forceinline__ float ss_net_forward(const MlpDev& M, const SplitNetDev& S, const float* xs, int ldx, char* xq, char* hq,
                                                float* s_part, const float* s_bias, int ldb, const float* s_wo4, const float* s_bo,
                                                int tid, float* const* stash_h, float* const* stash_z, size_t row0, DbgClock& dbg, unsigned& ovf,
                                                float* dmp = nullptr) {
    const int lane = tid & 63, wave = tid >> 6, m0 = (lane >> 4) << 2;
    constexpr int ROWB1 = 2 * 256 + 16;
    const int L = M.nl - 1, rowb0 = split_rowb(32 * S.kc[0]);
    const bool gelu = M.act == GOPS_ACT_GELU;
    plane_convert_x(xs, ldx, M.kp[0], 32 * S.kc[0], xq, rowb0, tid, SPLIT_FWD_SA, ovf);
    __syncthreads();
    DBG_TICK(1)
    float part[4][AMAX] = {};
    for (int j = 0; j < L; ++j) {
        f32x4 acc[4] = {}, accr[4] = {};
        float inv[4];
        if (j == 0) {
            switch (S.kc[0]) {
                case 1: ss_layer_gemm<1>(xq, rowb0, S.w1[0], S.r[0], S.inv[0], 16, tid, acc, accr, inv); break;
                case 2: ss_layer_gemm<2>(xq, rowb0, S.w1[0], S.r[0], S.inv[0], 16, tid, acc, accr, inv); break;
                case 4: ss_layer_gemm<4>(xq, rowb0, S.w1[0], S.r[0], S.inv[0], 16, tid, acc, accr, inv); break;
                default: ss_layer_gemm<8>(xq, rowb0, S.w1[0], S.r[0], S.inv[0], 16, tid, acc, accr, inv); break;
            }
        } else {
            ss_layer_gemm<8>(hq, ROWB1, S.w1[j], S.r[j], S.inv[j], 16, tid, acc, accr, inv);
            DBG_TICK(11)
            __syncthreads();   // every wave has read the activation image it is about to overwrite
            DBG_TICK(9)
        }
        if (j == 0) DBG_TICK(14)
        float* hrow = (stash_h != nullptr) ? stash_h[j + 1] + row0 * 256 : nullptr;
        float* zrow = (stash_z != nullptr && gelu) ? stash_z[j + 1] + row0 * 256 : nullptr;
        const bool last = j == L - 1;
        f32x4 hv[4];
        act_dispatch(M.act, [&]<int ACT>() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 64 * wave + 16 * q + (lane & 15);
                const float sc = inv[q], bn = s_bias[j * ldb + n];
                f32x4 zv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = split_preact(acc[q][r], accr[q][r], sc, bn);
                    float hr, dr = z;
                    if constexpr (ACT == GOPS_ACT_GELU) gelu_pair(z, hr, dr);
                    else hr = act_fwd_t<ACT>(z);
                    hv[q][r] = hr; zv[r] = dr;
                }
                if (hrow != nullptr) __builtin_nontemporal_store(hv[q], gptr(reinterpret_cast<f32x4*>(hrow + n * 16 + m0)));
                if (ACT == GOPS_ACT_GELU && zrow != nullptr) __builtin_nontemporal_store(zv, gptr(reinterpret_cast<f32x4*>(zrow + n * 16 + m0)));
            }
        });
#ifdef GOPS_DUMP
        if (dmp != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) gptr(dmp)[32 + 16 * (j > 0 ? 1 : 0) + 4 * q + r] = hv[q][r];
        }
#endif
        DBG_TICK(8)
        if (!last) {
            plane_store(hq, ROWB1, wave, lane, hv, SPLIT_FWD_SA, ovf);
            __syncthreads();
            DBG_TICK(12)
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 64 * wave + 16 * q + (lane & 15);
                const f32x4 wq = *reinterpret_cast<const f32x4*>(s_wo4 + n * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) part[r][a] = fmaf(hv[q][r], wq[a], part[r][a]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < AMAX; ++a) ps[a] = row16_sum(part[r][a]);
        if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(s_part + (wave * TB + m0 + r) * 4) = ps;
    }
    __syncthreads();
    const int hm = tid >> 4, la = tid & 15;
    float ya = 0.f;
    if (la < GOPS_MAX_ACT)
        ya = ((s_part[(0 * TB + hm) * 4 + la] + s_part[(1 * TB + hm) * 4 + la]) +
              (s_part[(2 * TB + hm) * 4 + la] + s_part[(3 * TB + hm) * 4 + la])) + s_bo[la];
    DBG_TICK(2)
    return ya;
}

// SK0 / SK1: k-chunks (16 inputs each) of hidden layers 0 / 1 when their weights are register-
// stationary (the layer must then be 256 wide), 0 = streamed.
// TAIL: the INFADP terminal value V_target(obs_H) is evaluated after the loop (compiled out for FHADP so
// that its streamed-GEMM registers do not add to the pressure of the stationary variants).
// F16: GOPS_DTYPE_F16 - the hidden layers run on v_mfma_f32_16x16x32_f16 (rollout_f16.h) and the
// activation stash is half precision; everything else in the step is the same fp32 code.
// (F16 kernels: 4 workgroups per CU - launch bound 4 waves / SIMD, <= 128 registers.)
// GEN (streamed fp32 kernels of the obs == state env kinds): ActionRepeatModel - the env blocks loop over
// GopsEnv.repeat_num sub-steps; with GEN = false the loops have the compile-time trip count 1
// SPLIT: plane-split contractions (SplitPolicy): SK0 then counts the 32-wide chunks of layer 0, SK1 = 8
// MULTI (SPLIT only): more tiles than workgroups - the kernel walks its tiles grid-stride (the single-tile instantiation
// keeps fewer values live across the step loop)
// SS: streamed-split forward (ss_net_forward): plane-split MFMAs with all weight planes streamed from L2, two workgroups per CU
// N64 (narrow launches of obs -> 64 -> 64 -> act policies: RolloutParams.narrow == 2): the hidden stack by mlp_hidden_forward_n64
template <int ENV, int SK0, int SK1, bool TAIL, bool F16 = false, bool GEN = false, bool SPLIT = false, bool MULTI = false, bool SS = false, bool N64 = false>
__global__ __launch_bounds__(NTHREADS, F16 ? 4 : (SS ? 2 : ((SK0 == 0 && SK1 == 0) ? 3 : 1))) void rollout_fwd_kernel(const RolloutParams* __restrict__ pp) {
    extern __shared__ __attribute__((aligned(16))) float smem_raw[];
    const RolloutParams& p = *pp;   // parameters live in device memory: uniform scalar loads
    const int tid = threadIdx.x;
#ifdef GOPS_DBG_BUILD
    const long long t_entry = clock64();   // (phase-counter build: cycles from kernel entry to the first step -> slot 13)
#endif
    // pyth_lq on the streamed-split forward: the env description in front of everything else in LDS (common.h: env_in_lds)
    constexpr bool ENVLDS = env_in_lds(ENV, SS);
    float* smem = smem_raw + (ENVLDS ? ENV_LDS_FLOATS : 0);
    const GopsEnv* env_ptr;
    if constexpr (ENVLDS) {
        for (int idx = tid; idx < (int)(sizeof(GopsEnv) / 4); idx += NTHREADS) smem_raw[idx] = gptr(reinterpret_cast<const float*>(&p.env))[idx];
        env_ptr = reinterpret_cast<const GopsEnv*>(smem_raw);
        __syncthreads();
    } else {
        env_ptr = &p.env;
    }
    const GopsEnv& env = *env_ptr;
    // SPLIT kernels are launched with at most one workgroup per CU and walk tiles tile, tile + gridDim.x, ... with their
    // weights resident; every other variant has one tile per workgroup (the tile loop below runs once)
    int tile = blockIdx.x;
    int b0 = tile * TB;
    int nvalid = min(TB, p.B - b0);
    const int O = p.env.obs_dim, A = p.env.act_dim;
    constexpr bool SURR = (ENV == GOPS_ENV_VEH3DOF_SURR);   // veh3dofconti + surrounding vehicles + constraint outputs
    constexpr bool VEH = (ENV == GOPS_ENV_VEH3DOFCONTI) || SURR;
    constexpr bool VEH2 = (ENV == GOPS_ENV_VEH2DOF);   // 2-DOF lateral model: state [4], reference points (y, phi)
    constexpr bool REF = VEH || VEH2;                   // models with a reference-trajectory table
    constexpr bool MOB = (ENV == GOPS_ENV_MOBILEROBOT); // obs == state [13], one constraint on the new state
    constexpr bool CSTR = SURR || VEH2 || MOB;          // models with constraint outputs
    // leading dimensions are compile-time constants in the register-stationary variants
    const int ldx = p.ldx, ldh = (SK1 > 0) ? 260 : p.ldh;
    float* xs = smem;                       // [TB][ldx]  current observation (+ time column)
    float* ha = xs + TB * ldx;              // [TB][ldh]  (SPLIT: only the tail value net uses ha / hb; they alias the plane images)
    float* hb = ha + hidden_tile_floats(ldh, F16);       // [TB][ldh] floats, or [TB][ldh + 4] halfs (F16)
    float* s_state = (SPLIT || SS) ? xs + TB * ldx : hb + hidden_tile_floats(ldh, F16);  // [TB][8]
    float* s_act = s_state + TB * 8;        // [TB][4] wrapped action
    float* s_th = s_act + TB * 4;           // [TB][4] tanh(head) (ENV_NONE: raw head output)
    float* s_done = s_th + TB * 4;          // [TB]
    float* s_bo = s_done + TB;              // [4]  head bias
    float* s_ac = s_bo + TB;                // [4][8] per-action constants (stage_act_const)
    float* s_wo = s_ac + 2 * TB;            // [4][ldh] head weights
    float* s_bias = s_wo + 4 * ldh;         // [GOPS_MAX_LAYERS-1][ldh] hidden-layer biases
    f32x4* s_ref = reinterpret_cast<f32x4*>(s_bias + (GOPS_MAX_LAYERS - 1) * ldh);   // veh: [TB][TL]
    const int TL = p.env.pre_horizon + 1 + p.H;   // reference-table points per trajectory
    // F16: half copy of the policy / value input tile, [TB][ldx16], behind the reference-table region
    _Float16* x16 = reinterpret_cast<_Float16*>(s_ref + (REF ? TB * TL : 0));
    const int ldx16 = (((p.ldx - 4) + 31) & ~31) + 8, ld16 = (p.ldh - 4) + 8;
    {   // one-time staging of everything the H-step loop would otherwise re-fetch from L2 (batched_fill: loads in flight together)
        const int Lh = p.pol.nl - 1, K = p.pol.dims[Lh], Ao = p.pol.dims[p.pol.nl];
        const GLOBAL_AS float* wl = gptr(p.pol.w[Lh]);
        // rows a >= Ao (and the pad columns) are zero: mlp_head<true>
        batched_fill<5>(GOPS_MAX_ACT * ldh, tid,
                        [&](int idx) {
                            int a, k;
                            if constexpr (SPLIT || SS) { k = idx >> 2; a = idx & 3; }   // feature-major [K][4]: a lane reads the four action weights of one of its columns as one vector
                            else { a = idx / ldh; k = idx - a * ldh; }
                            const bool ok = a < Ao && k < K;
                            const float v = wl[ok ? a * K + k : 0];
                            return ok ? v : 0.f;
                        },
                        [&](int idx, float v) { s_wo[idx] = v; });
        if (tid < GOPS_MAX_ACT) s_bo[tid] = (tid < Ao) ? gptr(p.pol.b[Lh])[tid] : 0.f;
        stage_act_const(p.env, s_ac, tid);
        batched_fill<GOPS_MAX_LAYERS - 1>(Lh * ldh, tid,
                                          [&](int idx) {
                                              const int j = idx / ldh, n = idx - j * ldh;
                                              const bool ok = n < p.pol.dims[j + 1];
                                              const float v = gptr(p.pol.b[j])[ok ? n : 0];
                                              return ok ? v : 0.f;
                                          },
                                          [&](int idx, float v) { s_bias[idx] = v; });
    }
    // narrow nets on the plain streamed fp32 kernels: the packed hidden-layer weights of the policy, resident in LDS (common.h)
    constexpr bool NARROWABLE = (SK0 == 0) && (SK1 == 0) && !F16 && !SPLIT && !SS;
    const f32x4* s_narrow = nullptr;
    if constexpr (NARROWABLE) {
        if (p.narrow) {
            f32x4* dst = reinterpret_cast<f32x4*>(smem_raw + p.narrow_off_fwd);
            s_narrow = dst;
            for (int j = 0; j < p.pol.nl - 1; ++j) {
                const int n4 = (p.pol.kp[j] * p.pol.dims[j + 1]) >> 2;
                narrow_fill(dst, p.pol.wp[j], n4, tid);
                dst += n4;
            }
        }
    }
    Hot64 hot64 = {};
    if constexpr (N64) {
        static_assert(NARROWABLE, "N64 is a variant of the plain streamed fp32 kernel");
        hot64.kch0 = keep_s(p.pol.kp[0] >> 4);
        hot64.act = keep_s(p.pol.act);
        const bool st = p.need_grad != 0;
        hot64.h1 = st ? keep_s(p.st.h[1]) : nullptr; hot64.h2 = st ? keep_s(p.st.h[2]) : nullptr;
        hot64.z1 = (st && p.pol.act == GOPS_ACT_GELU) ? keep_s(p.st.z[1]) : nullptr;
        hot64.z2 = (st && p.pol.act == GOPS_ACT_GELU) ? keep_s(p.st.z[2]) : nullptr;
        hot64.w0 = s_narrow;
        hot64.w1 = s_narrow + hot64.kch0 * 4 * 64;
    }
    typename std::conditional<(SK0 > 0 && !SPLIT), StatW<(SK0 > 0 ? SK0 : 1), 4>, NoW>::type W0;
    typename std::conditional<(SK1 > 0 && !SPLIT), StatW<(SK1 > 0 ? SK1 : 1), 4>, NoW>::type W1;
    if constexpr (SK0 > 0 && !SPLIT) W0.load(p.pol.wp[0], p.pol.dims[1] >> 4, tid, p.pol.kp[0] >> 4);
    if constexpr (SK1 > 0 && !SPLIT) W1.load(p.pol.wp[1], p.pol.dims[2] >> 4, tid);
    // SPLIT: plane images of the observation tile and of H_1 behind the reference-table region, then the head partials
    constexpr int AMAX = (ENV == GOPS_ENV_IDPENDULUM) ? 1 : ((ENV == GOPS_ENV_VEH3DOFCONTI) ? 2 : GOPS_MAX_ACT);
    typename std::conditional<SPLIT, SplitPolicy<(SK0 > 0 ? SK0 : 1), AMAX>, NoSplit>::type SP;
    // (SS: the observation image is sized for the wider of the policy's and the tail value net's padded inputs)
    const int rowb0 = SS ? split_rowb(32 * max(p.ssp.kc[0], TAIL ? p.ssv.kc[0] : 0)) : split_rowb(32 * SK0);
    char* xq = reinterpret_cast<char*>(s_ref + (REF ? TB * TL : 0));
    char* hq = xq + 4 * TB * rowb0;
    float* s_part = reinterpret_cast<float*>(hq + 4 * TB * split_rowb(256));   // [4 waves][TB][4]
    f16x8* r0_lds = reinterpret_cast<f16x8*>(s_part + 4 * TB * 4);             // layer 0's residual plane: 16 n-tiles x SK0 KiB
    if constexpr (SPLIT) {
        SP.load(p, tid, r0_lds);
        ha = reinterpret_cast<float*>(xq);   // tail value net (after the loop): its fp32 tiles alias the dead plane images
        hb = ha + TB * ldh;
    }
    if constexpr (SS && TAIL) {   // (p.tail_fp32: the same aliasing; 64 (rowb0 + 528) bytes >= two [TB][260] fp32 tiles for every rowb0)
        ha = reinterpret_cast<float*>(xq);
        hb = ha + TB * ldh;
    }
    const IdpConst IC = idp_const();
    const VehConst VC = veh_const();

    DbgClock dbg;
    dbg.init((p.dbg != nullptr) && blockIdx.x == 0 && tid == 0);
#ifdef GOPS_DBG_BUILD
    if (dbg.on) dbg.acc[13] = dbg.last - t_entry;
#endif
    const int ntiles = (p.B + TB - 1) / TB;
    // SPLIT: everything the step loop reads of the parameter block, pinned to scalar registers (common.h keep_s) - left to
    // hipcc each of these is an s_load + s_waitcnt lgkmcnt(0) per step (16 of them in the round-4 forward)
    const int hH = keep_s<SPLIT>(p.H), hfh = keep_s<SPLIT>(p.fh), hng = keep_s<SPLIT>(p.need_grad), hB = keep_s<SPLIT>(p.B);
    const int hkp0 = keep_s<SPLIT>(p.pol.kp[0]), hP = keep_s<SPLIT>(p.env.pre_horizon);
    const int hnomask = keep_s<SPLIT>(p.env.no_mask_at_done), hshaping = keep_s<SPLIT>(p.env.shaping);
    const float hrshift = keep_s<SPLIT>(p.env.reward_shift), hrscale = keep_s<SPLIT>(p.env.reward_scale);
    float* const hst_x = keep_s<SPLIT>(p.st.x);
    float* const hst_env = keep_s<SPLIT>(p.st.env);
    float* const hrewards = keep_s<SPLIT>(p.out.rewards);
    typename std::conditional<SPLIT, typename SplitPolicy<(SK0 > 0 ? SK0 : 1), AMAX>::Hot, NoSplit>::type sp_hot;
    if constexpr (SPLIT) {
        sp_hot.kp0 = hkp0;
        sp_hot.act = keep_s(p.pol.act);
        sp_hot.h1 = keep_s(p.st.h[1]); sp_hot.h2 = keep_s(p.st.h[2]);
        sp_hot.z1 = keep_s(p.st.z[1]); sp_hot.z2 = keep_s(p.st.z[2]);
    }
    do {   // ---- one tile of 16 trajectories (SPLIT: a grid-stride walk over the tiles) ----
    b0 = tile * TB;
    nvalid = min(TB, p.B - b0);
    if constexpr (SPLIT && MULTI && TAIL) {   // the previous tile's tail value net left ITS biases in s_bias
        for (int j = 0; j < p.pol.nl - 1; ++j)
            for (int n = tid; n < p.pol.dims[j + 1]; n += NTHREADS) s_bias[j * ldh + n] = gptr(p.pol.b[j])[n];
    }
    if (REF) {
        const GLOBAL_AS f32x4* tbl = gptr(reinterpret_cast<const f32x4*>(p.ref_table)) + (size_t)b0 * TL;
        const int nv = nvalid * TL;
        batched_fill<4>(TB * TL, tid,
                        [&](int idx) {
                            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                            const f32x4 v = tbl[idx < nv ? idx : 0];
                            return idx < nv ? v : z;
                        },
                        [&](int idx, const f32x4& v) { s_ref[idx] = v; });
    }
    {
        const GLOBAL_AS float* ob = gptr(p.in.obs) + (size_t)b0 * O;
        batched_fill<5>(TB * ldx, tid,
                        [&](int idx) {
                            const int m = idx / ldx, c = idx - m * ldx;
                            const bool ok = c < O && m < nvalid;
                            const float v = ob[ok ? m * O + c : 0];
                            return ok ? v : 0.f;
                        },
                        [&](int idx, float v) { xs[idx] = v; });
    }
    // (no MaskAtDoneModel in the chain: the base models ignore the done flags they are handed)
    if (tid < TB) s_done[tid] = (tid < nvalid && p.in.done != nullptr && !p.env.no_mask_at_done && gptr(p.in.done)[b0 + tid] != 0.f) ? 1.f : 0.f;
    if (VEH) {
        if (tid < TB * 6) {
            const int m = tid / 6, c = tid - m * 6;
            s_state[m * 8 + c] = (m < nvalid) ? gptr(p.in.state)[(size_t)(b0 + m) * 6 + c] : (c == 3 ? 1.f : 0.f);
        }
    }
    if (VEH2) {
        if (tid < TB * 8) {
            const int m = tid >> 3, c = tid & 7;
            s_state[m * 8 + c] = (m < nvalid && c < 4) ? gptr(p.in.state)[(size_t)(b0 + m) * 4 + c] : 0.f;
        }
    }
    float v_acc = 0.f;
    unsigned ss_ovf = 0;   // SS: half-range overflow of a plane conversion in this tile (SPLIT keeps it in SP.ovf)
    float c_ext = 0.f, c_lin = 0.f, c_int = 0.f, c_feas = 1.f;   // SURR: discounted constraint sums of trajectory tid (tid < TB)
    float c_mul[GOPS_MAX_CONSTRAINT] = {1.f, 1.f, 1.f}, c_safe[GOPS_MAX_CONSTRAINT] = {1.f, 1.f, 1.f};   // SPIL products
    float veh_s = 0.f, veh_c = 1.f;   // sin/cos of the current heading, carried across steps
    __syncthreads();
    if (VEH) sincosf(s_state[(tid & 15) * 8 + 2], &veh_s, &veh_c);
    // FASTV (plane-split stationary kernel, veh3dofconti = the headline workload): thread (m = tid & 15, part = tid >> 4) keeps the
    // state and the done flag of trajectory m in registers across the steps, sums the head partials of ITS trajectory, squashes
    // and wraps both actions itself - all 16 parts of a trajectory compute the same values - and walks its reference points.
    // Against the general path below that is two barriers per step less (behind tanh / wrap, and inside the env phase) and no
    // LDS round trip for actions, state and done flag.  Same functions in the same order: bit-identical results.
    constexpr bool FASTV = SPLIT && ENV == GOPS_ENV_VEH3DOFCONTI;
    float fs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, fdone = 0.f;
    ActC fc0 = {}, fc1 = {};
    if constexpr (FASTV) {
#pragma unroll
        for (int i = 0; i < 6; ++i) fs[i] = s_state[(tid & 15) * 8 + i];
        fdone = s_done[tid & 15];
        fc0 = act_const(s_ac, 0);
        fc1 = act_const(s_ac, 1);
    }
    settle_loads();
    for (int t = 0; t < hH; ++t) {
        if (hfh && tid < TB) xs[tid * ldx + O] = (float)(t + 1);
        __syncthreads();
        DBG_TICK(0)
        const float gpow_t = p.gpow[t];   // (requested at the top of the step: consumed by the reward bookkeeping at its end)
        const size_t row0 = ((size_t)tile * hH + t) * TB;   // tile-major stash: a tile's rows are contiguous over t
        if constexpr (F16) {
            // half copy of the input tile (LDS, and the stash rows the weight-gradient GEMM reads); the env
            // adjoints get the first 8 observation columns in fp32
            convert_x_h(xs, ldx, p.pol.kp[0], p.pol.kp32[0], x16, ldx16,
                        p.need_grad ? reinterpret_cast<_Float16*>(p.st.x) : nullptr, row0, tid);
            if (p.need_grad && tid < 2 * TB)
                *gptr(reinterpret_cast<f32x4*>(p.st.xf + (row0 + (tid >> 1)) * 8 + 4 * (tid & 1))) =
                    *reinterpret_cast<const f32x4*>(xs + (tid >> 1) * ldx + 4 * (tid & 1));
            __syncthreads();
        } else {
            if (hng) stash_tile_fm(xs, ldx, hkp0, hst_x + row0 * hkp0, tid);
        }
        if constexpr (!SPLIT) DBG_TICK(1)
        {
            float y[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
            float ya_split = 0.f;
            if constexpr (SPLIT) {
                ya_split = SP.run(sp_hot, xs, ldx, xq, rowb0, hq, s_part, s_bias, ldh, s_wo, s_bo, tid, hng != 0, row0, dbg, !FASTV);
            } else if constexpr (SS) {
#ifdef GOPS_DUMP
                float* dmp = (p.dbg != nullptr && VEH) ? reinterpret_cast<float*>(p.dbg) + ((((size_t)tile * p.H + t) * NTHREADS + tid) << 6) : nullptr;
                if (dmp != nullptr) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) gptr(dmp)[28 + k] = xs[(tid >> 4) * ldx + (tid & 15) + 16 * k];
                }
                ya_split = ss_net_forward<AMAX>(p.pol, p.ssp, xs, ldx, xq, hq, s_part, s_bias, ldh, s_wo, s_bo, tid,
                                                p.need_grad ? p.st.h : nullptr, p.need_grad ? p.st.z : nullptr, row0, dbg, ss_ovf, dmp);
                if (dmp != nullptr) gptr(dmp)[0] = ya_split;
#else
                ya_split = ss_net_forward<AMAX>(p.pol, p.ssp, xs, ldx, xq, hq, s_part, s_bias, ldh, s_wo, s_bo, tid,
                                                p.need_grad ? p.st.h : nullptr, p.need_grad ? p.st.z : nullptr, row0, dbg, ss_ovf);
#endif
            } else
            if (!p.open_loop) {
                if constexpr (F16) {
                    const _Float16* hcur = mlp_hidden_forward_h(p.pol, x16, ldx16, reinterpret_cast<_Float16*>(ha),
                                                                reinterpret_cast<_Float16*>(hb), ld16, tid, s_bias, ldh,
                                                                p.need_grad ? p.st.h : nullptr, p.need_grad ? p.st.z : nullptr,
                                                                row0, TB);
                    DBG_TICK(2)
                    mlp_head_h<true>(s_wo, ldh, s_bo, p.pol.dims[p.pol.nl - 1], p.pol.dims[p.pol.nl], hcur, ld16, tid, y);
                } else {
                    float* hcur;
                    if constexpr (N64)
                        hcur = mlp_hidden_forward_n64(hot64, xs, ldx, ha, hb, ldh, tid, s_bias, row0, dbg);
                    else
                        hcur = mlp_hidden_forward(p.pol, W0, W1, xs, ldx, ha, hb, ldh, tid, s_bias,
                                                  p.need_grad ? p.st.h : nullptr, p.need_grad ? p.st.z : nullptr,
                                                  row0, dbg, s_narrow);
                    DBG_TICK(2)
                    mlp_head<true>(s_wo, ldh, s_bo, p.pol.dims[p.pol.nl - 1], p.pol.dims[p.pol.nl], hcur, ldh, tid, y);
                    DBG_TICK(6)
                }
            }
            if constexpr (!FASTV) {   // lane a of each 16-lane trajectory group squashes and wraps action a (row16_sum left y in every lane)
                const int hm = tid >> 4, la = tid & 15;
                float ya = (la == 0) ? y[0] : (la == 1) ? y[1] : (la == 2) ? y[2] : y[3];
                if constexpr (SPLIT || SS) ya = ya_split;
                if (p.open_loop && la < A && hm < nvalid)   // FHADP2: the sequence was emitted by one MLP evaluation outside
                    ya = gptr(p.in.head_pre)[((size_t)(b0 + hm) * p.H + t) * A + la];
                if (la < A) {
                    if (ENV == GOPS_ENV_NONE) {
                        s_th[hm * 4 + la] = ya;
                    } else if (p.open_loop == 2) {   // shooting over the raw model: head_pre IS the model action
                        s_th[hm * 4 + la] = ya;
                        s_act[hm * 4 + la] = ya;
                    } else {
                        const ActC c = act_const(s_ac, la);
                        const float th = (SPLIT || SS) ? fast_tanh(ya) : tanhf(ya);
                        s_th[hm * 4 + la] = th;
                        s_act[hm * 4 + la] = wrap_action(c, c.sc * th + c.of);
                    }
                }
            }
        }
        if constexpr (!FASTV) {
        DBG_TICK(7)
        __syncthreads();
        DBG_TICK(3)
        }
        if (!FASTV && p.need_grad && tid < TB) {   // env stash row: tanh outputs, done_t, state_t
            GLOBAL_AS f32x4* er = gptr(reinterpret_cast<f32x4*>(p.st.env + (row0 + tid) * ENV_STASH));
            f32x4 e0 = {s_th[tid * 4 + 0], s_th[tid * 4 + 1], s_th[tid * 4 + 2], s_th[tid * 4 + 3]};
            if (VEH) {   // two actions: the free slots carry the wrapped (steer, a_x) for the backward sweep
                e0[2] = s_act[tid * 4 + 0];
                e0[3] = s_act[tid * 4 + 1];
            }
            if (VEH2) e0[2] = s_act[tid * 4 + 0];   // wrapped steer
            f32x4 e1 = {s_done[tid], s_state[tid * 8 + 0], s_state[tid * 8 + 1], s_state[tid * 8 + 2]};
            f32x4 e2 = {s_state[tid * 8 + 3], s_state[tid * 8 + 4], s_state[tid * 8 + 5], 0.f};
            er[0] = e0;
            er[1] = e1;
            er[2] = e2;
        }

        if constexpr (!FASTV) DBG_TICK(4)
        // ---------------- env model step + MaskAtDone / ShapingReward / ClipObservation ---------
        float r = 0.f;          // raw model reward (threads tid < TB)
        bool done_m = false;    // done flag from the base model
        if (ENV == GOPS_ENV_NONE) {
            if (tid < TB) r = s_th[tid * 4];
        } else if (ENV == GOPS_ENV_LQ) {
            if (tid < TB) {
                const int m = tid;
                // NS / NA: compile-time loop bounds - (4, 2) for BASELINE configs[4] (lq s4a2), the maxima otherwise
                auto lq_step = [&]<int NS, int NA>() {
                    constexpr bool EXACT = NS < GOPS_MAX_LQ_STATE;   // the dimensions ARE (NS, NA): no run-time guards
                    float x[GOPS_MAX_LQ_STATE], xn[GOPS_MAX_LQ_STATE], u[GOPS_MAX_ACT];
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { x[i] = (i < NS && (EXACT || i < O)) ? obs_unscale(env, i, xs[m * ldx + i]) : 0.f; xn[i] = 0.f; }
#pragma unroll
                    for (int j = 0; j < GOPS_MAX_ACT; ++j) u[j] = (j < NA && (EXACT || j < A)) ? s_act[m * 4 + j] : 0.f;
                    // MaskAtDone freezes the (unscaled) observation; ScaleObservation rescales, ClipObservation clips the result
                    const bool frozen = s_done[m] != 0.f;
                    const int nrep = GEN ? env.repeat_num : 1;   // ActionRepeat: sub-steps with the initial done flag
                    float rs = 0.f;
                    for (int rep = 0; rep < nrep; ++rep) {
                        if (rep > 0 && !frozen) {
#pragma unroll
                            for (int i = 0; i < NS; ++i) x[i] = xn[i];
                        }
                        lq_forward<NS, NA>(env, x, u, xn, r);
                        rs = (GEN && !env.repeat_last_reward) ? rs + r : r;
                    }
                    r = rs;
                    if (!frozen || env.clip_obs || env.scale_obs) {
#pragma unroll
                        for (int i = 0; i < NS; ++i)
                            if (EXACT || i < O) {
                                const float v = obs_rescale(env, i, sel_reg(frozen, x[i], xn[i]));
                                xs[m * ldx + i] = env.clip_obs ? clampf(v, env.obs_low[i], env.obs_high[i]) : v;
                            }
                    }
                };
                if (O == 4 && A == 2) lq_step.template operator()<4, 2>();
                else lq_step.template operator()<GOPS_MAX_LQ_STATE, GOPS_MAX_ACT>();
            }
        } else if (ENV == GOPS_ENV_CARTPOLE || ENV == GOPS_ENV_PENDULUM) {
            if (tid < TB) {   // gym-style models: obs == state, same wrapper handling as pyth_lq
                const int m = tid;
                constexpr int NS = (ENV == GOPS_ENV_CARTPOLE) ? 4 : 3;
                float x[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NS; ++i) x[i] = obs_unscale(p.env, i, xs[m * ldx + i]);
                const float a = s_act[m * 4];
                const bool frozen = s_done[m] != 0.f;
                const int nrep = GEN ? p.env.repeat_num : 1;
                float rs = 0.f;
                for (int rep = 0; rep < nrep; ++rep) {
                    if (rep > 0 && !frozen) {
#pragma unroll
                        for (int i = 0; i < NS; ++i) x[i] = xn[i];
                    }
                    if (ENV == GOPS_ENV_CARTPOLE) {
                        cart_forward(cart_const(), x, a, xn, r, done_m);
                    } else {
                        PendStep w;
                        pend_forward(x, a, xn, r, w);
                    }
                    rs = (GEN && !p.env.repeat_last_reward) ? rs + r : r;
                }
                r = rs;
                if (!frozen || p.env.clip_obs || p.env.scale_obs) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const float v = obs_rescale(p.env, i, sel_reg(frozen, x[i], xn[i]));
                        xs[m * ldx + i] = p.env.clip_obs ? clampf(v, p.env.obs_low[i], p.env.obs_high[i]) : v;
                    }
                }
            }
        } else if (ENV == GOPS_ENV_IDPENDULUM) {
            if (tid < TB) {
                const int m = tid;
                float s[6], sn[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) s[i] = obs_unscale(p.env, i, xs[m * ldx + i]);
                float s_in[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) s_in[i] = s[i];
                const float a = s_act[m * 4];
                const float u = 500.f * a;
                const int nrep = GEN ? p.env.repeat_num : 1;   // (finished rows advance too: nothing of them is written)
                float rs = 0.f;
                // the sweep's parking of the sub-steps (plane-split stationary kernels: p.st.idp), written from the registers that
                // hold them here instead of being recomputed there (the recomputation - libm sincosf, five sub-steps - was ~45 %
                // of the sweep's env phase, which is half of a sweep step at cfg2)
                GLOBAL_AS f32x4* parkg = (p.need_grad && p.st.idp != nullptr) ? gptr(reinterpret_cast<f32x4*>(p.st.idp + (row0 + m) * IDP_PARK)) : nullptr;
                auto park_store = [&](int k, const IdpSub& w) {
                    if (parkg == nullptr) return;
                    GLOBAL_AS f32x4* d = parkg + k * 6;
                    d[0] = f32x4{s[0], s[1], s[2], s[3]};
                    d[1] = f32x4{s[4], s[5], w.s1, w.c1};
                    d[2] = f32x4{w.s2, w.c2, w.s12, w.c12};
                    d[3] = f32x4{w.inv[0], w.inv[1], w.inv[2], w.inv[3]};
                    d[4] = f32x4{w.inv[4], w.inv[5], w.qdd[0], w.qdd[1]};
                    d[5] = f32x4{w.qdd[2], 0.f, 0.f, 0.f};
                };
                for (int rep = 0; rep < nrep; ++rep) {
                    IdpSub w;
                    idp_substep<true>(IC, s, u, 0.002f, sn, w);
                    park_store(0, w);
#if GOPS_IDP_FWD_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
                    for (int k = 1; k < 5; ++k) {
                        idp_advance_trig(s, 0.002f, w, w);   // sin / cos of the new angles from the old ones (rotation by tau * theta_dot)
#pragma unroll
                        for (int i = 0; i < 6; ++i) s[i] = sn[i];
                        idp_substep<false>(IC, s, u, 0.002f, sn, w);
                        park_store(k, w);
                    }
#if GOPS_IDP_FAST
                    idp_advance_trig(s, 0.002f, w, w);   // cosines of the new angles for the termination test
#endif
#pragma unroll
                    for (int i = 0; i < 6; ++i) s[i] = sn[i];
                    // ([126], [127]: tanh(head) and the done flag the step started with - what the sweep's env adjoint needs of the env row:
                    // the sweeps that read the parking fetch nothing else at the top of a step)
                    if (parkg != nullptr) { parkg[30] = f32x4{s[0], s[1], s[2], s[3]}; parkg[31] = f32x4{s[4], s[5], s_th[m * 4], s_done[m]}; }
                    r = idp_reward(s, a);
                    rs = (GEN && !p.env.repeat_last_reward) ? rs + r : r;
#if GOPS_IDP_FAST
                    done_m = idp_done_trig(IC, s, w.c1, w.c2);
#else
                    done_m = idp_done(IC, s);
#endif
                }
                r = rs;
                if (s_done[m] == 0.f) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) xs[m * ldx + i] = obs_rescale(p.env, i, s[i]);
                } else if (p.env.scale_obs) {   // frozen rows pass through unscale / rescale like in the reference
#pragma unroll
                    for (int i = 0; i < 6; ++i) xs[m * ldx + i] = obs_rescale(p.env, i, s_in[i]);
                }
            }
        } else if (ENV == GOPS_ENV_MOBILEROBOT) {
            if (tid < TB) {   // obs == state; MaskAtDone freezes the observation, ClipObservation clips the result, the constraint is not masked
                const int m = tid;
                const MobConst MC = mob_const();
                float x[MOB_OBS], xn[MOB_OBS];
#pragma unroll
                for (int i = 0; i < MOB_OBS; ++i) x[i] = xs[m * ldx + i];
                float nv = 0.f, nw = 0.f;
                if (p.in.noise != nullptr && m < nvalid) {
                    const GLOBAL_AS float* nz = gptr(p.in.noise) + ((size_t)t * p.B + b0 + m) * 2;
                    nv = nz[0]; nw = nz[1];
                }
                float c;
                MobStep w;
                mob_forward(MC, x, s_act[m * 4 + 0], s_act[m * 4 + 1], nv, nw, xn, r, c, done_m, w);
                if (m < nvalid) {
                    const float cp = fmaxf(c, 0.f), cm = fminf(c, 0.f);
                    c_ext += cp * cp * p.gpow[t];
                    c_lin += cp * p.gpow[t];
                    c_int += logf(-cm + 1e-8f) * p.gpow[t];
                    if (!(c < 0.f)) c_feas = 0.f;
                    float dlog;
                    c_mul[0] *= spil_phi(c, dlog);
                    if (!(c <= 0.f)) c_safe[0] = 0.f;
                    if (p.out.constraints != nullptr) gptr(p.out.constraints)[(size_t)t * p.B + b0 + m] = c;
                }
                const bool frozen = s_done[m] != 0.f;
                if (!frozen || p.env.clip_obs) {
#pragma unroll
                    for (int i = 0; i < MOB_OBS; ++i) {
                        const float v = sel_reg(frozen, x[i], xn[i]);
                        xs[m * ldx + i] = p.env.clip_obs ? clampf(v, p.env.obs_low[i], p.env.obs_high[i]) : v;
                    }
                }
            }
        } else if (ENV == GOPS_ENV_VEH2DOF) {
            if (tid < TB) {
                const int m = tid, P = p.env.pre_horizon;
                const Veh2Const C2 = veh2_const();
                float s[4], sn[4], o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { s[i] = s_state[m * 8 + i]; o[i] = xs[m * ldx + i]; }
                const float steer = s_act[m * 4];
                r = veh2_reward(o, steer);
                if (p.env.cstr_err && m < nvalid) {   // pyth_veh2dofconti_errcstr: c = |delta_y| - tol of the CURRENT observation (unmasked)
                    const float c = fabsf(o[0]) - p.env.err_tol[0];
                    const float cp = fmaxf(c, 0.f), cm = fminf(c, 0.f);
                    c_ext += cp * cp * p.gpow[t];
                    c_lin += cp * p.gpow[t];
                    c_int += logf(-cm + 1e-8f) * p.gpow[t];
                    if (!(c < 0.f)) c_feas = 0.f;
                    float dlog;
                    c_mul[0] *= spil_phi(c, dlog);
                    if (!(c <= 0.f)) c_safe[0] = 0.f;
                    if (p.out.constraints != nullptr) gptr(p.out.constraints)[(size_t)t * p.B + b0 + m] = c;
                }
                float sphi, cphi;
                sincosf(s[1], &sphi, &cphi);
                veh2_f_xu(C2, s, steer, sphi, cphi, sn);
#pragma unroll
                for (int i = 0; i < 4; ++i) s_state[m * 8 + i] = sn[i];   // the info state advances whatever `done` says
                const f32x4* tbl = s_ref + m * TL + (t + 1);              // reference points of the new time: (., y, phi, .)
                const float o0 = sn[0] - tbl[0][1], o1 = sn[1] - tbl[0][2];
                done_m = (fabsf(o0) > 2.f) || (fabsf(o1) > 3.14159265358979323846f);
                if (s_done[m] == 0.f) {
                    float* xo = xs + m * ldx;
                    xo[0] = o0; xo[1] = o1; xo[2] = sn[2]; xo[3] = sn[3];
                    for (int i = 1; i <= P; ++i) xo[3 + i] = sn[0] - tbl[i][1];
                }
            }
        } else if constexpr (FASTV) {   // GOPS_ENV_VEH3DOFCONTI on the plane-split stationary kernel: state / done flag / actions in registers
            const int m = tid & 15, part = tid >> 4;
            const int P = hP;
            // head pre-activations of trajectory m: the four waves' partials, summed in the order of SplitPolicy::run's own combine
            const float* sp = s_part + m * 4;
            const float y0 = ((sp[0] + sp[TB * 4]) + (sp[2 * TB * 4] + sp[3 * TB * 4])) + s_bo[0];
            const float y1 = ((sp[1] + sp[TB * 4 + 1]) + (sp[2 * TB * 4 + 1] + sp[3 * TB * 4 + 1])) + s_bo[1];
            const float th0 = fast_tanh(y0), th1 = fast_tanh(y1);
            const float steer = wrap_action(fc0, fc0.sc * th0 + fc0.of), ax = wrap_action(fc1, fc1.sc * th1 + fc1.of);
            const float dflag = fdone;
            DBG_TICK(7)
            GLOBAL_AS f32x4* er = gptr(reinterpret_cast<f32x4*>(hst_env + (row0 + tid) * ENV_STASH));
            if (hng && tid < TB) {   // env stash row: tanh outputs + wrapped (steer, a_x), done_t, state_t
                const f32x4 e0 = {th0, th1, steer, ax}, e1 = {dflag, fs[0], fs[1], fs[2]}, e2 = {fs[3], fs[4], fs[5], 0.f};
                er[0] = e0;
                er[1] = e1;
                er[2] = e2;
            }
            DBG_TICK(4)
            float sn[6];
            VehStep w;
            w.sphi = veh_s; w.cphi = veh_c;
            veh_f_xu(VC, fs, steer, ax, sn, w);
            if (part == 0) {
                float o[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) o[i] = xs[m * ldx + i];
                r = veh_reward(o, steer, ax);
            }
            const float s_old = veh_s, c_old = veh_c;
            sincosf(sn[2], &veh_s, &veh_c);                 // also next step's f_xu heading terms
            if (hng && tid < TB) {   // 4th quad of the env stash row: the backward sweep reuses both sin / cos pairs
                const f32x4 e3 = {s_old, c_old, veh_s, veh_c};
                er[3] = e3;
            }
            const float cn = veh_c, snn = -veh_s;           // cos(-phi'), sin(-phi')
            const f32x4* tbl = s_ref + m * TL + (t + 1);
            {   // point 0: the base model's done test reads its transform - every part evaluates it (the flag lives in registers)
                const f32x4 rp = tbl[0];
                const float dx = rp[0] - sn[0], dy = rp[1] - sn[1];
                const float xtf = dx * cn - dy * snn;
                const float ytf = dx * snn + dy * cn;
                const float ptf = angle_normalize(rp[2] - sn[2]);
                const float utf = rp[3] - sn[3];
                done_m = (fabsf(xtf) > 10.f) || (fabsf(ytf) > 10.f) || (fabsf(ptf) > 3.14159265358979323846f);
                if (part == 0 && dflag == 0.f) {
                    xs[m * ldx + 0] = xtf; xs[m * ldx + 1] = ytf; xs[m * ldx + 2] = ptf;
                    xs[m * ldx + 3] = utf; xs[m * ldx + 4] = sn[4]; xs[m * ldx + 5] = sn[5];
                }
            }
            for (int j = (part == 0) ? 16 : part; j <= P; j += 16) {
                const f32x4 rp = tbl[j];
                const float dx = rp[0] - sn[0], dy = rp[1] - sn[1];
                const float xtf = dx * cn - dy * snn;
                const float ytf = dx * snn + dy * cn;
                const float ptf = angle_normalize(rp[2] - sn[2]);
                const float utf = rp[3] - sn[3];
                if (dflag == 0.f) {
                    float* dst = xs + m * ldx + 6 + 4 * (j - 1);
                    dst[0] = xtf; dst[1] = ytf; dst[2] = ptf; dst[3] = utf;
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) fs[i] = sn[i];
        } else {   // GOPS_ENV_VEH3DOFCONTI: all 256 threads, thread = (trajectory m, part)
            const int m = tid & 15, part = tid >> 4;
            const int P = p.env.pre_horizon;
            float s[6], sn[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) s[i] = s_state[m * 8 + i];
            const float steer = s_act[m * 4 + 0], ax = s_act[m * 4 + 1];
            const float dflag = s_done[m];
            VehStep w;
            w.sphi = veh_s; w.cphi = veh_c;
            veh_f_xu(VC, s, steer, ax, sn, w);
#ifdef GOPS_DUMP
            float* dmpe = (SS && p.dbg != nullptr) ? reinterpret_cast<float*>(p.dbg) + ((((size_t)tile * p.H + t) * NTHREADS + tid) << 6) : nullptr;
            if (dmpe != nullptr) {
                gptr(dmpe)[1] = steer; gptr(dmpe)[2] = ax; gptr(dmpe)[3] = dflag;
#pragma unroll
                for (int i = 0; i < 6; ++i) { gptr(dmpe)[4 + i] = s[i]; gptr(dmpe)[10 + i] = sn[i]; }
            }
#endif
            float pen_c = 0.f;   // surrcstr_penalty: constraint of the CURRENT pose
            float err_c0 = 0.f, err_c1 = 0.f;   // errcstr: |delta_y| - tol, |delta_u| - tol of the current observation
            if (part == 0) {
                float o[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) o[i] = xs[m * ldx + i];
                r = SURR ? veh_reward_w(p.env.reward_w, o, steer, ax) : veh_reward(o, steer, ax);
                if constexpr (SURR) {
                    if (p.env.cstr_err) {   // errcstr model: constraints on the tracking errors of the CURRENT observation
                        err_c0 = fabsf(o[1]) - p.env.err_tol[0];
                        err_c1 = fabsf(o[3]) - p.env.err_tol[1];
                    }
                    if (p.env.surr_penalty && m < nvalid) {   // collision penalty on the CURRENT pose and surrounding vehicle
                        const f32x4 cur = gptr(p.surr_table)[((size_t)(b0 + m) * (p.H + 1) + t) * p.env.n_surr];
                        SurrCstr sc;
                        surr_constraint<false>(p.env, s[0], s[1], veh_s, veh_c, &cur, sc);
                        float dummy;
                        pen_c = sc.c[0];
                        r -= surr_penalty(pen_c, dummy);
                    }
                }
            }
            __syncthreads();   // every read of the old obs / state is done
            const float s_old = veh_s, c_old = veh_c;
            const float s_old_s = s_old, s_old_c = c_old;   // sin / cos of the CURRENT heading (surrcstr_penalty observation)
            sincosf(sn[2], &veh_s, &veh_c);                 // also next step's f_xu heading terms
            if (p.need_grad && tid < TB) {   // 4th quad of the env stash row: the backward sweep reuses both sin / cos pairs
                const f32x4 e3 = {s_old, c_old, veh_s, veh_c};
                gptr(reinterpret_cast<f32x4*>(p.st.env + (row0 + tid) * ENV_STASH))[3] = e3;
            }
            const float cn = veh_c, snn = -veh_s;           // cos(-phi'), sin(-phi')
            const f32x4* tbl = s_ref + m * TL + (t + 1);
            for (int j = part; j <= P; j += 16) {
                const f32x4 rp = tbl[j];
                const float dx = rp[0] - sn[0], dy = rp[1] - sn[1];
                const float xtf = dx * cn - dy * snn;
                const float ytf = dx * snn + dy * cn;
                const float ptf = angle_normalize(rp[2] - sn[2]);
                const float utf = rp[3] - sn[3];
#ifdef GOPS_DUMP
                if (dmpe != nullptr && j == part) {
                    gptr(dmpe)[16] = veh_s; gptr(dmpe)[17] = veh_c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) gptr(dmpe)[18 + i] = rp[i];
                    gptr(dmpe)[22] = xtf; gptr(dmpe)[23] = ytf; gptr(dmpe)[24] = ptf; gptr(dmpe)[25] = utf;
                }
#endif
                if (j == 0) {
                    done_m = (fabsf(xtf) > 10.f) || (fabsf(ytf) > 10.f) || (fabsf(ptf) > 3.14159265358979323846f);
                    if (SURR && p.env.surr_penalty) done_m = false;   // judge_done of the penalty model: never (:236-246)
                    if (dflag == 0.f) {
                        xs[m * ldx + 0] = xtf; xs[m * ldx + 1] = ytf; xs[m * ldx + 2] = ptf;
                        xs[m * ldx + 3] = utf; xs[m * ldx + 4] = sn[4]; xs[m * ldx + 5] = sn[5];
                    }
                } else if (dflag == 0.f) {
                    f32x4 ov = {xtf, ytf, ptf, utf};
                    float* dst = xs + m * ldx + 6 + 4 * (j - 1);
                    dst[0] = ov[0]; dst[1] = ov[1]; dst[2] = ov[2]; dst[3] = ov[3];
                }
            }
            if (part == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) s_state[m * 8 + i] = sn[i];
            }
            if constexpr (SURR) {
                // surrounding vehicles at the new time: observation columns (x, y, phi, u)_surr - (x, y, phi, u)_ego and
                // the constraint on the new ego pose; info["constraint"] is NOT masked at done (the algorithms sum it
                // for every trajectory), the observation is
                if (part == 0 && m < nvalid) {
                    const int ns = p.env.n_surr;
                    const GLOBAL_AS f32x4* sp = gptr(p.surr_table) + ((size_t)(b0 + m) * (p.H + 1) + (t + 1)) * ns;
                    f32x4 pts[GOPS_MAX_SURR];
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_SURR; ++i) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        pts[i] = (i < ns) ? sp[i] : z;
                    }
                    if (dflag == 0.f) {
                        if (p.env.surr_penalty) {   // NEXT surrounding vehicle in the ego frame of the CURRENT state (:117-125)
                            float* dst = xs + m * ldx + 6 + 4 * P;
                            const float dx = pts[0][0] - s[0], dy = pts[0][1] - s[1];
                            dst[0] = dx * s_old_c - dy * (-s_old_s);
                            dst[1] = dx * (-s_old_s) + dy * s_old_c;
                            dst[2] = angle_normalize(pts[0][2] - s[2]);
                            dst[3] = pts[0][3] - s[3];
                        } else {
#pragma unroll
                            for (int i = 0; i < GOPS_MAX_SURR; ++i)
                                if (i < ns) {
                                    float* dst = xs + m * ldx + 6 + 4 * P + 4 * i;
                                    dst[0] = pts[i][0] - sn[0]; dst[1] = pts[i][1] - sn[1];
                                    dst[2] = pts[i][2] - sn[2]; dst[3] = pts[i][3] - sn[3];
                                }
                        }
                    }
                    SurrCstr sc;
                    surr_constraint<false>(p.env, sn[0], sn[1], veh_s, veh_c, pts, sc);
                    // the penalty model fills info["constraint"] before its info dict is updated (:131-139): CURRENT pose
                    if (p.env.surr_penalty) sc.c[0] = pen_c;
                    if (p.env.cstr_err) { sc.c[0] = err_c0; sc.c[1] = err_c1; }
                    float e2 = 0.f, e1 = 0.f, lg = 0.f;
#pragma unroll
                    for (int k = 0; k < GOPS_MAX_CONSTRAINT; ++k) {
                        if (k < p.env.n_constraint) {
                            const float cp = fmaxf(sc.c[k], 0.f), cm = fminf(sc.c[k], 0.f);
                            e2 += cp * cp;
                            e1 += cp;
                            lg += logf(-cm + 1e-8f);
                            if (!(sc.c[k] < 0.f)) c_feas = 0.f;
                            float dlog;
                            c_mul[k] *= spil_phi(sc.c[k], dlog);
                            if (!(sc.c[k] <= 0.f)) c_safe[k] = 0.f;
                            if (p.out.constraints != nullptr)
                                gptr(p.out.constraints)[((size_t)t * p.B + b0 + m) * p.env.n_constraint + k] = sc.c[k];
                        }
                    }
                    c_ext += e2 * p.gpow[t];
                    c_lin += e1 * p.gpow[t];
                    c_int += lg * p.gpow[t];
                }
            }
        }
        if (tid < TB) {
            const float d = FASTV ? fdone : s_done[tid];
            float rr = (d != 0.f) ? 0.f : r;
            if (ENV != GOPS_ENV_NONE && hshaping) rr = (rr + hrshift) * hrscale;
            v_acc += rr * gpow_t;
#ifdef GOPS_DUMP
            if (SS && VEH && p.dbg != nullptr) {
                float* dm = reinterpret_cast<float*>(p.dbg) + ((((size_t)tile * p.H + t) * NTHREADS + tid) << 6);
                gptr(dm)[26] = rr; gptr(dm)[27] = v_acc;
            }
#endif
            if (hrewards != nullptr && tid < nvalid) gptr(hrewards)[(size_t)t * hB + b0 + tid] = rr;
            if (!FASTV && done_m && !hnomask) s_done[tid] = 1.f;
        }
        if constexpr (FASTV) {
            if (done_m && !hnomask) fdone = 1.f;   // (every part evaluated the done test of its trajectory)
        }
        DBG_TICK(5)
    }
    if constexpr (FASTV) {   // back to the LDS copies the code behind the loop reads
        if (tid < TB) {
            s_done[tid] = fdone;
#pragma unroll
            for (int i = 0; i < 6; ++i) s_state[tid * 8 + i] = fs[i];
        }
    }
    __syncthreads();
    if (p.env.no_mask_at_done && !SURR && tid < TB) {
        // mask_at_done = False: nothing was frozen or masked on the way; the rollout's done is the base model's test on
        // the LAST next state, which is the final observation (every done test of these models reads the new state only)
        const float* o = xs + tid * ldx;
        const float pi = 3.14159265358979323846f;
        bool dl = false;
        if (ENV == GOPS_ENV_IDPENDULUM) {
            float s6[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) s6[i] = obs_unscale(p.env, i, o[i]);
            dl = idp_done(IC, s6);
        } else if (ENV == GOPS_ENV_CARTPOLE) {
            const CartConst C = cart_const();
            const float x0 = obs_unscale(p.env, 0, o[0]), x2 = obs_unscale(p.env, 2, o[2]);
            dl = (x0 < -C.xth) || (x0 > C.xth) || (x2 < -C.thth) || (x2 > C.thth);
        } else if (ENV == GOPS_ENV_VEH3DOFCONTI) {
            dl = (fabsf(o[0]) > 10.f) || (fabsf(o[1]) > 10.f) || (fabsf(o[2]) > pi);
        } else if (ENV == GOPS_ENV_VEH2DOF) {
            dl = (fabsf(o[0]) > 2.f) || (fabsf(o[1]) > pi);
        } else if (ENV == GOPS_ENV_MOBILEROBOT) {
            const MobConst MC = mob_const();
            const float dx = o[8] - o[0], dy = o[9] - o[1];
            dl = (o[0] < -2.f) || (fabsf(o[1]) > 4.f) || (MC.safe_dis - sqrtf(dx * dx + dy * dy) > MC.margin);
        }
        s_done[tid] = dl ? 1.f : 0.f;
    }

    if (TAIL) {   // v += (~done_H) * gamma^H * V_target(obs_H)   (infadp.py:182-184, 210)
        if (p.fh && tid < TB) xs[tid * ldx + O] = 0.f;
        for (int j = 0; j < p.val.nl - 1; ++j)   // the policy biases are no longer needed
            for (int n = tid; n < p.val.dims[j + 1]; n += NTHREADS) s_bias[j * ldh + n] = gptr(p.val.b[j])[n];
        __syncthreads();
        float y[GOPS_MAX_ACT];
        const int Lv = p.val.nl - 1;
        if (SS && p.tail_fp32) {
            // relu / selu value net of a launch that keeps a gradient: exact fp32 products for THIS net (dV/d(obs) of a piecewise-linear
            // net jumps where a pre-activation changes sign under the 22-bit plane representation); the step loop stays plane-split
            float* hcur = mlp_hidden_forward(p.val, NoW{}, NoW{}, xs, ldx, ha, hb, ldh, tid, s_bias,
                                             p.need_grad ? p.st.tail_h : nullptr,
                                             p.need_grad ? p.st.tail_z : nullptr, (size_t)b0, dbg);
            mlp_head<false>(gptr(p.val.w[Lv]), p.val.dims[Lv], gptr(p.val.b[Lv]), p.val.dims[Lv], 1, hcur, ldh, tid, y);
        } else
        if constexpr (SS) {   // the value net on the same streamed plane-split routine: its head (one output) staged like the policy's
            const int Kv = p.val.dims[Lv];
            for (int idx = tid; idx < GOPS_MAX_ACT * ldh; idx += NTHREADS) {
                const int k = idx >> 2, a = idx & 3;
                s_wo[idx] = (a == 0 && k < Kv) ? gptr(p.val.w[Lv])[k] : 0.f;
            }
            if (tid < GOPS_MAX_ACT) s_bo[tid] = (tid == 0) ? gptr(p.val.b[Lv])[0] : 0.f;
            __syncthreads();
            y[0] = ss_net_forward<1>(p.val, p.ssv, xs, ldx, xq, hq, s_part, s_bias, ldh, s_wo, s_bo, tid,
                                     p.need_grad ? p.st.tail_h : nullptr, p.need_grad ? p.st.tail_z : nullptr, (size_t)b0, dbg, ss_ovf);
        } else
        if constexpr (F16) {
            convert_x_h(xs, ldx, p.val.kp[0], p.val.kp32[0], x16, ldx16, nullptr, 0, tid);
            __syncthreads();
            const _Float16* hcur = mlp_hidden_forward_h(p.val, x16, ldx16, reinterpret_cast<_Float16*>(ha),
                                                        reinterpret_cast<_Float16*>(hb), ld16, tid, s_bias, ldh,
                                                        p.need_grad ? p.st.tail_h : nullptr,
                                                        p.need_grad ? p.st.tail_z : nullptr, (size_t)b0, nvalid);
            mlp_head_h<false>(gptr(p.val.w[Lv]), p.val.dims[Lv], gptr(p.val.b[Lv]), p.val.dims[Lv], 1, hcur, ld16, tid, y);
        } else {
            float* hcur = mlp_hidden_forward(p.val, NoW{}, NoW{}, xs, ldx, ha, hb, ldh, tid, s_bias,
                                             p.need_grad ? p.st.tail_h : nullptr,
                                             p.need_grad ? p.st.tail_z : nullptr, (size_t)b0, dbg);
            mlp_head<false>(gptr(p.val.w[Lv]), p.val.dims[Lv], gptr(p.val.b[Lv]), p.val.dims[Lv], 1, hcur, ldh, tid, y);
        }
        if ((tid & 15) == 0) s_th[(tid >> 4) * 4] = y[0];
        __syncthreads();
        if (tid < TB) v_acc += ((p.tail_unmasked ? 1.f : 1.f - s_done[tid]) * p.gpow[p.H]) * s_th[tid * 4];
    }

    if (CSTR && tid < nvalid && p.out.constraint_sums != nullptr) {
        GLOBAL_AS float* cs = gptr(p.out.constraint_sums) + b0 + tid;
        cs[0] = c_ext; cs[(size_t)p.B] = c_lin; cs[(size_t)2 * p.B] = c_int; cs[(size_t)3 * p.B] = c_feas;
    }
    if (CSTR && tid < nvalid && p.out.constraint_prods != nullptr) {
        GLOBAL_AS float* cp = gptr(p.out.constraint_prods) + b0 + tid;
        const int nc = p.env.n_constraint;
#pragma unroll
        for (int k = 0; k < GOPS_MAX_CONSTRAINT; ++k)
            if (k < nc) { cp[(size_t)k * p.B] = c_mul[k]; cp[(size_t)(nc + k) * p.B] = c_safe[k]; }
    }
    if constexpr (SPLIT || SS) {
        // A value beyond the half range of the plane images (common.h SPLIT_FWD_SA: |a| >= 1.05e6) made part of this tile's contractions
        // non-finite - which relu / tanh can hide again: the tile's returns are poisoned instead, so that the failure is loud
        unsigned o = ss_ovf;
        if constexpr (SPLIT) { o = SP.ovf; SP.ovf = 0; }
        if (split_overflow_any(o, s_part, tid)) {   // (s_part: the head partials are consumed by now)
            v_acc = __builtin_nanf("");
            // ... and the launch is marked (RolloutParams::gscale[3], re-armed by the next forward's prologue): a backward after this
            // forward returns NaN for EVERY gradient element - its stash is fp32 and finite, but it belongs to a rollout that went wrong
            if (tid == 0 && p.gscale != nullptr) atomicOr(reinterpret_cast<unsigned*>(p.gscale) + 3, 1u);
        }
    }
    if (tid < nvalid) {
        gptr(p.out.v_pi)[b0 + tid] = v_acc;
        if (p.out.final_done != nullptr) gptr(p.out.final_done)[b0 + tid] = s_done[tid];
        if (p.need_grad && p.st.tail_done != nullptr) gptr(p.st.tail_done)[b0 + tid] = s_done[tid];
    }
    if (p.out.final_obs != nullptr) {
        for (int idx = tid; idx < TB * O; idx += NTHREADS) {
            const int m = idx / O, c = idx - m * O;
            if (m < nvalid) gptr(p.out.final_obs)[(size_t)(b0 + m) * O + c] = xs[m * ldx + c];
        }
    }
    if (VEH2 && p.out.final_state != nullptr && tid < TB * 4) {
        const int m = tid >> 2, c = tid & 3;
        if (m < nvalid) gptr(p.out.final_state)[(size_t)(b0 + m) * 4 + c] = s_state[m * 8 + c];
    }
    if (VEH && p.out.final_state != nullptr && tid < TB * 6) {
        const int m = tid / 6, c = tid - m * 6;
        if (m < nvalid) gptr(p.out.final_state)[(size_t)(b0 + m) * 6 + c] = s_state[m * 8 + c];
    }
    if constexpr (SPLIT && MULTI) __syncthreads();   // the next tile's set-up overwrites xs / s_state / s_done / s_ref
    } while (SPLIT && MULTI && (tile += gridDim.x) < ntiles);
    DBG_TICK(15)   // (everything behind the last step: final outputs)
    dbg.dump(p.dbg);
}

// split_k0: input width (multiple of 32) of hidden layer 0 when the plane-split kernel runs, else 0
// ss: the streamed-split forward (no residual plane in LDS whatever split_k0 is)
size_t rollout_fwd_lds_bytes(int ldx, int ldh, int ref_points, bool f16, int split_k0, bool ss = false) {
    size_t b = sizeof(float) * (size_t)(TB * ldx + 2 * hidden_tile_floats(ldh, f16) + TB * (8 + 4 + 4 + 1 + 1 + 2) + (4 + GOPS_MAX_LAYERS - 1) * ldh +
                                        4 * TB * ref_points);
    if (f16) b += sizeof(_Float16) * (size_t)TB * ((((ldx - 4) + 31) & ~31) + 8);   // x16
    if (split_k0 > 0) {   // plane images of X and H_1, head partials, layer 0's residual plane; no fp32 hidden tiles of their own
        b += (size_t)4 * TB * (split_rowb(split_k0) + split_rowb(256)) + sizeof(float) * 4 * TB * 4 +
             ((ss || split_k0 / 32 > SPLIT_MAX_RESIDENT_KC0) ? 0 : (size_t)(split_k0 / 32) * 16384);   // (wider inputs: W_0's planes stream from L2)
        b -= sizeof(float) * 2 * (size_t)hidden_tile_floats(ldh, false);
    }
    return b;
}
size_t rollout_bwd_lds_bytes(int ldx, int ldh, int ref_points, bool f16, bool split, bool ssb = false);

static int device_cus() {
    static int n_cu = 0;
    if (n_cu == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return n_cu;
}

int split_grid_limit() { return device_cus(); }

// Plane-split contractions (common.h SplitDev) replace the fp32 MFMAs of the register-stationary kernels for a closed-loop
// obs -> 256 -> 256 -> act policy on the BASELINE env kinds with an input of at most 128 columns (4 chunks of 32: the
// planes of both layers then fit the register file + LDS).  One workgroup per CU keeps the weights resident and walks
// the tiles grid-stride, whatever the batch size.  GOPS_SPLIT=0 keeps the fp32-MFMA kernels.
// Activations whose derivative jumps at 0 (relu, selu) in a launch with a tail value net that keeps a gradient: the gradient runs
// through dV/d(obs_H) of a piecewise-linear net, and every pre-activation of THAT net which changes sign under the 22-bit plane
// representation moves it by a finite amount - measured at cfg3 (relu, 256^3, B = 8192): 2.0e-4 from the reference with a
// plane-split tail value net, < 1e-4 with exact fp32 products.  The tail value net of such a launch is therefore evaluated with
// exact fp32 products (the stationary kernels do that for every tail; the streamed ones on RolloutParams.tail_fp32).  Launches
// without a gradient are not concerned: the VALUES are continuous in the weights.
static bool kinked_with_tail(const RolloutParams& p) {
    auto kinked = [](int a) { return a == GOPS_ACT_RELU || a == GOPS_ACT_SELU; };
    return p.need_grad && p.tail && (kinked(p.pol.act) || kinked(p.val.act));
}

bool split_eligible(const RolloutParams& p) {
    const MlpDev& M = p.pol;
    if (p.f16 || p.ext || p.open_loop || p.env.repeat_num > 1) return false;
    // (relu / selu nets with a tail value net are fine here: these kernels evaluate every tail value net - forward and input
    // adjoint - with exact fp32 products)
    if (p.env.kind != GOPS_ENV_LQ && p.env.kind != GOPS_ENV_IDPENDULUM && p.env.kind != GOPS_ENV_VEH3DOFCONTI) return false;
    if (M.nl != 3 || M.dims[1] != 256 || M.dims[2] != 256 || p.ldh != 260 || M.kp32[0] > 256 || p.ldx != M.kp[0] + 4) return false;
    // more than 128 inputs (veh3dofconti with P > 30): layer 0's planes stream from L2 - instantiated without the tail value net
    if (M.kp32[0] > 128 && (p.env.kind != GOPS_ENV_VEH3DOFCONTI || p.tail || (p.vflags & GOPS_VF_NO_SPLIT_STREAM0) != 0)) return false;
    // More tiles than CUs AND a tail value net: the tail is evaluated per tile with fp32 weights streamed from L2 by the one
    // resident workgroup, which exposes every L2 round trip (measured at B = 65536: no faster than the streamed kernels with
    // their three workgroups per CU) - those launches stay on the streamed kernels.
    if (p.tail && (p.B + TB - 1) / TB > device_cus() && !(p.vflags & GOPS_VF_SPLIT_TAIL_MULTI)) return false;
    if (p.vflags & (GOPS_VF_NO_STATIONARY_SPLIT | GOPS_VF_STREAMED_FP32 | GOPS_VF_STREAM_LAYER0)) return false;
    const int ref_pts = env_has_ref_table(p.env.kind) ? p.env.pre_horizon + 1 + p.H : (p.env.kind == GOPS_ENV_IDPENDULUM ? IDP_POINTS(true) : 0);
    if (rollout_fwd_lds_bytes(p.ldx, p.ldh, env_has_ref_table(p.env.kind) ? ref_pts : 0, false, M.kp32[0]) > 160 * 1024) return false;
    if (rollout_bwd_lds_bytes(p.ldx, p.ldh, ref_pts, false, true) > 160 * 1024) return false;
    return true;
}

// Streamed-split forward kernels: every hidden layer of the policy (and of the tail value net) 256 wide, at most 256 padded
// inputs, fp32, closed loop, every env model - the launches the register-stationary kernels do not take (three
// hidden layers, a tail value net with more tiles than CUs).  The backward sweep of such a launch: rollout_bwd.hip ssb_eligible
// (both forward variants write the same feature-major stash).  GOPS_VF_NO_STREAMED_SPLIT_FWD switches it off.
static bool ss_shape_ok(const RolloutParams& p) {
    if (p.f16 || p.ext || p.open_loop || p.env.repeat_num > 1) return false;
    // (value / MLP batches, GOPS_ENV_NONE: one step - half the MFMA time of the fp32 kernels; GOPS_SS_VALUE=0 keeps those)
    if (p.env.kind == GOPS_ENV_NONE && (p.vflags & GOPS_VF_NO_STREAMED_SPLIT_VALUE)) return false;
    if (p.vflags & (GOPS_VF_NO_STREAMED_SPLIT_FWD | GOPS_VF_STREAMED_FP32 | GOPS_VF_STREAM_LAYER0)) return false;
    auto net_ok = [](const MlpDev& M) {
        if (M.nl < 3 || M.kp32[0] > 256) return false;
        for (int j = 1; j < M.nl; ++j)
            if (M.dims[j] != 256) return false;
        return true;
    };
    if (!net_ok(p.pol) || p.ldh != 260 || (p.tail && !net_ok(p.val))) return false;
    const int k0 = 32 * std::max(ss_kc0(p.pol.kp32[0]), p.tail ? ss_kc0(p.val.kp32[0]) : 0);
    const int ref_pts = env_has_ref_table(p.env.kind) ? p.env.pre_horizon + 1 + p.H : 0;
    return rollout_fwd_lds_bytes(p.ldx, p.ldh, ref_pts, false, k0, true) + (env_in_lds(p.env.kind, true) ? 4 * ENV_LDS_FLOATS : 0) <= 80 * 1024;   // two workgroups per CU
}
// relu / selu with a tail value net, gradient kept (kinked_with_tail above): the step loop (policy net, env model: a relu POLICY alone is
// indifferent, 7.7e-6 vs 7.4e-6 at the target shape) is plane-split like any other launch, only the TAIL value net keeps exact fp32
// products (p.tail_fp32); the sweep is linear once the forward has fixed the activation pattern: plane-split.
bool ss_eligible(const RolloutParams& p) { return ss_shape_ok(p); }
bool ss_tail_exact(const RolloutParams& p) { return kinked_with_tail(p); }

// Picks the register-stationary variant when the policy is (kp0 in {16,48,128}) -> 256 -> 256 ...,
// else the fully streamed kernel.  sk[0] / sk[1] receive the chosen chunk counts (0 = streamed).
void rollout_variant(const RolloutParams& p, int sk[2], bool backward) {
    sk[0] = sk[1] = 0;
    const MlpDev& M = p.pol;
    if (p.f16) return;   // the half-precision path streams its (half as large) weights from L2
    if (p.env.repeat_num > 1 || p.ext) return;   // GEN / EXT instantiations exist for the streamed kernels only
    if (p.env.kind >= GOPS_ENV_VEH3DOF_SURR) return;   // constrained / gym-style models: streamed kernels only
    // Register-stationary weights pin one workgroup per CU.  That is the right trade only while there
    // is at most one tile per CU (B <= 16 * #CUs = 4096 on MI355X); with more tiles the streamed
    // kernels win because 2-3 workgroups per CU overlap each other's MFMA and VALU phases.
    if ((p.B + TB - 1) / TB > device_cus() && !(p.vflags & GOPS_VF_STATIONARY_ANY_BATCH)) return;
    if (M.nl - 1 < 2 || M.dims[1] != 256 || M.dims[2] != 256 || p.env.kind == GOPS_ENV_NONE || p.ldh != 260) return;
    sk[1] = 16;
    const int k0 = M.kp[0] >> 4;
    if (p.ldx == M.kp[0] + 4) {
        // veh3dofconti: 3 chunks (P = 10) fully stationary; from 6 chunks up, the first 6 stay in
        // registers and the rest streams (P = 30: 6 + 2, P = 50: 6 + 7)
        if (p.env.kind == GOPS_ENV_VEH3DOFCONTI) sk[0] = (k0 == 3) ? 3 : (k0 >= 6 ? 6 : 0);
        if ((p.env.kind == GOPS_ENV_LQ || p.env.kind == GOPS_ENV_IDPENDULUM) && k0 == 1) sk[0] = 1;
    }
    // The backward sweep's VALU phases need more than the 128 VGPRs left beside 384 weight registers:
    // it keeps only the 256-register layer-1 fragments (all in AGPRs) and streams layer 0.
    // Backward: sk[0] counts stationary K-chunks (of 16) of the delta_1 -> g_x GEMM.  kp0 = 128: 12 of
    // them (2 n-tiles per wave; 14 spills); kp0 = 16 (lq): all 16 (1 tile, wave 0); anything else streams.
    // The backward's stationary variants also stage this step's H_2 / H_1 tiles in LDS: exactly two hidden layers.
    if (backward && M.nl != 3) { sk[0] = sk[1] = 0; return; }
    if (backward) sk[0] = (sk[1] == 16 && M.kp[0] == 128) ? 12 : ((sk[1] == 16 && M.kp[0] == 16) ? 16 : 0);
    // tuning flags (benchmarks only): the plain streamed kernels, or layer 1 stationary only
    if (p.vflags & GOPS_VF_STREAM_LAYER0) sk[0] = 0;
    if (p.vflags & GOPS_VF_STREAMED_FP32) sk[0] = sk[1] = 0;
}

#define LAUNCH_FWD(ENV, A, B)                                                                            \
    do {                                                                                                 \
        if (p.tail) launch_with_lds(rollout_fwd_kernel<ENV, A, B, true>, grid, block, lds, stream, dp);   \
        else launch_with_lds(rollout_fwd_kernel<ENV, A, B, false>, grid, block, lds, stream, dp);         \
    } while (0)
// the plain streamed fp32 kernel, or its obs -> 64 -> 64 -> act form (RolloutParams.narrow == 2: mlp_hidden_forward_n64)
#define LAUNCH_FWD_PLAIN(ENV)                                                                                                              \
    do {                                                                                                                                   \
        if (p.narrow == 2) {                                                                                                               \
            if (p.tail) launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, true, false, false, false, false, false, true>, grid, block, lds, stream, dp);  \
            else launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, false, false, false, false, false, false, true>, grid, block, lds, stream, dp);        \
        } else LAUNCH_FWD(ENV, 0, 0);                                                                                                      \
    } while (0)
#define LAUNCH_FWD_H(ENV)                                                                                          \
    do {                                                                                                           \
        if (p.tail) launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, true, true>, grid, block, lds, stream, dp);       \
        else launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, false, true>, grid, block, lds, stream, dp);             \
    } while (0)

// `p` is the host copy (for shape dispatch), `dp` the device copy the kernel reads.
hipError_t launch_rollout_fwd_h64(const RolloutParams& p, const RolloutParams* dp, hipStream_t stream);   // rollout_h64.hip
hipError_t launch_rollout_fwd(const RolloutParams& p, const RolloutParams* dp, hipStream_t stream) {
#ifdef GOPS_ONLY_NARROW   // the same for the plain streamed fp32 kernel of pyth_idpendulum (cfg1, the example scripts' shapes): EXTRA=-DGOPS_ONLY_NARROW
#if GOPS_ONLY_NARROW == 2
    launch_with_lds(rollout_fwd_kernel<GOPS_ENV_IDPENDULUM, 0, 0, false, false, false, false, false, false, true>, dim3((p.B + TB - 1) / TB), dim3(NTHREADS), 4 * ((size_t)p.narrow_off_fwd + p.narrow_floats), stream, dp);
#else
    launch_with_lds(rollout_fwd_kernel<GOPS_ENV_IDPENDULUM, 0, 0, false>, dim3((p.B + TB - 1) / TB), dim3(NTHREADS), 4 * ((size_t)p.narrow_off_fwd + p.narrow_floats), stream, dp);
#endif
    return hipGetLastError();
#elif defined(GOPS_ONLY_TARGET)   // register / ISA studies (EXTRA=-DGOPS_ONLY_TARGET tools/kernel_regs.sh rollout_fwd.hip): ONE instantiation, seconds to compile
    launch_with_lds(rollout_fwd_kernel<GOPS_ENV_VEH3DOFCONTI, 4, 8, false, false, false, true>, dim3(1), dim3(NTHREADS), 0, stream, dp);
    return hipGetLastError();
#else
    if (p.h64) return launch_rollout_fwd_h64(p, dp, stream);   // half precision, 64-trajectory tiles
    const dim3 grid((p.B + TB - 1) / TB), block(NTHREADS);
    size_t lds = rollout_fwd_lds_bytes(p.ldx, p.ldh, env_has_ref_table(p.env.kind) ? p.env.pre_horizon + 1 + p.H : 0, p.f16 != 0,
                                       p.sp.on ? 32 * p.sp.kc[0] : 0);
    if (p.narrow) lds = 4 * ((size_t)p.narrow_off_fwd + p.narrow_floats);   // (api.hip: only ever set for the plain streamed fp32 kernels)
    int sk[2];
    rollout_variant(p, sk, false);
    const int key = sk[0] * 100 + sk[1];
    if (p.sp.on) {   // plane-split stationary kernels: layer 0 in KC0 chunks of 32 inputs
#define LAUNCH_FWD_SPLIT(ENV, KC0)                                                                                           \
    do {                                                                                                                     \
        if (multi) {                                                                                                         \
            if (p.tail) launch_with_lds(rollout_fwd_kernel<ENV, KC0, 8, true, false, false, true, true>, grid, block, lds, stream, dp);  \
            else launch_with_lds(rollout_fwd_kernel<ENV, KC0, 8, false, false, false, true, true>, grid, block, lds, stream, dp);        \
        } else if (p.tail) launch_with_lds(rollout_fwd_kernel<ENV, KC0, 8, true, false, false, true>, grid, block, lds, stream, dp);  \
        else launch_with_lds(rollout_fwd_kernel<ENV, KC0, 8, false, false, false, true>, grid, block, lds, stream, dp);        \
    } while (0)
        const int kc0 = p.sp.kc[0];
        const dim3 grid(std::min<int>((p.B + TB - 1) / TB, device_cus()));   // one workgroup per CU, grid-stride over the tiles
        const bool multi = (p.B + TB - 1) / TB > device_cus();
        if (p.env.kind == GOPS_ENV_LQ && kc0 == 1) LAUNCH_FWD_SPLIT(GOPS_ENV_LQ, 1);
        else if (p.env.kind == GOPS_ENV_IDPENDULUM && kc0 == 1) LAUNCH_FWD_SPLIT(GOPS_ENV_IDPENDULUM, 1);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && kc0 == 2) LAUNCH_FWD_SPLIT(GOPS_ENV_VEH3DOFCONTI, 2);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && kc0 == 3) LAUNCH_FWD_SPLIT(GOPS_ENV_VEH3DOFCONTI, 3);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && kc0 == 4) LAUNCH_FWD_SPLIT(GOPS_ENV_VEH3DOFCONTI, 4);
#define LAUNCH_FWD_SPLIT_NOTAIL(ENV, KC0)                                                                                    \
    do {                                                                                                                     \
        if (multi) launch_with_lds(rollout_fwd_kernel<ENV, KC0, 8, false, false, false, true, true>, grid, block, lds, stream, dp);  \
        else launch_with_lds(rollout_fwd_kernel<ENV, KC0, 8, false, false, false, true>, grid, block, lds, stream, dp);       \
    } while (0)
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && kc0 == 5 && !p.tail) LAUNCH_FWD_SPLIT_NOTAIL(GOPS_ENV_VEH3DOFCONTI, 5);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && kc0 == 6 && !p.tail) LAUNCH_FWD_SPLIT_NOTAIL(GOPS_ENV_VEH3DOFCONTI, 6);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && kc0 == 7 && !p.tail) LAUNCH_FWD_SPLIT_NOTAIL(GOPS_ENV_VEH3DOFCONTI, 7);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && kc0 == 8 && !p.tail) LAUNCH_FWD_SPLIT_NOTAIL(GOPS_ENV_VEH3DOFCONTI, 8);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.ss) {   // streamed-split forward
        const int k0 = 32 * std::max(p.ssp.kc[0], p.tail ? p.ssv.kc[0] : 0);
        const size_t lds_ss = rollout_fwd_lds_bytes(p.ldx, p.ldh, env_has_ref_table(p.env.kind) ? p.env.pre_horizon + 1 + p.H : 0, false, k0, true) +
                              (env_in_lds(p.env.kind, true) ? 4 * ENV_LDS_FLOATS : 0);
#define LAUNCH_FWD_SS(ENV)                                                                                                        \
    do {                                                                                                                          \
        if (p.tail) launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, true, false, false, false, false, true>, grid, block, lds_ss, stream, dp);   \
        else launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, false, false, false, false, false, true>, grid, block, lds_ss, stream, dp);         \
    } while (0)
        switch (p.env.kind) {
            case GOPS_ENV_NONE: LAUNCH_FWD_SS(GOPS_ENV_NONE); break;
            case GOPS_ENV_LQ: LAUNCH_FWD_SS(GOPS_ENV_LQ); break;
            case GOPS_ENV_IDPENDULUM: LAUNCH_FWD_SS(GOPS_ENV_IDPENDULUM); break;
            case GOPS_ENV_VEH3DOFCONTI: LAUNCH_FWD_SS(GOPS_ENV_VEH3DOFCONTI); break;
            case GOPS_ENV_VEH3DOF_SURR: LAUNCH_FWD_SS(GOPS_ENV_VEH3DOF_SURR); break;
            case GOPS_ENV_CARTPOLE: LAUNCH_FWD_SS(GOPS_ENV_CARTPOLE); break;
            case GOPS_ENV_PENDULUM: LAUNCH_FWD_SS(GOPS_ENV_PENDULUM); break;
            case GOPS_ENV_VEH2DOF: LAUNCH_FWD_SS(GOPS_ENV_VEH2DOF); break;
            case GOPS_ENV_MOBILEROBOT: LAUNCH_FWD_SS(GOPS_ENV_MOBILEROBOT); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (p.f16) {   // half-precision kernels: weights streamed from L2, four workgroups per CU
        switch (p.env.kind) {
            case GOPS_ENV_NONE: LAUNCH_FWD_H(GOPS_ENV_NONE); break;
            case GOPS_ENV_LQ: LAUNCH_FWD_H(GOPS_ENV_LQ); break;
            case GOPS_ENV_IDPENDULUM: LAUNCH_FWD_H(GOPS_ENV_IDPENDULUM); break;
            case GOPS_ENV_VEH3DOFCONTI: LAUNCH_FWD_H(GOPS_ENV_VEH3DOFCONTI); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (p.env.repeat_num > 1) {   // ActionRepeatModel: the GEN instantiations (streamed)
#define LAUNCH_FWD_GEN(ENV)                                                                                        \
    do {                                                                                                           \
        if (p.tail) launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, true, false, true>, grid, block, lds, stream, dp);  \
        else launch_with_lds(rollout_fwd_kernel<ENV, 0, 0, false, false, true>, grid, block, lds, stream, dp);        \
    } while (0)
        switch (p.env.kind) {
            case GOPS_ENV_LQ: LAUNCH_FWD_GEN(GOPS_ENV_LQ); break;
            case GOPS_ENV_IDPENDULUM: LAUNCH_FWD_GEN(GOPS_ENV_IDPENDULUM); break;
            case GOPS_ENV_CARTPOLE: LAUNCH_FWD_GEN(GOPS_ENV_CARTPOLE); break;
            case GOPS_ENV_PENDULUM: LAUNCH_FWD_GEN(GOPS_ENV_PENDULUM); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (p.env.kind) {
        case GOPS_ENV_NONE: LAUNCH_FWD_PLAIN(GOPS_ENV_NONE); break;
        case GOPS_ENV_LQ:
            if (key == 116) LAUNCH_FWD(GOPS_ENV_LQ, 1, 16); else LAUNCH_FWD_PLAIN(GOPS_ENV_LQ);
            break;
        case GOPS_ENV_IDPENDULUM:
            if (key == 116) LAUNCH_FWD(GOPS_ENV_IDPENDULUM, 1, 16); else LAUNCH_FWD_PLAIN(GOPS_ENV_IDPENDULUM);
            break;
        case GOPS_ENV_VEH3DOFCONTI:
            if (key == 616) LAUNCH_FWD(GOPS_ENV_VEH3DOFCONTI, 6, 16);
            else if (key == 316) LAUNCH_FWD(GOPS_ENV_VEH3DOFCONTI, 3, 16);
            else if (key == 16) LAUNCH_FWD(GOPS_ENV_VEH3DOFCONTI, 0, 16);
            else LAUNCH_FWD_PLAIN(GOPS_ENV_VEH3DOFCONTI);
            break;
        case GOPS_ENV_VEH3DOF_SURR: LAUNCH_FWD_PLAIN(GOPS_ENV_VEH3DOF_SURR); break;
        case GOPS_ENV_CARTPOLE: LAUNCH_FWD_PLAIN(GOPS_ENV_CARTPOLE); break;
        case GOPS_ENV_PENDULUM: LAUNCH_FWD_PLAIN(GOPS_ENV_PENDULUM); break;
        case GOPS_ENV_VEH2DOF: LAUNCH_FWD_PLAIN(GOPS_ENV_VEH2DOF); break;
        case GOPS_ENV_MOBILEROBOT: LAUNCH_FWD_PLAIN(GOPS_ENV_MOBILEROBOT); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
}
