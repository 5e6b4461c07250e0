// Backward sweep of the horizon rollout (replaces `loss.backward()` of fhadp.py:108 /
// infadp.py:151 for the policy path).  Same tiling as the forward kernel: one workgroup owns 16
// trajectories and walks t = H-1 ... 0 carrying the adjoint of the observation (LDS tile G) and of
// the env state (registers).  Per step: env-model adjoint -> adjoint of the action -> wrapper /
// tanh -> head -> hidden-layer deltas on MFMA with the transposed-packed weights.  The deltas are
// written to the stash; the weight gradients are formed afterwards by the dW GEMM kernels.
#include <algorithm>
#include <type_traits>

#include "common.h"
#ifndef GOPS_SWEEP_STAGE_NT
#define GOPS_SWEEP_STAGE_NT true   // the stash rows the sweep stages one step ahead are read once by this launch
#endif
#ifndef GOPS_IDP_BWD_UNROLL
#define GOPS_IDP_BWD_UNROLL 1   // the five sub-step adjoints of pyth_idpendulum unrolled: the parking reads of all of them issue up front (cfg2 sweep 197 -> 181 us)
#endif
#include "env_models.h"
#include "rollout_f16.h"

// LDS floats of the per-tile buffers (everything except the optional staged tiles at the end).
// split: the plane-split sweep keeps the plane images of delta_2 / delta_1 where the fp32 delta tiles would be (the
// tail value net's fp32 tiles, dead once the loop starts, alias them) and needs no LDS copy of the head weights
// ssb: the streamed-split sweep - ONE delta plane image (rewritten in place) where the two fp32 delta tiles would be
__host__ __device__ inline int bwd_lds_floats(int ldx, int ldh, int ref_points, bool f16 = false, bool split = false, bool ssb = false) {
    return TB * ldx + (split ? 2 * split_tile_floats(256) : (ssb ? split_tile_floats(256) : 2 * hidden_tile_floats(ldh, f16))) + TB * 4 + 4 + 4 * TB * 8 +
           (split ? 0 : 4 * ldh) + 4 * TB * ref_points;
}

// delta_y in s_gy[TB][4]  ->  hidden deltas (stashed to stash_d when non-null) and, if want_gx,
// G[m][n] += (delta_1 W_0)[m][n] for n < ncols.
//
// Every delta tile is formed in the MFMA result layout (lane: feature n = lane & 15 of an n-tile, rows
// 4 (lane >> 4) .. +3), which is exactly one 16-byte vector of the feature-major stash: act' comes from one
// vector load of H_j (Z_j for GELU) - global, or the LDS image `stage` of the tile the caller fetched one step
// ahead - and delta_j leaves for the stash as one non-temporal vector store from the registers that formed it.
// The head delta_L = (delta_y W_o) * act'(z_L) is a K = act_dim <= 4 contraction: ONE v_mfma_f32_16x16x4_f32 per
// n-tile (delta_y zero-padded to 4 columns), n-tiles dealt to the waves like in gemm_layer.
template <bool STAGED, class W0T, class W1T, class WP, class Hook>
__device__ __forceinline__ void mlp_backward(const MlpDev& M, const W0T& WT0, const W1T& WT1,
                                             WP Wo, int ldw, const float* s_gy, float* da,
                                             float* db, int ldh, float* G, int ldg, int tid,
                                             float* const* stash_h, float* const* stash_z,
                                             float* const* stash_d, float* stash_dy, size_t row0,
                                             int nvalid, bool want_gx, int ncols, DbgClock& dbg,
                                             Hook&& after_head, const float* stage_head = nullptr,
                                             const float* stage_h1 = nullptr, const float* ext_delta = nullptr,
                                             const f32x4* narrow = nullptr) {
    // narrow: LDS image of the transposed packings wpt[L-1], .., wpt[1], wpt[0] in this order (common.h gemm_layer_lds), or null
    // stage_head / stage_h1: LDS images (direct global->LDS loads issued at the top of the step) of the FM tiles
    // the head / the layer-1 epilogue take act' from; each wave reads only the 4 KiB its own lanes fetched.
    const int lane = tid & 63;
    const int L = M.nl - 1, A = M.dims[M.nl];
    const bool gelu = (M.act == GOPS_ACT_GELU);
    const int m0 = (lane >> 4) << 2;
    // epilogue of the delta tile of layer j: out[m][n] = acc * act'(.), stash_d[j] tile <- the same vector
    auto delta_epi = [&](int j, float* out, const float* stage, bool from_ext) {
        const int N = M.dims[j];
        const GLOBAL_AS float* src = gptr((gelu ? stash_z[j] : stash_h[j]) + row0 * N);
        float* dst = (stash_d != nullptr) ? stash_d[j] + row0 * N : nullptr;
        return [=, &M]<int CNT>(const f32x4 (&acc)[4], int nt0) {
            act_dispatch(M.act, [&]<int ACT>() {
                f32x4 hv[CNT];
                if constexpr (!STAGED) {   // issue every stash load before the math (global latency)
#pragma unroll
                    for (int q = 0; q < CNT; ++q) hv[q] = ld4(src + (((nt0 + q) << 4) + (lane & 15)) * 16 + m0);
                }
#pragma unroll
                for (int q = 0; q < CNT; ++q) {
                    const int n = ((nt0 + q) << 4) + (lane & 15);
                    if constexpr (STAGED) hv[q] = *reinterpret_cast<const f32x4*>(stage + n * 16 + m0);
                    f32x4 dv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a = acc[q][r];
                        if constexpr (!STAGED) {   // gops_mlp_backward: the adjoint of this activation comes from the wide output layer
                            if (from_ext) a = (m0 + r < nvalid) ? gptr(ext_delta)[(row0 + m0 + r) * N + n] : 0.f;
                        }
                        dv[r] = (m0 + r < nvalid) ? a * act_bwd_t<ACT>(hv[q][r], hv[q][r]) : 0.f;
                        out[(m0 + r) * ldh + n] = dv[r];
                    }
                    if (dst != nullptr) __builtin_nontemporal_store(dv, gptr(reinterpret_cast<f32x4*>(dst + n * 16 + m0)));
                }
            });
        };
    };
    // ---- head: delta_L = (delta_y W_o) * act'(z_L) ----
    {
        const int nt_tot = M.dims[L] >> 4, per = (nt_tot + 3) >> 2;
        const int wave = tid >> 6, kk = lane >> 4;
        const bool from_ext = (!STAGED) && (ext_delta != nullptr);
        const float ga = (kk < A && !from_ext) ? s_gy[(lane & 15) * 4 + kk] : 0.f;   // A operand: delta_y[m = lane & 15][k = lane >> 4]
        auto epi = delta_epi(L, da, stage_head, from_ext);
        int nt = wave * per;
        const int nt_end = min(nt_tot, nt + per);
        while (nt < nt_end) {
            const int left = nt_end - nt;
            f32x4 acc[4] = {};
            auto tiles = [&]<int CNT>() {
                float bw[CNT];
#pragma unroll
                for (int q = 0; q < CNT; ++q) bw[q] = (kk < A) ? Wo[kk * ldw + ((nt + q) << 4) + (lane & 15)] : 0.f;   // B: W_o[k][n]
#pragma unroll
                for (int q = 0; q < CNT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, bw[q], acc[q], 0, 0, 0);
                epi.template operator()<CNT>(acc, nt);
                nt += CNT;
            };
            if (left >= 4) tiles.template operator()<4>();
            else if (left >= 2) tiles.template operator()<2>();
            else tiles.template operator()<1>();
        }
        if (stash_dy != nullptr && tid < TB) {
            f32x4 v = {s_gy[tid * 4 + 0], s_gy[tid * 4 + 1], s_gy[tid * 4 + 2], s_gy[tid * 4 + 3]};
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                if (a >= A || tid >= nvalid) v[a] = 0.f;
            *gptr(reinterpret_cast<f32x4*>(stash_dy + (row0 + tid) * 4)) = v;
        }
    }
    DBG_TICK(3)
    __syncthreads();
    DBG_TICK(4)
    after_head();   // long stretch without dependent global loads ahead: the caller's prefetches go here
    DBG_TICK(5)
    float* cur = da;
    float* out = db;
    // ---- hidden layers j = L-1 .. 1: delta_j = (delta_{j+1} W_j) * act'(z_j) ----
    for (int j = L - 1; j >= 1; --j) {
        const int N = M.dims[j], kch = M.dims[j + 1] >> 4, nt_tot = N >> 4;
        auto epi = delta_epi(j, out, stage_h1, false);   // (staged variants: L == 2, so j == 1 only)
        bool done = false;
        if constexpr (!std::is_same<W1T, NoW>::value) {
            if (j == 1) { gemm_layer_stat(cur, ldh, WT1, nt_tot, tid, epi); done = true; }
        }
        if (!done) {
            if (narrow != nullptr) {
                gemm_layer_lds(cur, ldh, kch, nt_tot, narrow, tid, epi);
                narrow += kch * nt_tot * 64;
            } else {
                gemm_layer(cur, ldh, kch, nt_tot, M.wpt[j], tid, epi);
            }
        }
        DBG_TICK(6)
        __syncthreads();
        DBG_TICK(7)
        DBG_TICK(8)
        float* tmp = cur; cur = out; out = tmp;
    }
    // ---- input adjoint g_x = delta_1 W_0 ----
    if (want_gx) {
        const int kch = M.dims[1] >> 4, nt_tot = M.kp[0] >> 4;
        auto epi = [&]<int CNT>(const f32x4 (&acc)[4], int nt0) {
            // every read of the tile first, then the adds, then the writes: as `G[..] += acc` hipcc serialises the 4 CNT
            // read-modify-writes (each a full LDS round trip) because it cannot prove the addresses distinct
            float gold[CNT][4];
#pragma unroll
            for (int q = 0; q < CNT; ++q) {
                const int n = min(((nt0 + q) << 4) + (lane & 15), ncols - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) gold[q][r] = G[(((lane >> 4) << 2) + r) * ldg + n];
            }
#pragma unroll
            for (int q = 0; q < CNT; ++q) {
                const int n = ((nt0 + q) << 4) + (lane & 15);
                if (n < ncols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) G[(((lane >> 4) << 2) + r) * ldg + n] = gold[q][r] + acc[q][r];
                }
            }
        };
        if constexpr (!std::is_same<W0T, NoW>::value) gemm_layer_stat(cur, ldh, WT0, nt_tot, tid, epi, kch, M.wpt[0]);
        else if (narrow != nullptr) gemm_layer_lds(cur, ldh, kch, nt_tot, narrow, tid, epi);
        else gemm_layer(cur, ldh, kch, nt_tot, M.wpt[0], tid, epi);
        DBG_TICK(9)
    }
}

// obs -> 64 -> 64 -> act policies on the narrow launches (rollout_fwd.hip: mlp_hidden_forward_n64): mlp_backward written out for
// that shape - one n-tile per wave in every phase, no layer loop, nothing of the parameter block read inside the step loop (Hot64B,
// pinned once per launch), and BOTH act' vectors of the wave's tile requested at the top of the step's backward (this instantiation
// has the registers for it: the streamed GEMM's fragment ring is not part of it).  Same products in the same order as mlp_backward.
struct Hot64B {
    int act, A, nt0;               // activation kind, action dimension, 16-wide n-tiles of the (padded) policy input
    const float *a1, *a2;          // act' sources of the two hidden layers: H_j, or Z_j (= gelu'(z)) for GELU
    float *d1, *d2, *dy;           // delta stashes (null: not written)
    const f32x4 *w1t, *w0t;        // LDS images of the transposed packings wpt[1], wpt[0] (narrow_fill)
};
template <class WP, class Hook>
__device__ __forceinline__ void mlp_backward_n64(const Hot64B& hn, WP Wo, int ldw, const float* s_gy, float* da, float* db, int ldh,
                                                 float* G, int ldg, int tid, size_t row0, int nvalid, bool want_gx, int ncols,
                                                 DbgClock& dbg, Hook&& after_head) {
    const int lane = tid & 63, wave = tid >> 6;
    const int n = (wave << 4) + (lane & 15), m0 = (lane >> 4) << 2, kk = lane >> 4;
    const f32x4 hv2 = ld4(gptr(hn.a2 + row0 * 64) + n * 16 + m0);
    const f32x4 hv1 = ld4(gptr(hn.a1 + row0 * 64) + n * 16 + m0);
    auto delta_tile = [&](const f32x4& acc, const f32x4& hv, float* out, float* dst) {
        act_dispatch(hn.act, [&]<int ACT>() {
            f32x4 dv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dv[r] = (m0 + r < nvalid) ? acc[r] * act_bwd_t<ACT>(hv[r], hv[r]) : 0.f;
                out[(m0 + r) * ldh + n] = dv[r];
            }
            if (dst != nullptr) __builtin_nontemporal_store(dv, gptr(reinterpret_cast<f32x4*>(dst + row0 * 64 + n * 16 + m0)));
        });
    };
    {   // ---- head: delta_2 = (delta_y W_o) * act'(z_2): ONE v_mfma_f32_16x16x4_f32 ----
        const float ga = (kk < hn.A) ? s_gy[(lane & 15) * 4 + kk] : 0.f;   // A operand: delta_y[m = lane & 15][k = lane >> 4]
        const float bw = (kk < hn.A) ? Wo[kk * ldw + n] : 0.f;             // B: W_o[k][n]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, bw, acc, 0, 0, 0);
        delta_tile(acc, hv2, da, hn.d2);
        if (hn.dy != nullptr && tid < TB) {
            f32x4 v = {s_gy[tid * 4 + 0], s_gy[tid * 4 + 1], s_gy[tid * 4 + 2], s_gy[tid * 4 + 3]};
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                if (a >= hn.A || tid >= nvalid) v[a] = 0.f;
            *gptr(reinterpret_cast<f32x4*>(hn.dy + (row0 + tid) * 4)) = v;
        }
    }
    DBG_TICK(3)
    __syncthreads();
    DBG_TICK(4)
    after_head();
    DBG_TICK(5)
    auto gemm64 = [&](const float* Atile, const f32x4* Wl) -> f32x4 {   // [16 x 64] x this wave's [64 x 16] tile: 8 reads, 16 MFMAs
        const float* arow = Atile + (lane & 15) * ldh + 4 * (lane >> 4);
        const f32x4* wl = Wl + (wave * 4 * 64 + lane);
        f32x4 a[4], b[4], acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[c] = *reinterpret_cast<const f32x4*>(arow + 16 * c); b[c] = wl[c * 64]; }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][i], b[c][i], acc, 0, 0, 0);
        return acc;
    };
    {   // ---- delta_1 = (delta_2 W_1) * act'(z_1) ----
        const f32x4 acc = gemm64(da, hn.w1t);
        delta_tile(acc, hv1, db, hn.d1);
    }
    DBG_TICK(6)
    __syncthreads();
    DBG_TICK(7)
    DBG_TICK(8)
    if (want_gx && wave < hn.nt0) {   // ---- input adjoint g_x = delta_1 W_0 ----
        const f32x4 acc = gemm64(db, hn.w0t);
        const int nr = min(n, ncols - 1);
        float gold[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gold[r] = G[(m0 + r) * ldg + nr];
        if (n < ncols) {
#pragma unroll
            for (int r = 0; r < 4; ++r) G[(m0 + r) * ldg + n] = gold[r] + acc[r];
        }
    }
    DBG_TICK(9)
}

// Plane-split backward through the policy MLP (common.h SplitDev): delta_y (s_gy) -> delta_2 -> delta_1 -> g_x.
// The head delta is one exact fp32 MFMA per n-tile as in mlp_backward; every delta tile leaves its epilogue registers
// three ways: one 16-byte vector to the fp32 FM stash (weight-gradient GEMM), and - through plane_store - as the bf16 / f16
// plane image the next contraction reads.  The half plane carries the step's power-of-two scale s_scale[0] (from
// max|delta_y| of the tile), taken out again with the residual accumulator's factor.
// AMAX: compile-time bound of the action dimension.  The output layer's weight gradient dW_o = sum delta_y^T H_2 is formed
// here as well (activations other than GELU, where the act' operand IS H_2): every lane keeps dW_o[a][its four columns]
// over its four rows in AMAX x 4 registers across all steps (and tiles) of the workgroup; one partial per workgroup goes to
// the split-K reduction - the separate dw_out pass over H_2 (126 MB at the target) is not launched.
template <int PT0, int AMAX>
struct SplitSweep {
    float dwo[AMAX][4];          // dW_o[a][64 wave + 16 q + (lane & 15)], partial over rows 4 (lane >> 4) .. +3
    float dbo[AMAX];             // d b_o[a], same partial
    static constexpr int PIN1 = (AMAX == 2) ? 3 : GOPS_PIN_MODE;   // (rollout_fwd.hip SplitPolicy: both planes pinned for veh3dofconti: sweep 187.0 -> 183.8 us)
    StatQ<8, 4, false, PIN1> QT1;      // delta_2 -> delta_1 through W_1: both planes in registers
    // delta_1 -> g_x through W_0: bf16 plane in registers, half residual plane in LDS; more than 128 inputs (PT0 > 2 n-tiles
    // per wave): both planes stream from L2 (StreamQ, common.h)
    static constexpr bool STREAMT0 = PT0 > 2;
    static constexpr int PTS = (PT0 + 1) & ~1;   // streamed: n-tiles per wave rounded up to pairs
    typename std::conditional<STREAMT0, StreamQ<8, PTS>, StatQ<8, PT0, true>>::type QT0;
    StreamRing<8> ring0;         // (streamed W_0 only)
    float wo[4];                 // W_o[k = lane >> 4][64 wave + 16 q + (lane & 15)]: B operand of the head-delta MFMA
    f32x4 hv[4];                 // act' operands of this lane's four columns: H_2 (Z_2 for GELU) of the step, then H_1
    __device__ __forceinline__ void load(const RolloutParams& p, int tid, f16x8* rt0_lds) {
        const int lane = tid & 63, wave = tid >> 6;
        const MlpDev& M = p.pol;
        QT1.load(p.sp.w1t[1], p.sp.rt[1], p.sp.invt[1], M.dims[1] >> 4, tid);
        if constexpr (STREAMT0) QT0.load(p.sp.w1t[0], p.sp.rt[0], p.sp.invt[0], M.kp[0] >> 4, tid);
        else QT0.load(p.sp.w1t[0], p.sp.rt[0], p.sp.invt[0], M.kp[0] >> 4, tid, rt0_lds);
        const int K = M.dims[2], A = M.dims[3], kk = lane >> 4;
        hot.act = keep_s(M.act);
        hot.A = keep_s(A);
        const bool gelu_l = M.act == GOPS_ACT_GELU;
        hot.a[0] = nullptr; hot.d[0] = nullptr;
        {   // (pinned first, selected after: a select on the loaded pointers themselves may be formed on the vector unit)
            const float* const h1 = keep_s<true, const float*>(p.st.h[1]), * const h2 = keep_s<true, const float*>(p.st.h[2]);
            const float* const z1 = keep_s<true, const float*>(p.st.z[1]), * const z2 = keep_s<true, const float*>(p.st.z[2]);
            hot.a[1] = gelu_l ? z1 : h1;
            hot.a[2] = gelu_l ? z2 : h2;
        }
        hot.d[1] = keep_s(p.st.d[1]); hot.d[2] = keep_s(p.st.d[2]);
        hot.dy = keep_s(p.st.dy);
        hot.h2 = keep_s<true, const float*>(p.st.h[2]);
#pragma unroll
        for (int q = 0; q < 4; ++q) wo[q] = (kk < A) ? gptr(M.w[2])[kk * K + 64 * wave + 16 * q + (lane & 15)] : 0.f;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            dbo[a] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) dwo[a][q] = 0.f;
        }
    }
    // this workgroup's partial of dW_o / d b_o -> part[blockIdx.x][a][K], part_b[blockIdx.x][a]   (rows a >= A of a slab are not read)
    __device__ __forceinline__ void store_out_grad(const RolloutParams& p, float* part, float* part_b, int tid) {
        const int lane = tid & 63, wave = tid >> 6, K = p.pol.dims[2], A = p.pol.dims[3];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            float tb = dbo[a];
            tb += __shfl_xor(tb, 16);
            tb += __shfl_xor(tb, 32);
            if (a < A && tid == 0) gptr(part_b)[(size_t)blockIdx.x * A + a] = tb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = dwo[a][q];
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                if (a < A && lane < 16) gptr(part)[((size_t)blockIdx.x * A + a) * K + 64 * wave + 16 * q + lane] = t;
            }
        }
    }
    // The act' operands come straight from the FM stash into registers (one 16-byte vector per column), issued a phase
    // ahead of their use: H_2 at the top of the step (used after the env adjoint), H_1 right after the head (used after the
    // delta_1 contraction).  (The fp32-MFMA variants stage these tiles in LDS; here that LDS holds W_0's residual plane.)
    // what fetch() / run() read of the parameter block, pinned to scalar registers by the kernel (common.h keep_s)
    struct Hot {
        int act, A;
        const float* a[3];   // act' operand tensors of hidden layers 1, 2: H_j, or gelu'(z_j) for GELU
        float* d[3];         // delta stashes of hidden layers 1, 2
        float* dy;
        const float* h2;     // H_2 (fused output-layer gradient with GELU)
    };
    Hot hot;
    unsigned ovf = 0;   // half-range overflow of a delta plane conversion (common.h split2h), tested at the end of a tile
    __device__ __forceinline__ void fetch(const RolloutParams&, int j, size_t row0, int tid) {
        const int lane = tid & 63, wave = tid >> 6, m0 = (lane >> 4) << 2;
        const GLOBAL_AS float* src = gptr(hot.a[j] + row0 * 256);
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = ld4(src + (64 * wave + 16 * q + (lane & 15)) * 16 + m0);
    }
    template <class Hook>
    __device__ __forceinline__ void run(const RolloutParams& p, const float* s_gy, const float* s_scale, char* dq2, char* dq1,
                                        float* G, int ldg, int tid, size_t row0, int nvalid, bool want_gx, int ncols,
                                        DbgClock& dbg, Hook&& after_head, bool fuse_out) {
        const int lane = tid & 63, wave = tid >> 6, m0 = (lane >> 4) << 2;
        constexpr int ROWB = 2 * 256 + 16;
        const int A = hot.A;
        const float s = s_scale[0], inv_s = s_scale[1];
        // delta tile of layer j from the contraction result `a`: * act'(.), zero for padding rows, -> stash + plane image
        auto finish = [&](int j, f32x4 (&a)[4], char* planes) {
            float* dst = hot.d[j] + row0 * 256;
            act_dispatch(hot.act, [&]<int ACT>() {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = 64 * wave + 16 * q + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[q][r] = (m0 + r < nvalid) ? a[q][r] * act_bwd_t<ACT>(hv[q][r], hv[q][r]) : 0.f;
                    __builtin_nontemporal_store(a[q], gptr(reinterpret_cast<f32x4*>(dst + n * 16 + m0)));
                }
            });
            plane_store(planes, ROWB, wave, lane, a, s, ovf);
        };
        {   // ---- head: delta_2 = (delta_y W_o) * act'(z_2) ----
            const int kk = lane >> 4;
            const float ga = (kk < A) ? s_gy[(lane & 15) * 4 + kk] : 0.f;   // A operand: delta_y[m = lane & 15][k = lane >> 4]
            // GELU with the fused output-layer gradient: hv holds gelu'(z_2), H_2 itself is fetched here and lands behind the
            // head delta's epilogue (a short live range: as a member held across the env adjoint it spilled)
            const bool gelu_fuse = fuse_out && hot.act == GOPS_ACT_GELU;
            f32x4 h2v[4] = {};
            if (gelu_fuse) {
                const GLOBAL_AS float* s2 = gptr(hot.h2 + row0 * 256);
#pragma unroll
                for (int q = 0; q < 4; ++q) h2v[q] = ld4(s2 + (64 * wave + 16 * q + (lane & 15)) * 16 + m0);
            }
            f32x4 acc[4] = {};
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, wo[q], acc[q], 0, 0, 0);
            finish(2, acc, dq2);
            if (fuse_out) {   // dW_o += delta_y^T H_2 over this lane's rows and columns (hv still holds H_2; GELU: h2v does)
                const bool gelu = gelu_fuse;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 dyr = *reinterpret_cast<const f32x4*>(s_gy + (m0 + r) * 4);
                    const bool ok = m0 + r < nvalid;
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) {
                        const float d = ok ? dyr[a] : 0.f;
                        dbo[a] += d;
#pragma unroll
                        for (int q = 0; q < 4; ++q) dwo[a][q] = fmaf(d, gelu ? h2v[q][r] : hv[q][r], dwo[a][q]);
                    }
                }
            }
            fetch(p, 1, row0, tid);   // H_1 of this step: lands behind the delta_1 contraction
            if (tid < TB) {
                f32x4 v = {s_gy[tid * 4 + 0], s_gy[tid * 4 + 1], s_gy[tid * 4 + 2], s_gy[tid * 4 + 3]};
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a)
                    if (a >= A || tid >= nvalid) v[a] = 0.f;
                *gptr(reinterpret_cast<f32x4*>(hot.dy + (row0 + tid) * 4)) = v;
            }
        }
        DBG_TICK(3)
        __syncthreads();
        DBG_TICK(4)
        after_head();
        DBG_TICK(5)
        {   // ---- delta_1 = (delta_2 W_1) * act'(z_1) ----
            f32x4 acc[4] = {}, accr[4] = {};
            gemm_split(dq2, ROWB, QT1, lane, acc, accr);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] = split_combine(acc[q][r], accr[q][r], QT1.inv[q], inv_s);
            }
            if constexpr (STREAMT0) {
                if (want_gx) QT0.prime(ring0, 0);   // W_0's first chunks travel during delta_1's epilogue
            }
            finish(1, acc, dq1);
        }
        DBG_TICK(6)
        __syncthreads();
        DBG_TICK(7)
        DBG_TICK(8)
        if constexpr (STREAMT0) {
            if (want_gx) {   // ---- g_x = delta_1 W_0 with W_0 streamed: pairs of n-tiles PTS wave + 2 k, + 1 ----
#pragma unroll
                for (int k = 0; k < PTS / 2; ++k) {
                    f32x4 acc[2] = {}, accr[2] = {};
                    gemm_split_pair(dq1, ROWB, QT0, ring0, k, lane, acc, accr);
                    if (k + 1 < PTS / 2) QT0.prime(ring0, k + 1);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = 16 * (PTS * wave + 2 * k + j) + (lane & 15);
                        if (n < ncols) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) G[(m0 + r) * ldg + n] += split_combine(acc[j][r], accr[j][r], QT0.inv[2 * k + j], inv_s);
                        }
                    }
                }
                DBG_TICK(9)
            }
        } else
        if (want_gx) {   // ---- input adjoint g_x = delta_1 W_0, accumulated into G ----
            f32x4 acc[PT0] = {}, accr[PT0] = {};
            gemm_split(dq1, ROWB, QT0, lane, acc, accr);
            float gold[PT0][4];
#pragma unroll
            for (int j = 0; j < PT0; ++j) {
                const int n = min(16 * (PT0 * wave + j) + (lane & 15), ncols - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) gold[j][r] = G[(m0 + r) * ldg + n];
            }
#pragma unroll
            for (int j = 0; j < PT0; ++j) {
                const int n = 16 * (PT0 * wave + j) + (lane & 15);
                if (n < ncols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) G[(m0 + r) * ldg + n] = gold[j][r] + split_combine(acc[j][r], accr[j][r], QT0.inv[j], inv_s);
                }
            }
            DBG_TICK(9)
        }
    }
};
struct NoSweep {};

// Env kinds whose streamed-split sweep has the registers for it (the vehicle / idpendulum instantiations would spill): those
// instantiations also walk their tiles grid-stride (one partial slab per workgroup).
__host__ __device__ constexpr bool ssb_fuse_kind(int env) {
    return env == GOPS_ENV_LQ || env == GOPS_ENV_CARTPOLE || env == GOPS_ENV_PENDULUM || env == GOPS_ENV_MOBILEROBOT;
}
bool ssb_fuses_out(const RolloutParams& p) { return p.ssb && ssb_fuse_kind(p.env.kind); }
// Output-layer weight gradient accumulated inside the streamed-split sweep (as SplitSweep does): lane (f = lane & 15,
// g = lane >> 4) of wave w holds dW_o[a][64 w + 16 q + f] over the rows 4g .. 4g+3 of every tile and step the workgroup walked.
struct SsOutGrad {
    float dwo[GOPS_MAX_ACT][4];
    float dbo[GOPS_MAX_ACT];
    unsigned ovf;   // half-range overflow of a delta plane conversion (common.h split2h), tested at the end of a tile
};
// this workgroup's partial -> part[blockIdx.x][a][K], part_b[blockIdx.x][a]
__device__ __forceinline__ void ss_store_out_grad(const SsOutGrad& og, int K, int A, float* part, float* part_b, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int a = 0; a < GOPS_MAX_ACT; ++a) {
        float tb = og.dbo[a];
        tb += __shfl_xor(tb, 16);
        tb += __shfl_xor(tb, 32);
        if (a < A && tid == 0) gptr(part_b)[(size_t)blockIdx.x * A + a] = tb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t = og.dwo[a][q];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (a < A && lane < 16) gptr(part)[((size_t)blockIdx.x * A + a) * K + 64 * wave + 16 * q + lane] = t;
        }
    }
}

// Streamed-split sweep through one net (any number of 256-wide hidden layers; the counterpart of ss_net_forward): delta_y
// (s_gy[TB][4]) -> hidden deltas (FM stash st_d when non-null) and, if want_gx, G += delta_1 W_0.  Weight planes (transposed
// packing, SplitNetDev) stream from L2; ONE delta plane image `dq`, rewritten in place behind a barrier; act' operands come
// from the FM stash straight into registers, requested one phase ahead (vmcnt retires in order: requested in front of a
// contraction they would sit between it and its first weight fragment).  Wo: head weights [A][ldw]
// (LDS copy or global).
template <class WP, class Hook>
__device__ __forceinline__ void ss_net_backward(const MlpDev& M, const SplitNetDev& ST, WP Wo, int ldw, const float* s_gy, const float* s_scale,
                                                char* dq, float* G, int ldg, int tid, float* const* st_h, float* const* st_z,
                                                float* const* st_d, float* st_dy, size_t row0, int nvalid, bool want_gx, int ncols,
                                                Hook&& after_head, SsOutGrad& og, bool fuse_out) {
    const int lane = tid & 63, wave = tid >> 6, m0 = (lane >> 4) << 2;
    constexpr int ROWB = 2 * 256 + 16;
    const int L = M.nl - 1, A = M.dims[M.nl];
    const bool gelu = M.act == GOPS_ACT_GELU;
    const float s = s_scale[0], inv_s = s_scale[1];
    auto fetch = [&](int j, f32x4 (&hv)[4]) {
        const GLOBAL_AS float* src = gptr((gelu ? st_z[j] : st_h[j]) + row0 * 256);
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = ld4(src + (64 * wave + 16 * q + (lane & 15)) * 16 + m0);
    };
    auto finish = [&](int j, f32x4 (&a)[4], const f32x4 (&hv)[4]) {   // * act'(.), zero for padding rows, -> stash + plane image
        float* dst = (st_d != nullptr) ? st_d[j] + row0 * 256 : nullptr;
        act_dispatch(M.act, [&]<int ACT>() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 64 * wave + 16 * q + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) a[q][r] = (m0 + r < nvalid) ? a[q][r] * act_bwd_t<ACT>(hv[q][r], hv[q][r]) : 0.f;
                if (dst != nullptr) __builtin_nontemporal_store(a[q], gptr(reinterpret_cast<f32x4*>(dst + n * 16 + m0)));
            }
        });
        plane_store(dq, ROWB, wave, lane, a, s, og.ovf);
    };
    f32x4 hvn[4];   // act' operands requested one phase ahead
    {   // ---- head: delta_L = (delta_y W_o) * act'(z_L): one K = 4 fp32 MFMA per n-tile ----
        f32x4 hv[4];
        fetch(L, hv);
        if (L >= 2) fetch(L - 1, hvn);   // the next phase's act' operands: in flight during this phase's math
        const int kk = lane >> 4;
        const float ga = (kk < A) ? s_gy[(lane & 15) * 4 + kk] : 0.f;
        f32x4 acc[4] = {};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float wo = (kk < A) ? Wo[kk * ldw + 64 * wave + 16 * q + (lane & 15)] : 0.f;
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, wo, acc[q], 0, 0, 0);
        }
        // fused output-layer gradient, GELU: hv holds gelu'(z_L); H_L itself is fetched here (a short live range)
        f32x4 hlv[4] = {};
        if (fuse_out && gelu) {
            const GLOBAL_AS float* sl = gptr(st_h[L] + row0 * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) hlv[q] = ld4(sl + (64 * wave + 16 * q + (lane & 15)) * 16 + m0);
        }
        finish(L, acc, hv);
        if (fuse_out) {   // dW_o += delta_y^T H_L over this lane's rows and columns
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 dyr = *reinterpret_cast<const f32x4*>(s_gy + (m0 + r) * 4);
                const bool ok = m0 + r < nvalid;
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a) {
                    const float d = ok ? dyr[a] : 0.f;
                    og.dbo[a] += d;
#pragma unroll
                    for (int q = 0; q < 4; ++q) og.dwo[a][q] = fmaf(d, gelu ? hlv[q][r] : hv[q][r], og.dwo[a][q]);
                }
            }
        }
        if (st_dy != nullptr && tid < TB) {
            f32x4 v = {s_gy[tid * 4 + 0], s_gy[tid * 4 + 1], s_gy[tid * 4 + 2], s_gy[tid * 4 + 3]};
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                if (a >= A || tid >= nvalid) v[a] = 0.f;
            *gptr(reinterpret_cast<f32x4*>(st_dy + (row0 + tid) * 4)) = v;
        }
    }
    __syncthreads();
    after_head();
    for (int j = L - 1; j >= 1; --j) {   // ---- delta_j = (delta_{j+1} W_j) * act'(z_j) ----
        f32x4 hv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = hvn[q];
        f32x4 acc[4] = {}, accr[4] = {};
        float inv[4];
        ss_layer_gemm<8>(dq, ROWB, ST.w1[j], ST.r[j], ST.inv[j], 16, tid, acc, accr, inv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] = split_combine(acc[q][r], accr[q][r], inv[q], inv_s);
        }
        if (j >= 2) fetch(j - 1, hvn);   // (before the barrier and this layer's epilogue)
        __syncthreads();   // every wave has read the delta image it is about to overwrite
        finish(j, acc, hv);
        __syncthreads();
    }
    if (want_gx) {   // ---- g_x = delta_1 W_0 into G: n-tiles over the (16-padded) inputs ----
        f32x4 acc[4] = {}, accr[4] = {};
        float inv[4];
        ss_layer_gemm<8>(dq, ROWB, ST.w1[0], ST.r[0], ST.inv[0], M.kp[0] >> 4, tid, acc, accr, inv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 16 * (4 * wave + q) + (lane & 15);
            if (n < ncols) {
#pragma unroll
                for (int r = 0; r < 4; ++r) G[(m0 + r) * ldg + n] += split_combine(acc[q][r], accr[q][r], inv[q], inv_s);
            }
        }
    }
}

// SK0 / SK1 as in the forward kernel: here the stationary fragments are the TRANSPOSED packings
// (delta_2 -> delta_1 through W_1: 16 chunks x 4 tiles).  SK0 here = number of the 16 K-chunks of
// delta_1 -> g_x (through W_0^T, PT0 n-tiles per wave) that stay in registers; the rest streams.
// F16: GOPS_DTYPE_F16 - deltas and act' operands are half, the contractions run on
// v_mfma_f32_16x16x32_f16 (rollout_f16.h), and every adjoint carries the launch's power-of-two scale
// (gscale[0]) that the reduce kernel takes out of the parameter gradients again.
// (F16 kernels: 4 workgroups per CU - launch bound 4 waves / SIMD, <= 128 registers.)
// EXT (streamed fp32 kernels of the obs == state env kinds only): terminal observation adjoint in, initial
// observation adjoint out, parameter deltas of step 0 only - gops_rollout_backward_adj / gops_mlp_backward_x
// SPLIT: plane-split contractions (SplitSweep; SK1 > 0, PT0 = n-tiles of g_x per wave)
// MULTI (SPLIT only): more tiles than workgroups - grid-stride walk over the tiles
// SSB: streamed-split sweep (ss_net_backward): plane-split MFMAs with all (transposed) weight planes streamed from L2, two
// workgroups per CU; the tail value net's input adjoint on the same routine
// N64 (narrow launches of obs -> 64 -> 64 -> act policies: RolloutParams.narrow == 2): the policy's backward by mlp_backward_n64
template <int ENV, int SK0, int SK1, bool TAIL, int PT0 = 1, bool F16 = false, bool EXT = false, bool SPLIT = false, bool MULTI = false, bool SSB = false, bool N64 = false>
__global__ __launch_bounds__(NTHREADS, F16 ? 4 : (SSB ? 2 : ((SK0 == 0 && SK1 == 0) ? 3 : 1))) void rollout_bwd_kernel(const RolloutParams* __restrict__ pp, const BwdPatch q) {
    extern __shared__ __attribute__((aligned(16))) float smem_raw[];
    const RolloutParams& p = *pp;
    const int tid = threadIdx.x;
    // pyth_lq on the streamed-split sweep: the env description in front of everything else in LDS (common.h: env_in_lds)
    constexpr bool ENVLDS = env_in_lds(ENV, SSB);
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
    int tile = blockIdx.x;   // SPLIT: grid-stride walk over the tiles with the weights resident (else one tile per workgroup)
    int b0 = tile * TB;
    int nvalid = min(TB, p.B - b0);
    const int O = p.env.obs_dim, A = p.env.act_dim;
    constexpr bool SURR = (ENV == GOPS_ENV_VEH3DOF_SURR);   // veh3dofconti + surrounding vehicles + constraint outputs
    constexpr bool VEH = (ENV == GOPS_ENV_VEH3DOFCONTI) || SURR;
    constexpr bool VEH2 = (ENV == GOPS_ENV_VEH2DOF);
    constexpr bool REF = VEH || VEH2;
    constexpr bool CSTR = SURR || VEH2 || (ENV == GOPS_ENV_MOBILEROBOT);   // models with constraint outputs
    const int ldx = p.ldx, ldh = (SK1 > 0) ? 260 : p.ldh;
    float* G = smem;                    // [TB][ldx] adjoint of obs_{t+1}
    float* da = G + TB * ldx;           // [TB][ldh]
    float* db = da + hidden_tile_floats(ldh, F16);    // [TB][ldh] floats, or [TB][ldh + 4] halfs (F16)
    // SPLIT: the plane images of delta_2 / delta_1 take the place of the fp32 delta tiles (which only the tail value net uses)
    char* dq2 = reinterpret_cast<char*>(da);
    char* dq1 = dq2 + 4 * split_tile_floats(256);
    float* s_gy = SPLIT ? da + 2 * split_tile_floats(256) : (SSB ? da + split_tile_floats(256) : db + hidden_tile_floats(ldh, F16));  // [TB][4]
    float* s_scale = s_gy + TB * 4;     // [4] SPLIT: this step's power-of-two delta scale and its inverse
    float* red = s_scale + 4;           // [4][TB][8]
    float* s_wo = red + 4 * TB * 8;     // [4][ldh] head weights (not in the SPLIT layout: SplitSweep keeps its columns in registers)
    f32x4* s_ref = reinterpret_cast<f32x4*>(s_wo + (SPLIT ? 0 : 4 * ldh));   // veh: [TB][TL]
    float* s_idp = reinterpret_cast<float*>(s_ref);             // idpendulum: [TB][5][24] sub-step parking
    // One-workgroup-per-CU variants: LDS copies of this step's H_2 / H_1 (Z for GELU) tiles, [2][TB][256]
    constexpr bool STAGE = (SK1 > 0);   // (those variants are only selected for obs-256-256-act policies)
    float* s_stage = smem + bwd_lds_floats(ldx, ldh, REF ? p.env.pre_horizon + 1 + p.H
                                                                             : (ENV == GOPS_ENV_IDPENDULUM ? IDP_POINTS(SPLIT) : 0), F16, SPLIT, SSB);

    const int ld16 = (p.ldh - 4) + 8;                 // F16: leading dimension (halfs) of the delta tiles
    // fp32 observation column i of row m of the stash tile at row0 (the env adjoints read the first few):
    // F16 keeps a row-major [S][8] fp32 copy, the fp32 stash is feature-major
    auto x_col = [&](size_t row0, int m, int i) -> float {
        if constexpr (F16) return gptr(p.st.xf)[(row0 + m) * 8 + i];
        else return gptr(p.st.x)[(row0 * p.pol.kp[0]) + i * 16 + m];
    };
    const IdpConst IC = idp_const();
    const VehConst VC = veh_const();
    const int TL = p.env.pre_horizon + 1 + p.H;
    const int kp0 = p.pol.kp[0];
    {
        const int Lh = p.pol.nl - 1, K = p.pol.dims[Lh], Ao = p.pol.dims[p.pol.nl];
        if constexpr (!SPLIT) {
            for (int idx = tid; idx < Ao * K; idx += NTHREADS) {
                const int a = idx / K, k = idx - a * K;
                s_wo[a * ldh + k] = gptr(p.pol.w[Lh])[idx];
            }
        }
    }
    // narrow nets on the plain streamed fp32 sweeps: the transposed packings of the policy's hidden layers resident in LDS, in the
    // order the sweep walks them (common.h gemm_layer_lds)
    constexpr bool NARROWABLE = (SK0 == 0) && (SK1 == 0) && !F16 && !SPLIT && !SSB;
    const f32x4* s_narrow = nullptr;
    if constexpr (NARROWABLE) {
        if (p.narrow) {
            f32x4* dst = reinterpret_cast<f32x4*>(smem_raw + p.narrow_off_bwd);
            s_narrow = dst;
            for (int j = p.pol.nl - 2; j >= 0; --j) {
                const int n4 = (p.pol.kp[j] * p.pol.dims[j + 1]) >> 2;
                narrow_fill(dst, p.pol.wpt[j], n4, tid);
                dst += n4;
            }
        }
    }
    Hot64B hot64 = {};
    if constexpr (N64) {
        static_assert(NARROWABLE && !EXT, "N64 is a variant of the plain streamed fp32 sweep");
        const bool gl = p.pol.act == GOPS_ACT_GELU;
        hot64.act = keep_s(p.pol.act);
        hot64.A = keep_s(p.pol.dims[3]);
        hot64.nt0 = keep_s(p.pol.kp[0] >> 4);
        hot64.a1 = keep_s(gl ? p.st.z[1] : p.st.h[1]);
        hot64.a2 = keep_s(gl ? p.st.z[2] : p.st.h[2]);
        hot64.d1 = keep_s(p.st.d[1]); hot64.d2 = keep_s(p.st.d[2]); hot64.dy = keep_s(p.st.dy);
        hot64.w1t = s_narrow;
        hot64.w0t = s_narrow + 4 * 4 * 64;
    }
    typename std::conditional<(SK0 > 0 && !SPLIT), StatW<(SK0 > 0 ? SK0 : 1), PT0>, NoW>::type WT0;
    typename std::conditional<(SK1 > 0 && !SPLIT), StatW<16, 4>, NoW>::type WT1;
    if constexpr (SK0 > 0 && !SPLIT) WT0.load(p.pol.wpt[0], kp0 >> 4, tid, p.pol.dims[1] >> 4);
    if constexpr (SK1 > 0 && !SPLIT) WT1.load(p.pol.wpt[1], p.pol.dims[1] >> 4, tid);
    constexpr int AMAX = (ENV == GOPS_ENV_IDPENDULUM) ? 1 : ((ENV == GOPS_ENV_VEH3DOFCONTI) ? 2 : GOPS_MAX_ACT);
    typename std::conditional<SPLIT, SplitSweep<PT0, AMAX>, NoSweep>::type SS;
    // SPLIT: W_0's residual plane ((kp0 / 16) n-tiles x 8 chunks x 1 KiB) behind the two small staging halves
    if constexpr (SPLIT) SS.load(p, tid, reinterpret_cast<f16x8*>(s_stage + 2 * (TB * ENV_STASH + TB * 8)));

    DbgClock dbg;
    dbg.init((q.dbg != nullptr) && blockIdx.x == 0 && tid == 0);
    // One-workgroup-per-CU variants: everything step tt reads from the stash - the H_2 / H_1 tiles the
    // head and the layer-1 epilogue take act' from (Z for GELU), the env rows and the first 8 columns of
    // the observation rows - is fetched straight into LDS (global_load_lds: no VGPRs) ONE STEP AHEAD, into
    // the half of the staging area selected by the step's parity.  For the tiles each wave fetches
    // exactly what its own lanes read later: rows 4w..4w+3 of H_2 (one 1-KiB row per instruction) and
    // columns 64w..64w+63 of H_1 (4 rows x 64 columns per instruction); the small rows are fetched by
    // waves 0 / 1 and read by everyone after the end-of-step barrier (which drains the loads).
    constexpr int STAGE_TILES = SPLIT ? 0 : 2 * TB * 256;   // (SPLIT: act' operands go stash -> registers, SplitSweep::fetch)
    constexpr int STAGE_FLOATS = STAGE_TILES + TB * ENV_STASH + TB * 8;
    // SPLIT: what the step loop reads of the parameter block, pinned to scalar registers (common.h keep_s) - left to hipcc each
    // of these is an s_load + s_waitcnt lgkmcnt(0) per step (22 of them in the round-4 sweep)
    const int hH = keep_s<SPLIT>(p.H), hP = keep_s<SPLIT>(p.env.pre_horizon), hshaping = keep_s<SPLIT>(p.env.shaping);
    const float hrscale = keep_s<SPLIT>(p.env.reward_scale);
    const float* const hst_env = keep_s<SPLIT, const float*>(p.st.env);
    const float* const hst_x = keep_s<SPLIT, const float*>(p.st.x);
    ActC ha0 = {}, ha1 = {};   // per-action constants of the policy squash and the wrapper chain (actions 0, 1)
    if constexpr (VEH) {
        // (sc / of are float arithmetic, i.e. vector-unit results: the INPUTS are pinned, the two operations stay where they are used)
        const float ph0 = keep_s<SPLIT>(p.env.policy_high[0]), pl0 = keep_s<SPLIT>(p.env.policy_low[0]);
        const float ph1 = keep_s<SPLIT>(p.env.policy_high[1]), pl1 = keep_s<SPLIT>(p.env.policy_low[1]);
        ha0.sc = (ph0 - pl0) / 2.f; ha0.of = (ph0 + pl0) / 2.f;
        ha1.sc = (ph1 - pl1) / 2.f; ha1.of = (ph1 + pl1) / 2.f;
        ha0.min_action = keep_s<SPLIT>(p.env.min_action[0]); ha0.max_action = keep_s<SPLIT>(p.env.max_action[0]);
        ha0.act_low = keep_s<SPLIT>(p.env.act_low[0]); ha0.act_high = keep_s<SPLIT>(p.env.act_high[0]);
        ha1.min_action = keep_s<SPLIT>(p.env.min_action[1]); ha1.max_action = keep_s<SPLIT>(p.env.max_action[1]);
        ha1.act_low = keep_s<SPLIT>(p.env.act_low[1]); ha1.act_high = keep_s<SPLIT>(p.env.act_high[1]);
        ha0.inv_range = exact_inverse_or_zero(ha0.max_action - ha0.min_action);
        ha1.inv_range = exact_inverse_or_zero(ha1.max_action - ha1.min_action);
    }
    auto stage_step = [&](int tt) {
        const size_t r0 = ((size_t)tile * hH + tt) * TB;
        const float* dst = s_stage + (tt & 1) * STAGE_FLOATS;
        const bool gelu_s = p.pol.act == GOPS_ACT_GELU;
        const float* src2 = (gelu_s ? p.st.z[2] : p.st.h[2]) + r0 * 256;
        const float* src1 = (gelu_s ? p.st.z[1] : p.st.h[1]) + r0 * 256;
        const int ln = tid & 63, wv = tid >> 6;
        // FM tiles are 16 KiB of contiguous memory, features 64 w .. 64 w + 63 (wave w's n-tiles) 4 KiB of it
        if constexpr (!SPLIT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                async_copy16_to_lds<GOPS_SWEEP_STAGE_NT>(src2 + wv * 1024 + q * 256 + 4 * ln, dst + wv * 1024 + q * 256);
                async_copy16_to_lds<GOPS_SWEEP_STAGE_NT>(src1 + wv * 1024 + q * 256 + 4 * ln, dst + TB * 256 + wv * 1024 + q * 256);
            }
        }
        if (wv == 0)        // env rows: 16 x 64 B, contiguous
            async_copy16_to_lds<GOPS_SWEEP_STAGE_NT>(hst_env + r0 * ENV_STASH + 4 * ln, dst + STAGE_TILES);
        if (wv == 1 && ln < 2 * TB)   // first 8 observation columns: 8 x 64 B, contiguous in the FM tile -> st_x[i * 16 + m]
            async_copy16_to_lds<GOPS_SWEEP_STAGE_NT>(hst_x + r0 * kp0 + 4 * ln, dst + STAGE_TILES + TB * ENV_STASH);
        if constexpr (SPLIT && ENV == GOPS_ENV_IDPENDULUM) {   // the forward's sub-step parking of the tile: 16 x 512 B, contiguous
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
                async_copy16_to_lds<GOPS_SWEEP_STAGE_NT>(p.st.idp + r0 * IDP_PARK + (wv * 2 + q2) * 256 + 4 * ln, s_idp + (tt & 1) * (TB * IDP_PARK) + (wv * 2 + q2) * 256);
        }
    };
    // pyth_idpendulum on the kernels that stage nothing else (streamed fp32, EXT, streamed-split): the forward's sub-step parking
    // of step tt (p.st.idp, 16 x 512 B contiguous) -> s_idp, issued one step ahead behind the head of step tt + 1 (after_head:
    // the env phase that read s_idp is over, nothing latency-critical follows) and drained by the step's closing barrier.  Without
    // it the env adjoint recomputes the five Euler sub-steps (libm sincosf included): 9.7 - 10.9 k of a 19.5 k-cycle step at cfg1.
    constexpr bool IDPPARK = (ENV == GOPS_ENV_IDPENDULUM) && !STAGE && !F16;
    const bool idp_parked = IDPPARK && p.st.idp != nullptr;
    auto stage_idp = [&](int tt) {
        if constexpr (IDPPARK) {
            if (idp_parked) {
                const size_t r0 = ((size_t)tile * hH + tt) * TB;
                const int ln = tid & 63, wv = tid >> 6;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
                    async_copy16_to_lds<GOPS_SWEEP_STAGE_NT>(p.st.idp + r0 * IDP_PARK + (wv * 2 + q2) * 256 + 4 * ln, s_idp + (wv * 2 + q2) * 256);
            }
        }
    };
    const int ntiles = (p.B + TB - 1) / TB;
    SsOutGrad og = {};   // (SSB with the fused output-layer gradient; otherwise unused)
    do {   // ---- one tile of 16 trajectories ----
    b0 = tile * TB;
    nvalid = min(TB, p.B - b0);
    for (int idx = tid; idx < TB * ldx; idx += NTHREADS) G[idx] = 0.f;
    if constexpr (EXT) {
        if (q.adj_gfo != nullptr) {   // the caller's terminal term: G starts as d(loss)/d(obs_H)
            __syncthreads();
            for (int idx = tid; idx < nvalid * O; idx += NTHREADS) {
                const int m = idx / O, i = idx - m * O;
                G[m * ldx + i] = gptr(q.adj_gfo)[(size_t)(b0 + m) * O + i];
            }
        }
    }
    float gv = (tid < nvalid) ? gptr(q.grad_v)[b0 + tid] : 0.f;
    if constexpr (F16) {
        gv *= f16_grad_scale(gptr(p.gscale)[0]);
    } else if (tid < TB && p.gscale != nullptr) {   // max|grad_v| of the launch for the weight-gradient GEMMs' delta scale
        const float mx = row16_max(fabsf(gv));
        if (tid == 0 && mx < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(p.gscale), __float_as_uint(mx));
    }
    float dy_run = 0.f;   // SPLIT / SSB, thread 0: the largest |delta_y| of the tile's steps (-> gscale[1], below)
    float gc_ext = 0.f, gc_lin = 0.f, gc_int = 0.f;   // SURR: d(loss)/d(constraint sums) of trajectory tid
    if (CSTR && tid < nvalid && q.in.grad_constraint != nullptr) {
        const GLOBAL_AS float* gcp = gptr(q.in.grad_constraint) + b0 + tid;
        gc_ext = gcp[0]; gc_lin = gcp[(size_t)p.B]; gc_int = gcp[(size_t)2 * p.B];
    }
    float gc_mul[GOPS_MAX_CONSTRAINT] = {0.f, 0.f, 0.f};   // SPIL: d(loss)/d(P_k) * P_k of trajectory tid
    if (CSTR && tid < nvalid && q.in.grad_constraint_prod != nullptr) {
#pragma unroll
        for (int k = 0; k < GOPS_MAX_CONSTRAINT; ++k)
            if (k < p.env.n_constraint) gc_mul[k] = gptr(q.in.grad_constraint_prod)[(size_t)k * p.B + b0 + tid];
    }
    // d(loss)/d(c_tk) handed in per step (GopsRolloutIn.grad_constraint_step), trajectory m of this tile
    auto gc_step = [&](int t, int m, int k) -> float {
        if (!CSTR || q.in.grad_constraint_step == nullptr) return 0.f;
        return gptr(q.in.grad_constraint_step)[((size_t)t * p.B + b0 + m) * p.env.n_constraint + k];
    };
    float lam[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // adjoint of the veh3dof state (tid < TB)
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
    if (TAIL) {
        if (tid < TB) {
            const float dH = (tid < nvalid) ? gptr(p.st.tail_done)[b0 + tid] : 1.f;
            s_gy[tid * 4 + 0] = gv * ((p.tail_unmasked ? 1.f : 1.f - dH) * p.gpow[p.H]);
            s_gy[tid * 4 + 1] = s_gy[tid * 4 + 2] = s_gy[tid * 4 + 3] = 0.f;
        }
        if constexpr (SSB) {
            if (tid < TB) {
                const float mx = row16_max(fabsf(s_gy[tid * 4]));
                if (tid == 0) split_delta_scale(mx, s_scale);
            }
        }
        __syncthreads();
        if constexpr (SSB)
            ss_net_backward(p.val, p.ssvt, gptr(p.val.w[p.val.nl - 1]), p.val.dims[p.val.nl - 1], s_gy, s_scale, dq2, G, ldx, tid,
                            p.st.tail_h, p.st.tail_z, nullptr, nullptr, (size_t)b0, nvalid, true, O, [] {}, og, false);
        else
        if constexpr (F16)
            mlp_backward_h(p.val, gptr(p.val.w[p.val.nl - 1]), p.val.dims[p.val.nl - 1], s_gy, reinterpret_cast<_Float16*>(da),
                           reinterpret_cast<_Float16*>(db), ld16, G, ldx, tid, p.st.tail_h, p.st.tail_z, nullptr, nullptr,
                           (size_t)b0, nvalid, true, O, [] {});
        else
        mlp_backward<false>(p.val, NoW{}, NoW{}, gptr(p.val.w[p.val.nl - 1]), p.val.dims[p.val.nl - 1], s_gy, da, db, ldh, G, ldx,
                     tid, p.st.tail_h, p.st.tail_z, nullptr, nullptr,
                     (size_t)b0, nvalid, true, O, dbg, [] {});
    }
    __syncthreads();

    if constexpr (STAGE) {
        stage_step(hH - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if constexpr (IDPPARK) {
        if (idp_parked) {
            stage_idp(hH - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    unsigned l2_sink = 0, l2_pf[TOUCH_SLOTS] = {0u, 0u, 0u, 0u};
    for (int t = hH - 1; t >= 0; --t) {
        const size_t row0 = ((size_t)tile * hH + t) * TB;   // tile-major stash rows
        const size_t prow = row0 - TB;
        const float* st_cur = s_stage + (t & 1) * STAGE_FLOATS;       // this step's staged data (STAGE only)
        const float* st_env = st_cur + STAGE_TILES;
        const float* st_x = st_env + TB * ENV_STASH;
        // (the staged variants fetch their tiles straight into LDS and measured faster without it)
        const int tmode = (t > 0 && !STAGE && !F16) ? p.touch_mode : 0;
        // L2 warm-up of what step t-1 will read (written long ago by the forward kernel): HBM-latency
        // loads whose values are only XOR-ed into a sink at the end of the step.  Loads return in order,
        // so they are issued where no latency-critical load follows for thousands of cycles - right
        // after the head, ahead of the hidden-layer GEMMs - never in front of the env-row / act' loads.
        auto warm_up = [&]() {
            if (t > 0) stage_idp(t - 1);
            if (tmode != 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) l2_pf[q] = touch_fetch(p.pol, p.st, prow, tid + NTHREADS * q);
            }
        };
        DBG_TICK(0)
        float g_r = gv * p.gpow[t];                         // adjoint of the shaped reward
        if (ENV != GOPS_ENV_NONE && hshaping) g_r *= hrscale;
        if constexpr (STAGE) {
            // Next step's tiles travel HBM -> LDS during this whole step.  Issued AFTER the first use of a
            // loaded value in the iteration (g_r above): hipcc drains vmcnt(0) there on every trip, and the
            // copies must not be outstanding at that point.
            if (t > 0) stage_step(t - 1);
            if constexpr (SPLIT) SS.fetch(p, 2, row0, tid);   // act' operands of the head delta: in flight during the env adjoint
        }

        if (ENV == GOPS_ENV_NONE) {
            if (tid < TB) {
                s_gy[tid * 4 + 0] = g_r;
                s_gy[tid * 4 + 1] = s_gy[tid * 4 + 2] = s_gy[tid * 4 + 3] = 0.f;
            }
        } else if (ENV == GOPS_ENV_LQ && !EXT) {
            // pyth_lq without ActionRepeat / adjoint I/O: the arithmetic of the generic block below, as a routine with compile-time
            // loop bounds - (4, 2) for BASELINE configs[4] (lq s4a2), the maxima otherwise.  The generic form (run-time guards on
            // every state / action index, ~100 scalars of the description live at once) took 9.9 k of the 32 k cycles of a
            // streamed-split sweep step at cfg5, on 16 lanes.
            if (tid < TB) {
                const int m = tid;
                auto lq_adjoint = [&]<int NS, int NA>() {
                    constexpr bool EXACT = NS < GOPS_MAX_LQ_STATE;   // the dimensions ARE (NS, NA): no run-time guards
                    float th[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, dflag = 1.f;
                    float x[GOPS_MAX_LQ_STATE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (m < nvalid) {
                        f32x4 e0, e1;
                        if constexpr (STAGE) {
                            const f32x4* er = reinterpret_cast<const f32x4*>(st_env + m * ENV_STASH);
                            e0 = er[0]; e1 = er[1];
#pragma unroll
                            for (int i = 0; i < NS; ++i)
                                if (EXACT || i < O) x[i] = st_x[i * 16 + m];
                        } else {
                            const GLOBAL_AS f32x4* er = gptr(reinterpret_cast<const f32x4*>(p.st.env + (row0 + m) * ENV_STASH));
                            e0 = er[0]; e1 = er[1];
#pragma unroll
                            for (int i = 0; i < NS; ++i)
                                if (EXACT || i < O) x[i] = x_col(row0, m, i);
                        }
                        th[0] = e0[0]; th[1] = e0[1]; th[2] = e0[2]; th[3] = e0[3];
                        dflag = e1[0];
#pragma unroll
                        for (int i = 0; i < NS; ++i)
                            if (EXACT || i < O) x[i] = obs_unscale(env, i, x[i]);   // the stash holds the (scaled) policy input
                    }
                    float abar[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, u[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, sc[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f},
                          gu[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        sc[a] = (env.policy_high[a] - env.policy_low[a]) / 2.f;
                        abar[a] = sc[a] * th[a] + (env.policy_high[a] + env.policy_low[a]) / 2.f;
                        u[a] = (EXACT || a < A) ? (p.open_loop == 2 ? th[a] : wrap_action(env, a, abar[a])) : 0.f;   // open_loop 2: raw actions
                    }
                    const bool dn = dflag != 0.f;
                    const float g_rm = dn ? 0.f : g_r;
                    float Gin[GOPS_MAX_LQ_STATE], gx[GOPS_MAX_LQ_STATE];
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { Gin[i] = (i < NS && (EXACT || i < O)) ? G[m * ldx + i] : 0.f; gx[i] = 0.f; }
                    if (env.clip_obs) {
                        float xn[GOPS_MAX_LQ_STATE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rdummy;
                        lq_forward<NS, NA>(env, x, u, xn, rdummy);
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const float pre = obs_rescale(env, i, dn ? x[i] : xn[i]);   // what ClipObservation saw
                            if ((EXACT || i < O) && !(pre >= env.obs_low[i] && pre <= env.obs_high[i])) Gin[i] = 0.f;
                        }
                    }
                    if (env.scale_obs) {   // d(scaled next obs) / d(next obs) = scale
#pragma unroll
                        for (int i = 0; i < NS; ++i)
                            if (EXACT || i < O) Gin[i] *= env.obs_scale[i];
                    }
                    float gxn[GOPS_MAX_LQ_STATE];
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { gxn[i] = dn ? 0.f : Gin[i]; gx[i] = dn ? Gin[i] : 0.f; }
                    lq_backward<NS, NA>(env, x, u, gxn, g_rm, gx, gu);
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        if (EXACT || i < O) G[m * ldx + i] = env.scale_obs ? gx[i] / env.obs_scale[i] : gx[i];   // d(obs / scale - shift) / d(obs)
#pragma unroll
                    for (int a = 0; a < GOPS_MAX_ACT; ++a)
                        s_gy[m * 4 + a] = (a < NA && (EXACT || a < A))
                                              ? (p.open_loop == 2 ? gu[a] : wrap_action_bwd(env, a, abar[a], gu[a]) * sc[a] * (1.f - th[a] * th[a]))
                                              : 0.f;
                };
                if (O == 4 && A == 2) lq_adjoint.template operator()<4, 2>();
                else lq_adjoint.template operator()<GOPS_MAX_LQ_STATE, GOPS_MAX_ACT>();
            }
        } else if (ENV == GOPS_ENV_LQ || ENV == GOPS_ENV_IDPENDULUM || ENV == GOPS_ENV_CARTPOLE || ENV == GOPS_ENV_PENDULUM) {
            if (tid < TB) {
                const int m = tid;
                float th[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, dflag = 1.f;
                float x[GOPS_MAX_LQ_STATE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (m < nvalid) {
                    f32x4 e0, e1;
                    if constexpr (STAGE) {
                        const f32x4* er = reinterpret_cast<const f32x4*>(st_env + m * ENV_STASH);
                        e0 = er[0]; e1 = er[1];
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                            if (i < O) x[i] = st_x[i * 16 + m];
                    } else if (IDPPARK && idp_parked) {
                        // tanh(head) and the done flag ride in the parking (slots 126 / 127, rollout_fwd.hip): already in LDS, and the
                        // observation is not needed (the sub-step states are parked) - no global load at the top of the step
                        const float* pk = s_idp + m * IDP_PARK;
                        e0 = f32x4{pk[126], 0.f, 0.f, 0.f};
                        e1 = f32x4{pk[127], 0.f, 0.f, 0.f};
                    } else {
                        const GLOBAL_AS f32x4* er = gptr(reinterpret_cast<const f32x4*>(p.st.env + (row0 + m) * ENV_STASH));
                        e0 = er[0]; e1 = er[1];
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                            if (i < O) x[i] = x_col(row0, m, i);
                    }
                    th[0] = e0[0]; th[1] = e0[1]; th[2] = e0[2]; th[3] = e0[3];
                    dflag = e1[0];
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                        if (i < O) x[i] = obs_unscale(p.env, i, x[i]);   // the stash holds the (scaled) policy input
                }
                float abar[GOPS_MAX_ACT], u[GOPS_MAX_ACT], sc[GOPS_MAX_ACT], gu[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a) {
                    sc[a] = (p.env.policy_high[a] - p.env.policy_low[a]) / 2.f;
                    abar[a] = sc[a] * th[a] + (p.env.policy_high[a] + p.env.policy_low[a]) / 2.f;
                    u[a] = (a < A) ? (p.open_loop == 2 ? th[a] : wrap_action(p.env, a, abar[a])) : 0.f;   // open_loop 2: raw actions
                }
                const bool dn = dflag != 0.f;
                const float g_rm = dn ? 0.f : g_r;
                float Gin[GOPS_MAX_LQ_STATE], gx[GOPS_MAX_LQ_STATE];
#pragma unroll
                for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { Gin[i] = (i < O) ? G[m * ldx + i] : 0.f; gx[i] = 0.f; }
                bool repeated = false;
                if constexpr (EXT) {
                    if (p.env.repeat_num > 1) {
                        // ActionRepeatModel: x^(k+1) = step(x^(k), u), k < n, with the initial done flag; the sub-step
                        // states are recomputed from the stashed observation, then the sub-steps are walked backwards
                        // (every sub-step's reward carries g_r, or only the last one's)
                        repeated = true;
                        const int nrep = min(p.env.repeat_num, GOPS_MAX_REPEAT);
                        auto fwd1 = [&](const float* xi, float* xo) {
                            float rd = 0.f;
                            if constexpr (ENV == GOPS_ENV_LQ) {
                                lq_forward(p.env, xi, u, xo, rd);
                            } else if constexpr (ENV == GOPS_ENV_CARTPOLE) {
                                bool dd;
                                cart_forward(cart_const(), xi, u[0], xo, rd, dd);
                            } else if constexpr (ENV == GOPS_ENV_PENDULUM) {
                                PendStep w;
                                pend_forward(xi, u[0], xo, rd, w);
                            } else {
                                float s5[6];
#pragma unroll
                                for (int i = 0; i < 6; ++i) s5[i] = xi[i];
                                IdpSub w;
                                idp_substep<true>(IC, s5, 500.f * u[0], 0.002f, xo, w);
#pragma unroll 1
                                for (int k = 1; k < 5; ++k) {
                                    idp_advance_trig(s5, 0.002f, w, w);
#pragma unroll
                                    for (int i = 0; i < 6; ++i) s5[i] = xo[i];
                                    idp_substep<false>(IC, s5, 500.f * u[0], 0.002f, xo, w);
                                }
                            }
                        };
                        auto bwd1 = [&](const float* xi, const float* gn, float gr, float* gxo, float* guo) {
#pragma unroll
                            for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) gxo[i] = 0.f;
                            if constexpr (ENV == GOPS_ENV_LQ) {
                                lq_backward(p.env, xi, u, gn, gr, gxo, guo);
                            } else if constexpr (ENV == GOPS_ENV_CARTPOLE) {
                                cart_backward(cart_const(), xi, u[0], gn, gxo, guo[0]);
                            } else if constexpr (ENV == GOPS_ENV_PENDULUM) {
                                pend_backward(xi, u[0], gn, gr, gxo, guo[0]);
                            } else {
                                float* park = s_idp + m * (5 * 24);
                                float sc_[6], sn_[6];
#pragma unroll
                                for (int i = 0; i < 6; ++i) sc_[i] = xi[i];
                                const float a = u[0], force = 500.f * a;
                                IdpSub w;
#pragma unroll 1
                                for (int k = 0; k < 5; ++k) {
                                    if (k == 0) idp_substep<true>(IC, sc_, force, 0.002f, sn_, w);
                                    else idp_substep<false>(IC, sc_, force, 0.002f, sn_, w);
                                    float* pk = park + k * 24;
#pragma unroll
                                    for (int i = 0; i < 6; ++i) pk[i] = sc_[i];
                                    pk[6] = w.s1; pk[7] = w.c1; pk[8] = w.s2; pk[9] = w.c2; pk[10] = w.s12; pk[11] = w.c12;
#pragma unroll
                                    for (int i = 0; i < 6; ++i) pk[12 + i] = w.inv[i];
#pragma unroll
                                    for (int i = 0; i < 3; ++i) pk[18 + i] = w.qdd[i];
                                    idp_advance_trig(sc_, 0.002f, w, w);
#pragma unroll
                                    for (int i = 0; i < 6; ++i) sc_[i] = sn_[i];
                                }
                                float g[6];
#pragma unroll
                                for (int i = 0; i < 6; ++i) g[i] = gn[i];
                                g[1] += gr * (-10.f * sc_[1]);
                                g[2] += gr * (-20.f * sc_[2]);
                                g[3] += gr * (-1.f * sc_[3]);
                                g[4] += gr * (-1.f * sc_[4]);
                                g[5] += gr * (-2.f * sc_[5]);
                                float gforce = 0.f;
#pragma unroll 1
                                for (int k = 4; k >= 0; --k) {
                                    const float* pk = park + k * 24;
                                    IdpSub wk;
                                    float sk[6];
#pragma unroll
                                    for (int i = 0; i < 6; ++i) sk[i] = pk[i];
                                    wk.s1 = pk[6]; wk.c1 = pk[7]; wk.s2 = pk[8]; wk.c2 = pk[9]; wk.s12 = pk[10]; wk.c12 = pk[11];
#pragma unroll
                                    for (int i = 0; i < 6; ++i) wk.inv[i] = pk[12 + i];
#pragma unroll
                                    for (int i = 0; i < 3; ++i) wk.qdd[i] = pk[18 + i];
                                    idp_substep_bwd(IC, sk, 0.002f, wk, g, gforce);
                                }
                                guo[0] = 500.f * gforce + gr * (-2.f * a);
#pragma unroll
                                for (int i = 0; i < 6; ++i) gxo[i] = g[i];
                            }
                        };
                        float xk[GOPS_MAX_REPEAT][GOPS_MAX_LQ_STATE], xfin[GOPS_MAX_LQ_STATE];
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { xk[0][i] = x[i]; xfin[i] = x[i]; }
#pragma unroll
                        for (int rep = 1; rep < GOPS_MAX_REPEAT; ++rep) {
#pragma unroll
                            for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) xk[rep][i] = 0.f;
                            if (rep < nrep && !dn) fwd1(xk[rep - 1], xk[rep]);
                        }
                        if (p.env.clip_obs && ENV != GOPS_ENV_IDPENDULUM) {   // ClipObservation saw the last sub-step's (rescaled) result
                            if (!dn) {
#pragma unroll
                                for (int rep = 0; rep < GOPS_MAX_REPEAT; ++rep)
                                    if (rep == nrep - 1) fwd1(xk[rep], xfin);
                            }
#pragma unroll
                            for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) {
                                const float pre = obs_rescale(p.env, i, xfin[i]);
                                if (i < O && !(pre >= p.env.obs_low[i] && pre <= p.env.obs_high[i])) Gin[i] = 0.f;
                            }
                        }
                        if (p.env.scale_obs) {
#pragma unroll
                            for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                                if (i < O) Gin[i] *= p.env.obs_scale[i];
                        }
                        float g[GOPS_MAX_LQ_STATE];
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) g[i] = dn ? 0.f : Gin[i];
#pragma unroll
                        for (int rep = GOPS_MAX_REPEAT - 1; rep >= 0; --rep) {
                            if (rep < nrep && !dn) {
                                const float gr = (!p.env.repeat_last_reward || rep == nrep - 1) ? g_rm : 0.f;
                                float gxo[GOPS_MAX_LQ_STATE], guo[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
                                bwd1(xk[rep], g, gr, gxo, guo);
#pragma unroll
                                for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) g[i] = gxo[i];
#pragma unroll
                                for (int a = 0; a < GOPS_MAX_ACT; ++a) gu[a] += guo[a];
                            }
                        }
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) gx[i] = dn ? Gin[i] : g[i];
                    }
                }
                if (repeated) {
                } else if (ENV == GOPS_ENV_CARTPOLE || ENV == GOPS_ENV_PENDULUM) {
                    constexpr int NS = (ENV == GOPS_ENV_CARTPOLE) ? 4 : 3;
                    if (p.env.clip_obs) {   // ClipObservation on the (rescaled) next observation
                        float xn[4] = {0.f, 0.f, 0.f, 0.f}, rdummy;
                        bool ddummy;
                        PendStep wdummy;
                        if (ENV == GOPS_ENV_CARTPOLE) cart_forward(cart_const(), x, u[0], xn, rdummy, ddummy);
                        else pend_forward(x, u[0], xn, rdummy, wdummy);
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const float pre = obs_rescale(p.env, i, dn ? x[i] : xn[i]);
                            if (!(pre >= p.env.obs_low[i] && pre <= p.env.obs_high[i])) Gin[i] = 0.f;
                        }
                    }
                    if (p.env.scale_obs) {
#pragma unroll
                        for (int i = 0; i < NS; ++i) Gin[i] *= p.env.obs_scale[i];
                    }
                    float gxn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NS; ++i) { gxn[i] = dn ? 0.f : Gin[i]; gx[i] = dn ? Gin[i] : 0.f; }
                    if (ENV == GOPS_ENV_CARTPOLE) cart_backward(cart_const(), x, u[0], gxn, gx, gu[0]);
                    else pend_backward(x, u[0], gxn, g_rm, gx, gu[0]);
                } else if (ENV == GOPS_ENV_LQ) {
                    if (p.env.clip_obs) {
                        float xn[GOPS_MAX_LQ_STATE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rdummy;
                        lq_forward(p.env, x, u, xn, rdummy);
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) {
                            const float pre = obs_rescale(p.env, i, dn ? x[i] : xn[i]);   // what ClipObservation saw
                            if (i < O && !(pre >= p.env.obs_low[i] && pre <= p.env.obs_high[i])) Gin[i] = 0.f;
                        }
                    }
                    if (p.env.scale_obs) {   // d(scaled next obs) / d(next obs) = scale
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                            if (i < O) Gin[i] *= p.env.obs_scale[i];
                    }
                    float gxn[GOPS_MAX_LQ_STATE];
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { gxn[i] = dn ? 0.f : Gin[i]; gx[i] = dn ? Gin[i] : 0.f; }
                    lq_backward(p.env, x, u, gxn, g_rm, gx, gu);
                } else {
                    // Recompute the 5 Euler sub-steps once, parking each sub-step's input state and
                    // intermediates (M^-1, qdd, sin/cos) in LDS; then walk them backwards from there.
                    if (p.env.scale_obs) {   // d(scaled next obs) / d(next obs) = scale
#pragma unroll
                        for (int i = 0; i < 6; ++i) Gin[i] *= p.env.obs_scale[i];
                    }
                    float sc_[6];
                    const float a = u[0], force = 500.f * a;
                    const float* park;   // [5][24]: sub-step k's input state, sin / cos, M^-1, qdd
                    if constexpr (SPLIT && STAGE) {
                        // parked by the forward (p.st.idp), staged one step ahead: nothing is recomputed
                        park = s_idp + (t & 1) * (TB * IDP_PARK) + m * IDP_PARK;
#pragma unroll
                        for (int i = 0; i < 6; ++i) sc_[i] = park[120 + i];   // the state after the step
                    } else if (idp_parked) {   // parked by the forward, copied one step ahead (stage_idp)
                        park = s_idp + m * IDP_PARK;
#pragma unroll
                        for (int i = 0; i < 6; ++i) sc_[i] = park[120 + i];
                    } else {
                        float* parkw = s_idp + m * (5 * 24);
                        park = parkw;
                        float sn_[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) sc_[i] = x[i];
                        IdpSub w;
#pragma unroll 1
                        for (int k = 0; k < 5; ++k) {
                            if (k == 0) idp_substep<true>(IC, sc_, force, 0.002f, sn_, w);
                            else idp_substep<false>(IC, sc_, force, 0.002f, sn_, w);   // w.s1 .. w.c2 advanced at the end of the last trip
                            float* pk = parkw + k * 24;
#pragma unroll
                            for (int i = 0; i < 6; ++i) pk[i] = sc_[i];
                            pk[6] = w.s1; pk[7] = w.c1; pk[8] = w.s2; pk[9] = w.c2; pk[10] = w.s12; pk[11] = w.c12;
#pragma unroll
                            for (int i = 0; i < 6; ++i) pk[12 + i] = w.inv[i];
#pragma unroll
                            for (int i = 0; i < 3; ++i) pk[18 + i] = w.qdd[i];
                            idp_advance_trig(sc_, 0.002f, w, w);
#pragma unroll
                            for (int i = 0; i < 6; ++i) sc_[i] = sn_[i];
                        }
                    }
                    float g[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) g[i] = dn ? 0.f : Gin[i];
                    g[1] += g_rm * (-10.f * sc_[1]);
                    g[2] += g_rm * (-20.f * sc_[2]);
                    g[3] += g_rm * (-1.f * sc_[3]);
                    g[4] += g_rm * (-1.f * sc_[4]);
                    g[5] += g_rm * (-2.f * sc_[5]);
                    float gforce = 0.f;
#if GOPS_IDP_BWD_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
                    for (int k = 4; k >= 0; --k) {
                        const float* pk = park + k * 24;
                        IdpSub w;
                        float sk[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) sk[i] = pk[i];
                        w.s1 = pk[6]; w.c1 = pk[7]; w.s2 = pk[8]; w.c2 = pk[9]; w.s12 = pk[10]; w.c12 = pk[11];
#pragma unroll
                        for (int i = 0; i < 6; ++i) w.inv[i] = pk[12 + i];
#pragma unroll
                        for (int i = 0; i < 3; ++i) w.qdd[i] = pk[18 + i];
                        idp_substep_bwd(IC, sk, 0.002f, w, g, gforce);
                    }
                    gu[0] = 500.f * gforce + g_rm * (-2.f * a);
#pragma unroll
                    for (int i = 0; i < 6; ++i) gx[i] = g[i] + (dn ? Gin[i] : 0.f);
                }
#pragma unroll
                for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                    if (i < O) G[m * ldx + i] = p.env.scale_obs ? gx[i] / p.env.obs_scale[i] : gx[i];   // d(obs / scale - shift) / d(obs)
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a)
                    s_gy[m * 4 + a] = (a < A) ? (p.open_loop == 2 ? gu[a] : wrap_action_bwd(p.env, a, abar[a], gu[a]) * sc[a] * (1.f - th[a] * th[a])) : 0.f;
            }
        } else if (ENV == GOPS_ENV_MOBILEROBOT) {
            if (tid < TB) {
                const int m = tid;
                const MobConst MC = mob_const();
                float th[2] = {0.f, 0.f}, dflag = 1.f, nv = 0.f, nw = 0.f;
                float x[MOB_OBS];
#pragma unroll
                for (int i = 0; i < MOB_OBS; ++i) x[i] = 0.f;
                if (m < nvalid) {
                    const GLOBAL_AS f32x4* er = gptr(reinterpret_cast<const f32x4*>(p.st.env + (row0 + m) * ENV_STASH));
                    const f32x4 e0 = er[0], e1 = er[1];
                    th[0] = e0[0]; th[1] = e0[1]; dflag = e1[0];
#pragma unroll
                    for (int i = 0; i < MOB_OBS; ++i) x[i] = x_col(row0, m, i);
                    if (q.in.noise != nullptr) {
                        const GLOBAL_AS float* nz = gptr(q.in.noise) + ((size_t)t * p.B + b0 + m) * 2;
                        nv = nz[0]; nw = nz[1];
                    }
                }
                float abar[2], u[2], sc[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    sc[a] = (p.env.policy_high[a] - p.env.policy_low[a]) / 2.f;
                    abar[a] = sc[a] * th[a] + (p.env.policy_high[a] + p.env.policy_low[a]) / 2.f;
                    u[a] = p.open_loop == 2 ? th[a] : wrap_action(p.env, a, abar[a]);
                }
                const bool dn = dflag != 0.f;
                const float g_rm = dn ? 0.f : g_r;
                float xn[MOB_OBS], rdummy, c;
                bool ddummy;
                MobStep w;
                mob_forward(MC, x, u[0], u[1], nv, nw, xn, rdummy, c, ddummy, w);
                float gxn[MOB_OBS], gx[MOB_OBS];
#pragma unroll
                for (int i = 0; i < MOB_OBS; ++i) {
                    float gi = G[m * ldx + i];
                    if (p.env.clip_obs) {   // ClipObservation on the (masked) next observation
                        const float pre = dn ? x[i] : xn[i];
                        if (!(pre >= p.env.obs_low[i] && pre <= p.env.obs_high[i])) gi = 0.f;
                    }
                    gxn[i] = dn ? 0.f : gi;   // MaskAtDone: a finished trajectory's observation is frozen ...
                    gx[i] = dn ? gi : 0.f;
                }
                float gck = 0.f;               // ... but info["constraint"] still comes from the model's step
                if (m < nvalid) {
                    gck = gc_ext * 2.f * fmaxf(c, 0.f) + (c > 0.f ? gc_lin : 0.f);
                    if (c < 0.f) gck += gc_int * (-1.f / (-c + 1e-8f));
                    gck *= p.gpow[t];
                    float dlog;
                    (void)spil_phi(c, dlog);
                    gck += gc_mul[0] * dlog;
                    gck += gc_step(t, m, 0);
                }
                float gu[2];
                mob_backward(MC, x, u[0], u[1], nv, nw, gxn, g_rm, gck, gx, gu);
#pragma unroll
                for (int i = 0; i < MOB_OBS; ++i) G[m * ldx + i] = gx[i];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    s_gy[m * 4 + a] = p.open_loop == 2 ? gu[a] : wrap_action_bwd(p.env, a, abar[a], gu[a]) * sc[a] * (1.f - th[a] * th[a]);
                s_gy[m * 4 + 2] = s_gy[m * 4 + 3] = 0.f;
            }
        } else if (ENV == GOPS_ENV_VEH2DOF) {
            if (tid < TB) {
                const int m = tid, P = p.env.pre_horizon;
                const Veh2Const C2 = veh2_const();
                float th0 = 0.f, dflag = 1.f, steer = 0.f, s[4] = {0.f, 0.f, 0.f, 0.f}, o[4] = {0.f, 0.f, 0.f, 0.f};
                if (m < nvalid) {
                    const GLOBAL_AS f32x4* er = gptr(reinterpret_cast<const f32x4*>(p.st.env + (row0 + m) * ENV_STASH));
                    const f32x4 e0 = er[0], e1 = er[1], e2 = er[2];
                    th0 = e0[0]; steer = e0[2]; dflag = e1[0];
                    s[0] = e1[1]; s[1] = e1[2]; s[2] = e1[3]; s[3] = e2[0];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = x_col(row0, m, i);
                }
                const bool dn = dflag != 0.f;
                // adjoint of the next state: what later steps left in `lam` plus this step's observation adjoint
                // (MaskAtDone: a finished trajectory's observation is frozen, its adjoint stays on obs_t)
                float* gp = G + m * ldx;
                float ln[4] = {lam[0], lam[1], lam[2], lam[3]};
                if (!dn) {
                    float gy = gp[0];
                    for (int i = 1; i <= P; ++i) { gy += gp[3 + i]; gp[3 + i] = 0.f; }
                    ln[0] += gy; ln[1] += gp[1]; ln[2] += gp[2]; ln[3] += gp[3];
                    gp[0] = gp[1] = gp[2] = gp[3] = 0.f;
                }
                float sphi, cphi, g_steer;
                sincosf(s[1], &sphi, &cphi);
                float l[4];
                veh2_f_xu_bwd(C2, s, sphi, cphi, ln, l, g_steer);
#pragma unroll
                for (int i = 0; i < 4; ++i) lam[i] = l[i];
                const float g_rm = dn ? 0.f : g_r;
                if (m < nvalid) {
                    gp[0] += g_rm * (-0.08f * o[0]);
                    gp[1] += g_rm * (-0.04f * o[1]);
                    gp[2] += g_rm * (-0.02f * o[2]);
                    gp[3] += g_rm * (-0.02f * o[3]);
                    if (p.env.cstr_err) {   // errcstr: c = |obs[0]| - tol of THIS observation (unmasked sums / products)
                        const float c = fabsf(o[0]) - p.env.err_tol[0];
                        float gck = gc_ext * 2.f * fmaxf(c, 0.f) + (c > 0.f ? gc_lin : 0.f);
                        if (c < 0.f) gck += gc_int * (-1.f / (-c + 1e-8f));
                        gck *= p.gpow[t];
                        float dlog;
                        (void)spil_phi(c, dlog);
                        gck += gc_mul[0] * dlog;
                        gck += gc_step(t, m, 0);
                        gp[0] += gck * (o[0] > 0.f ? 1.f : (o[0] < 0.f ? -1.f : 0.f));
                    }
                }
                g_steer += g_rm * (-0.02f * steer);
                const float sc0 = (p.env.policy_high[0] - p.env.policy_low[0]) / 2.f;
                const float abar0 = sc0 * th0 + (p.env.policy_high[0] + p.env.policy_low[0]) / 2.f;
                s_gy[m * 4 + 0] = p.open_loop == 2 ? g_steer : wrap_action_bwd(p.env, 0, abar0, g_steer) * sc0 * (1.f - th0 * th0);
                s_gy[m * 4 + 1] = s_gy[m * 4 + 2] = s_gy[m * 4 + 3] = 0.f;
            }
        } else {   // GOPS_ENV_VEH3DOFCONTI
            const int m = tid & 15, part = tid >> 4, lane = tid & 63, wave = tid >> 6;
            const int P = hP;
            float th0 = 0.f, th1 = 0.f, dflag = 1.f, st_steer = 0.f, st_ax = 0.f;
            float s[6] = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f};
            f32x4 e3 = {0.f, 1.f, 0.f, 1.f};   // sin / cos of the heading before and after the step (forward's values)
            if (m < nvalid) {
                f32x4 e0, e1, e2;
                if constexpr (STAGE) {
                    const f32x4* er = reinterpret_cast<const f32x4*>(st_env + m * ENV_STASH);
                    e0 = er[0]; e1 = er[1]; e2 = er[2]; e3 = er[3];
                } else {
                    const GLOBAL_AS f32x4* er = gptr(reinterpret_cast<const f32x4*>(p.st.env + (row0 + m) * ENV_STASH));
                    e0 = er[0]; e1 = er[1]; e2 = er[2]; e3 = er[3];
                }
                th0 = e0[0]; th1 = e0[1]; st_steer = e0[2]; st_ax = e0[3]; dflag = e1[0];
                s[0] = e1[1]; s[1] = e1[2]; s[2] = e1[3]; s[3] = e2[0]; s[4] = e2[1]; s[5] = e2[2];
            }
            const bool dn = dflag != 0.f;
            const float sc0 = ha0.sc, sc1 = ha1.sc;
            const float abar0 = sc0 * th0 + ha0.of;
            const float abar1 = sc1 * th1 + ha1.of;
            const float steer = st_steer, ax = st_ax;   // = wrap_action(abar0 / abar1), stashed by the forward kernel
            float sn[6];
            VehStep w;
            w.sphi = e3[0]; w.cphi = e3[1];
            veh_f_xu(VC, s, steer, ax, sn, w);
            const float cn = e3[3], snn = -e3[2];   // cos(-phi') = cos(phi'), sin(-phi') = -sin(phi')
            // partial adjoints of (x', y', phi', u') and of cos/sin(-phi') over this thread's points
            float px = 0.f, py = 0.f, pphi = 0.f, pu = 0.f, pc = 0.f, ps = 0.f, g4 = 0.f, g5 = 0.f;
            const f32x4* tbl = s_ref + m * TL + (t + 1);
            for (int j = part; j <= P; j += 16) {
                float* gp = G + m * ldx + (j == 0 ? 0 : 6 + 4 * (j - 1));
                float gx_ = gp[0], gy_ = gp[1], gph = gp[2], gu_ = gp[3];
                if (j == 0) { g4 = gp[4]; g5 = gp[5]; }
                if (dn) {
                    gx_ = gy_ = gph = gu_ = 0.f; g4 = g5 = 0.f;      // MaskAtDone: adjoint stays on obs_t
                } else {
                    gp[0] = gp[1] = gp[2] = gp[3] = 0.f;
                    if (j == 0) { gp[4] = 0.f; gp[5] = 0.f; }
                }
                const f32x4 rp = tbl[j];
                const float dx = rp[0] - sn[0], dy = rp[1] - sn[1];
                px -= gx_ * cn + gy_ * snn;
                py -= -gx_ * snn + gy_ * cn;
                pc += gx_ * dx + gy_ * dy;
                ps += -gx_ * dy + gy_ * dx;
                pphi -= gph;
                pu -= gu_;
            }
            DBG_TICK(10)
            // reduce over the 16 `part` threads of each trajectory: lanes m+16q in-wave, then 4 waves
            float v6[6] = {px, py, pphi, pu, pc, ps};
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                v6[i] += __shfl_xor(v6[i], 16);
                v6[i] += __shfl_xor(v6[i], 32);
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 6; ++i) red[(wave * TB + m) * 8 + i] = v6[i];
            }
            __syncthreads();
            DBG_TICK(11)
            if (tid < TB) {
                float tot[6];
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    tot[i] = red[(0 * TB + m) * 8 + i] + red[(1 * TB + m) * 8 + i] + red[(2 * TB + m) * 8 + i] + red[(3 * TB + m) * 8 + i];
                float lamn[6];
                lamn[0] = lam[0] + tot[0];
                lamn[1] = lam[1] + tot[1];
                lamn[2] = lam[2] + tot[2] + tot[4] * snn - tot[5] * cn;
                lamn[3] = lam[3] + tot[3];
                lamn[4] = lam[4] + g4;
                lamn[5] = lam[5] + g5;
                if constexpr (SURR) {
                    if (m < nvalid) {
                        const int ns = p.env.n_surr;
                        const GLOBAL_AS f32x4* sp = gptr(p.surr_table) + ((size_t)(b0 + m) * (p.H + 1) + (t + 1)) * ns;
                        f32x4 pts[GOPS_MAX_SURR];
#pragma unroll
                        for (int i = 0; i < GOPS_MAX_SURR; ++i) {
                            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                            pts[i] = (i < ns) ? sp[i] : z;
                        }
                        // adjoint of the (unmasked) constraint sums w.r.t. the new ego pose
                        SurrCstr sc;
                        surr_constraint<true>(p.env, sn[0], sn[1], e3[2], e3[3], pts, sc);
#pragma unroll
                        for (int k = 0; k < GOPS_MAX_CONSTRAINT; ++k) {
                            if (k < ((p.env.surr_penalty || p.env.cstr_err) ? 0 : p.env.n_constraint)) {
                                const float c = sc.c[k];
                                float gck = gc_ext * 2.f * fmaxf(c, 0.f) + (c > 0.f ? gc_lin : 0.f);
                                if (c < 0.f) gck += gc_int * (-1.f / (-c + 1e-8f));
                                gck *= p.gpow[t];
                                float dlog;
                                (void)spil_phi(c, dlog);
                                gck += gc_mul[k] * dlog;   // d P_k / d c_tk = P_k Phi'(c_tk) / Phi(c_tk); gc_mul carries dL/dP_k * P_k
                                gck += gc_step(t, m, k);
                                lamn[0] += gck * sc.dx[k];
                                lamn[1] += gck * sc.dy[k];
                                lamn[2] += gck * sc.dphi[k];
                            }
                        }
                        // observation columns (x, y, phi, u)_surr - (x, y, phi, u)_ego: MaskAtDone keeps the adjoint on obs_t
                        if (!dn && !p.env.surr_penalty) {
#pragma unroll
                            for (int i = 0; i < GOPS_MAX_SURR; ++i)
                                if (i < ns) {
                                    float* gp = G + m * ldx + 6 + 4 * P + 4 * i;
                                    lamn[0] -= gp[0]; lamn[1] -= gp[1]; lamn[2] -= gp[2]; lamn[3] -= gp[3];
                                    gp[0] = gp[1] = gp[2] = gp[3] = 0.f;
                                }
                        }
                    }
                }
                float g_steer, g_ax;
                veh_f_xu_bwd(VC, s, steer, w, lamn, lam, g_steer, g_ax);
                if constexpr (SURR) {
                    if (p.env.surr_penalty && m < nvalid) {
                        // surrcstr_penalty: both the appended observation (ego frame of the CURRENT state, NEXT vehicle) and the
                        // collision penalty in the reward depend on state_t, whose adjoint `lam` is now complete from the step
                        const GLOBAL_AS f32x4* sp = gptr(p.surr_table) + ((size_t)(b0 + m) * (p.H + 1) + t) * p.env.n_surr;
                        const f32x4 cur = sp[0], nxt = sp[p.env.n_surr];
                        const float sphi = e3[0], cphi = e3[1];
                        if (!dn) {
                            float* gp = G + m * ldx + 6 + 4 * P;
                            const float g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3];
                            const float dx = nxt[0] - s[0], dy = nxt[1] - s[1];
                            // x_tf = dx cos(phi) + dy sin(phi);  y_tf = -dx sin(phi) + dy cos(phi)
                            lam[0] += -g0 * cphi + g1 * sphi;
                            lam[1] += -g0 * sphi - g1 * cphi;
                            lam[2] += g0 * (-dx * sphi + dy * cphi) + g1 * (-dx * cphi - dy * sphi) - g2;
                            lam[3] -= g3;
                            gp[0] = gp[1] = gp[2] = gp[3] = 0.f;
                        }
                        SurrCstr sc;
                        surr_constraint<true>(p.env, s[0], s[1], sphi, cphi, &cur, sc);
                        float dpen;
                        (void)surr_penalty(sc.c[0], dpen);
                        // r = ... - pen(c(state_t)), masked like the reward.  info["constraint"] of this model is computed from
                        // DETACHED copies of the info dict (:129-139): the constraint sums carry no gradient here
                        const float gpen = (dn ? 0.f : g_r) * (-dpen);
                        lam[0] += gpen * sc.dx[0];
                        lam[1] += gpen * sc.dy[0];
                        lam[2] += gpen * sc.dphi[0];
                    }
                }
                const float g_rm = dn ? 0.f : g_r;
                if (m < nvalid) {
                    float xr[6];
                    if constexpr (STAGE) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) xr[i] = st_x[i * 16 + m];
                    } else {
#pragma unroll
                        for (int i = 0; i < 6; ++i) xr[i] = x_col(row0, m, i);
                    }
                    if constexpr (SURR) {
                        const float* rw = p.env.reward_w;
                        G[m * ldx + 0] += g_rm * (-2.f * rw[0] * xr[0]);
                        G[m * ldx + 1] += g_rm * (-2.f * rw[1] * xr[1]);
                        G[m * ldx + 2] += g_rm * (-2.f * rw[2] * xr[2]);
                        G[m * ldx + 3] += g_rm * (-2.f * rw[3] * xr[3]);
                        G[m * ldx + 5] += g_rm * (-2.f * rw[4] * xr[5]);
                        G[m * ldx + 4] += g_rm * (-2.f * rw[7] * xr[4]);
                        if (p.env.cstr_err) {   // errcstr: c_k = |obs[1 / 3]| - tol of THIS observation (unmasked sums / products)
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float e = xr[1 + 2 * k], c = fabsf(e) - p.env.err_tol[k];
                                float gck = gc_ext * 2.f * fmaxf(c, 0.f) + (c > 0.f ? gc_lin : 0.f);
                                if (c < 0.f) gck += gc_int * (-1.f / (-c + 1e-8f));
                                gck *= p.gpow[t];
                                float dlog;
                                (void)spil_phi(c, dlog);
                                gck += gc_mul[k] * dlog;
                                gck += gc_step(t, m, k);
                                G[m * ldx + 1 + 2 * k] += gck * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
                            }
                        }
                    } else {
                        G[m * ldx + 0] += g_rm * (-0.08f * xr[0]);
                        G[m * ldx + 1] += g_rm * (-0.08f * xr[1]);
                        G[m * ldx + 2] += g_rm * (-0.04f * xr[2]);
                        G[m * ldx + 3] += g_rm * (-0.04f * xr[3]);
                        G[m * ldx + 5] += g_rm * (-0.02f * xr[5]);
                    }
                }
                g_steer += g_rm * ((SURR ? -2.f * p.env.reward_w[5] : -0.02f) * steer);
                g_ax += g_rm * ((SURR ? -2.f * p.env.reward_w[6] : -0.02f) * ax);
                // (open_loop == 2, the raw-action rollouts of OptController, never reaches a SPLIT kernel)
                s_gy[m * 4 + 0] = (!SPLIT && p.open_loop == 2) ? g_steer : wrap_action_bwd(ha0, abar0, g_steer) * sc0 * (1.f - th0 * th0);
                s_gy[m * 4 + 1] = (!SPLIT && p.open_loop == 2) ? g_ax : wrap_action_bwd(ha1, abar1, g_ax) * sc1 * (1.f - th1 * th1);
                s_gy[m * 4 + 2] = 0.f;
                s_gy[m * 4 + 3] = 0.f;
            }
        }
        if constexpr (SPLIT || SSB) {
            if (tid < TB) {   // (the same 16 threads wrote s_gy above) -> this step's delta scale
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(s_gy + tid * 4);
                const float mx = row16_max(fmaxf(fmaxf(fabsf(g4[0]), fabsf(g4[1])), fmaxf(fabsf(g4[2]), fabsf(g4[3]))));
                if (tid == 0) {
                    split_delta_scale(mx, s_scale);
                    dy_run = fmaxf(dy_run, mx);
                }
            }
        }
        __syncthreads();
        DBG_TICK(1)
        if constexpr (SPLIT) {
            SS.run(p, s_gy, s_scale, dq2, dq1, G, ldx, tid, row0, nvalid, /*want_gx=*/t > 0 && ENV != GOPS_ENV_NONE, O, dbg, warm_up,
                   q.out_part != nullptr);
        } else if constexpr (SSB) {
            ss_net_backward(p.pol, p.sspt, s_wo, ldh, s_gy, s_scale, dq2, G, ldx, tid, p.st.h, p.st.z, p.st.d, p.st.dy, row0, nvalid,
                            /*want_gx=*/t > 0, O, warm_up, og, ssb_fuse_kind(ENV) && q.out_part != nullptr);
        } else
        if (!p.open_loop) {
            if constexpr (F16)
                mlp_backward_h(p.pol, s_wo, ldh, s_gy, reinterpret_cast<_Float16*>(da), reinterpret_cast<_Float16*>(db), ld16, G, ldx,
                               tid, p.st.h, p.st.z, p.st.d, p.st.dy, row0, nvalid, /*want_gx=*/t > 0 && ENV != GOPS_ENV_NONE, O, [] {});
            else
            if constexpr (EXT) {
                const bool keep = !(q.adj_first_only && t > 0);   // later steps act through the frozen policy copy
                mlp_backward<STAGE>(p.pol, WT0, WT1, s_wo, ldh, s_gy, da, db, ldh, G, ldx, tid, p.st.h, p.st.z,
                             keep ? p.st.d : nullptr, keep ? p.st.dy : nullptr, row0, nvalid,
                             /*want_gx=*/(t > 0 && ENV != GOPS_ENV_NONE) || q.adj_gobs != nullptr, O, dbg, warm_up, st_cur,
                             st_cur + TB * 256, ENV == GOPS_ENV_NONE ? q.ext_delta : nullptr, s_narrow);
            } else
            if constexpr (N64)
                mlp_backward_n64(hot64, s_wo, ldh, s_gy, da, db, ldh, G, ldx, tid, row0, nvalid, /*want_gx=*/t > 0 && ENV != GOPS_ENV_NONE, O, dbg, warm_up);
            else
            mlp_backward<STAGE>(p.pol, WT0, WT1, s_wo, ldh, s_gy, da, db, ldh, G, ldx, tid, p.st.h, p.st.z, p.st.d, p.st.dy, row0,
                         nvalid, /*want_gx=*/t > 0 && ENV != GOPS_ENV_NONE, O, dbg, warm_up, st_cur, st_cur + TB * 256,
                         ENV == GOPS_ENV_NONE ? q.ext_delta : nullptr, s_narrow);
        } else {   // open loop: the head adjoint IS the result; no policy input adjoint
            if (t > 0) stage_idp(t - 1);
            if (tid < nvalid) {
                GLOBAL_AS float* gp = gptr(q.g_head_pre) + ((size_t)(b0 + tid) * p.H + t) * A;
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a)
                    if (a < A) gp[a] = s_gy[tid * 4 + a];
            }
        }
        // retire this step's warm-up loads inside the same iteration: the compiler can then count the
        // memory operations issued since (exact vmcnt) instead of draining everything at the back-edge
#pragma unroll
        for (int q = 0; q < TOUCH_SLOTS; ++q) l2_sink ^= l2_pf[q];
        if constexpr (STAGE || IDPPARK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next step's staged data has landed
        __syncthreads();
        DBG_TICK(2)
    }
    if (l2_sink == 0x9e3779b9u && q.dbg != nullptr) gptr(q.dbg)[15] = l2_sink;   // keeps the warm-up loads alive
    if constexpr (EXT) {
        if (q.adj_gobs != nullptr) {   // (the loop's closing barrier made every G update visible)
            for (int idx = tid; idx < nvalid * O; idx += NTHREADS) {
                const int m = idx / O, i = idx - m * O;
                gptr(q.adj_gobs)[(size_t)(b0 + m) * O + i] = G[m * ldx + i];
            }
        }
    }
    if constexpr (SPLIT) {
        if (q.out_part != nullptr && tile + (int)gridDim.x >= ntiles) SS.store_out_grad(p, q.out_part, q.out_part_b, tid);   // after the last tile
    }
    if constexpr (SSB && ssb_fuse_kind(ENV)) {
        if (q.out_part != nullptr && tile + (int)gridDim.x >= ntiles)
            ss_store_out_grad(og, p.pol.dims[p.pol.nl - 1], p.pol.dims[p.pol.nl], q.out_part, q.out_part_b, tid);   // after the last tile
    }
    if constexpr ((SPLIT || SSB) && !F16) {
        // The weight-gradient GEMMs scale the hidden deltas by a power of two before they split them into half planes.  max|grad_v|
        // (gscale[0]) is a poor yardstick for that when the rollout amplifies adjoints over the horizon (H = 50, a half-trained
        // policy: deltas 10^3 .. 10^4 x grad_v, saturated blocks, the exact redo doubled the GEMM's time - cfg4, round 5); the
        // largest |delta_y| any step of the sweep saw is the right one: hidden deltas are within |W_o|, |W_j| row sums of it.
        if (tid == 0 && p.gscale != nullptr && dy_run < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(p.gscale) + 1, __float_as_uint(dy_run));
    }
    if constexpr (SPLIT || SSB) {
        // A delta beyond the half range of the plane images (the per-step scale leaves 2^12 of headroom above max|delta_y|) made part of
        // this tile's sweep non-finite, which act' = 0 can hide again: the tile poisons ONE element of its first hidden layer's delta
        // stash (its own: feature 0, row 0 of step 0 was written by thread 0), so that the first layer's weight gradient comes out
        // non-finite and the failure is loud
        unsigned o = og.ovf;
        og.ovf = 0;
        if constexpr (SPLIT) { o = SS.ovf; SS.ovf = 0; }
        if (split_overflow_any(o, red, tid) && tid == 0) {   // (red: the env adjoint's partials are consumed by now)
            if (p.st.d[1] != nullptr) gptr(p.st.d[1])[(size_t)tile * hH * TB * 256] = __builtin_nanf("");
            if (p.gscale != nullptr) atomicOr(reinterpret_cast<unsigned*>(p.gscale) + 3, 1u);   // the reduce poisons every gradient element of the call
        }
    }
    } while ((SPLIT || SSB) && MULTI && (tile += gridDim.x) < ntiles);   // (every step ends with a barrier: the next tile's set-up may overwrite G / s_ref)
    dbg.dump(q.dbg);
    if (q.ad_st != nullptr && blockIdx.x == 0 && threadIdx.x == 0) adam_snapshot(q.ad_st, q.ad_snap, q.ad_b1, q.ad_b2);   // (gops_rollout_backward_update)
}

// ref_points: reference-table points per trajectory (veh3dofconti), 30 (= 5 x 24 / 4) for the
// idpendulum sub-step parking area, else 0
size_t rollout_bwd_lds_bytes(int ldx, int ldh, int ref_points, bool f16, bool split, bool ssb = false) {
    size_t b = sizeof(float) * (size_t)bwd_lds_floats(ldx, ldh, ref_points, f16, split, ssb);
    if (split) b += sizeof(float) * 2 * (TB * ENV_STASH + TB * 8) +
                    (ldx - 4 > 128 ? 0 : (size_t)((ldx - 4) >> 4) * 8 * 1024);   // small staging halves + W_0's residual plane (streamed beyond 128 inputs)
    return b;
}

void rollout_variant(const RolloutParams& p, int sk[2], bool backward);
int split_grid_limit();   // rollout_fwd.hip: CUs of the device
int ssb_grid_limit() { return 2 * split_grid_limit(); }   // workgroups of the streamed-split sweep (= slabs of its fused output-layer gradient)

// The sweep of a streamed-split forward launch (p.ss) on the streamed-split sweep as well: same conditions, its LDS image at
// two workgroups per CU.  GOPS_SSB=0 keeps the fp32-MFMA sweep.
bool ssb_eligible(const RolloutParams& p) {
    if (!p.ss) return false;
    if (p.vflags & GOPS_VF_NO_STREAMED_SPLIT_BWD) return false;
    const int ref_pts = env_has_ref_table(p.env.kind) ? p.env.pre_horizon + 1 + p.H : (p.env.kind == GOPS_ENV_IDPENDULUM ? IDP_POINTS(false) : 0);
    return rollout_bwd_lds_bytes(p.ldx, p.ldh, ref_pts, false, false, true) + (env_in_lds(p.env.kind, true) ? 4 * ENV_LDS_FLOATS : 0) <= 80 * 1024;
}

#define LAUNCH_BWD(ENV, A, B)                                                                            \
    do {                                                                                                 \
        if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, A, B, true>, grid, block, lds, stream, dp, q);   \
        else launch_with_lds(rollout_bwd_kernel<ENV, A, B, false>, grid, block, lds, stream, dp, q);         \
    } while (0)
// the plain streamed fp32 sweep, or its obs -> 64 -> 64 -> act form (RolloutParams.narrow == 2: mlp_backward_n64)
#define LAUNCH_BWD_PLAIN(ENV)                                                                                                                       \
    do {                                                                                                                                            \
        if (p.narrow == 2 && q.ext_delta == nullptr) {   /* (gops_mlp_backward with a wide output layer: the generic head) */                       \
            if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, true, 1, false, false, false, false, false, true>, grid, block, lds, stream, dp, q);  \
            else launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, false, 1, false, false, false, false, false, true>, grid, block, lds, stream, dp, q);        \
        } else LAUNCH_BWD(ENV, 0, 0);                                                                                                               \
    } while (0)

#define LAUNCH_BWD_H(ENV)                                                                                       \
    do {                                                                                                        \
        if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, true, 1, true>, grid, block, lds, stream, dp, q); \
        else launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, false, 1, true>, grid, block, lds, stream, dp, q);       \
    } while (0)

#define LAUNCH_BWD2(ENV, A, B, PT)                                                                          \
    do {                                                                                                    \
        if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, A, B, true, PT>, grid, block, lds, stream, dp, q);  \
        else launch_with_lds(rollout_bwd_kernel<ENV, A, B, false, PT>, grid, block, lds, stream, dp, q);        \
    } while (0)

hipError_t launch_rollout_bwd_h64(const RolloutParams& p, const RolloutParams* dp, const BwdPatch& q, hipStream_t stream);   // rollout_h64.hip
hipError_t launch_rollout_bwd(const RolloutParams& p, const RolloutParams* dp, const BwdPatch& q, hipStream_t stream) {
#ifdef GOPS_ONLY_NARROW   // the same for the plain streamed fp32 kernel of pyth_idpendulum (cfg1, the example scripts' shapes): EXTRA=-DGOPS_ONLY_NARROW
#if GOPS_ONLY_NARROW == 2
    launch_with_lds(rollout_bwd_kernel<GOPS_ENV_IDPENDULUM, 0, 0, false, 1, false, false, false, false, false, true>, dim3((p.B + TB - 1) / TB), dim3(NTHREADS), 4 * ((size_t)p.narrow_off_bwd + p.narrow_floats), stream, dp, q);
#else
    launch_with_lds(rollout_bwd_kernel<GOPS_ENV_IDPENDULUM, 0, 0, false>, dim3((p.B + TB - 1) / TB), dim3(NTHREADS), 4 * ((size_t)p.narrow_off_bwd + p.narrow_floats), stream, dp, q);
#endif
    return hipGetLastError();
#elif defined(GOPS_ONLY_TARGET)   // register / spill studies (EXTRA=-DGOPS_ONLY_TARGET tools/kernel_regs.sh rollout_bwd.hip): ONE instantiation, seconds to compile
    launch_with_lds(rollout_bwd_kernel<GOPS_ENV_VEH3DOFCONTI, 8, 8, false, 2, false, false, true>, dim3(1), dim3(NTHREADS), 0, stream, dp, q);
    return hipGetLastError();
#else
    if (p.h64) return launch_rollout_bwd_h64(p, dp, q, stream);   // half precision, 64-trajectory tiles
    const dim3 grid((p.B + TB - 1) / TB), block(NTHREADS);
    const int ref_pts = env_has_ref_table(p.env.kind) ? p.env.pre_horizon + 1 + p.H : (p.env.kind == GOPS_ENV_IDPENDULUM ? IDP_POINTS(false) : 0);
    size_t lds = rollout_bwd_lds_bytes(p.ldx, p.ldh, ref_pts, p.f16 != 0, false);
    if (p.narrow) lds = 4 * ((size_t)p.narrow_off_bwd + p.narrow_floats);   // (api.hip: only ever set for the plain streamed fp32 sweeps, EXT included)
    if (p.ext) {   // adjoint I/O / ActionRepeat: streamed fp32 kernels of the obs == state kinds
        if (p.f16) return hipErrorInvalidValue;
#define LAUNCH_BWD_EXT(ENV)                                                                                             \
    do {                                                                                                                \
        if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, true, 1, false, true>, grid, block, lds, stream, dp, q);  \
        else launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, false, 1, false, true>, grid, block, lds, stream, dp, q);        \
    } while (0)
        switch (p.env.kind) {
            case GOPS_ENV_NONE:
                if (p.tail) return hipErrorInvalidValue;
                launch_with_lds(rollout_bwd_kernel<GOPS_ENV_NONE, 0, 0, false, 1, false, true>, grid, block, lds, stream, dp, q);
                break;
            case GOPS_ENV_LQ: LAUNCH_BWD_EXT(GOPS_ENV_LQ); break;
            case GOPS_ENV_IDPENDULUM: LAUNCH_BWD_EXT(GOPS_ENV_IDPENDULUM); break;
            case GOPS_ENV_CARTPOLE: LAUNCH_BWD_EXT(GOPS_ENV_CARTPOLE); break;
            case GOPS_ENV_PENDULUM: LAUNCH_BWD_EXT(GOPS_ENV_PENDULUM); break;
            case GOPS_ENV_MOBILEROBOT: LAUNCH_BWD_EXT(GOPS_ENV_MOBILEROBOT); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (p.ssb && !p.ext && !p.open_loop && q.ext_delta == nullptr) {   // streamed-split sweep (gops_mlp_backward's hidden-stack deltas: the fp32 sweep)
        const size_t lds_ss = rollout_bwd_lds_bytes(p.ldx, p.ldh, ref_pts, false, false, true) + (env_in_lds(p.env.kind, true) ? 4 * ENV_LDS_FLOATS : 0);
        const dim3 grid_ss(std::min<int>((p.B + TB - 1) / TB, ssb_grid_limit()));   // two workgroups per CU walk the tiles grid-stride
#define LAUNCH_BWD_SS(ENV)                                                                                                                  \
    do {                                                                                                                                    \
        constexpr bool MT = ssb_fuse_kind(ENV);   /* grid-stride walk + fused output-layer gradient */                                        \
        const dim3 g = MT ? grid_ss : grid;                                                                                                   \
        if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, true, 1, false, false, false, MT, true>, g, block, lds_ss, stream, dp, q);   \
        else launch_with_lds(rollout_bwd_kernel<ENV, 0, 0, false, 1, false, false, false, MT, true>, g, block, lds_ss, stream, dp, q);         \
    } while (0)
        switch (p.env.kind) {
            case GOPS_ENV_NONE: LAUNCH_BWD_SS(GOPS_ENV_NONE); break;
            case GOPS_ENV_LQ: LAUNCH_BWD_SS(GOPS_ENV_LQ); break;
            case GOPS_ENV_IDPENDULUM: LAUNCH_BWD_SS(GOPS_ENV_IDPENDULUM); break;
            case GOPS_ENV_VEH3DOFCONTI: LAUNCH_BWD_SS(GOPS_ENV_VEH3DOFCONTI); break;
            case GOPS_ENV_VEH3DOF_SURR: LAUNCH_BWD_SS(GOPS_ENV_VEH3DOF_SURR); break;
            case GOPS_ENV_CARTPOLE: LAUNCH_BWD_SS(GOPS_ENV_CARTPOLE); break;
            case GOPS_ENV_PENDULUM: LAUNCH_BWD_SS(GOPS_ENV_PENDULUM); break;
            case GOPS_ENV_VEH2DOF: LAUNCH_BWD_SS(GOPS_ENV_VEH2DOF); break;
            case GOPS_ENV_MOBILEROBOT: LAUNCH_BWD_SS(GOPS_ENV_MOBILEROBOT); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (p.sp.on) {   // plane-split stationary sweep: PT0 = n-tiles of g_x per wave
        lds = rollout_bwd_lds_bytes(p.ldx, p.ldh, p.env.kind == GOPS_ENV_IDPENDULUM ? IDP_POINTS(true) : ref_pts, false, true);
#define LAUNCH_BWD_SPLIT(ENV, PT)                                                                                                  \
    do {                                                                                                                          \
        if (multi) {                                                                                                              \
            if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, 8, 8, true, PT, false, false, true, true>, grid, block, lds, stream, dp, q);    \
            else launch_with_lds(rollout_bwd_kernel<ENV, 8, 8, false, PT, false, false, true, true>, grid, block, lds, stream, dp, q);          \
        } else if (p.tail) launch_with_lds(rollout_bwd_kernel<ENV, 8, 8, true, PT, false, false, true>, grid, block, lds, stream, dp, q);    \
        else launch_with_lds(rollout_bwd_kernel<ENV, 8, 8, false, PT, false, false, true>, grid, block, lds, stream, dp, q);          \
    } while (0)
        const int pt = (p.pol.kp[0] + 63) >> 6;
        const dim3 grid(std::min<int>((p.B + TB - 1) / TB, split_grid_limit()));   // one workgroup per CU, grid-stride over the tiles
        const bool multi = (p.B + TB - 1) / TB > split_grid_limit();
        if (p.env.kind == GOPS_ENV_LQ && pt == 1) LAUNCH_BWD_SPLIT(GOPS_ENV_LQ, 1);
        else if (p.env.kind == GOPS_ENV_IDPENDULUM && pt == 1) LAUNCH_BWD_SPLIT(GOPS_ENV_IDPENDULUM, 1);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && pt == 1) LAUNCH_BWD_SPLIT(GOPS_ENV_VEH3DOFCONTI, 1);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && pt == 2) LAUNCH_BWD_SPLIT(GOPS_ENV_VEH3DOFCONTI, 2);
#define LAUNCH_BWD_SPLIT_NOTAIL(ENV, PT)                                                                                          \
    do {                                                                                                                          \
        if (multi) launch_with_lds(rollout_bwd_kernel<ENV, 8, 8, false, PT, false, false, true, true>, grid, block, lds, stream, dp, q);  \
        else launch_with_lds(rollout_bwd_kernel<ENV, 8, 8, false, PT, false, false, true>, grid, block, lds, stream, dp, q);       \
    } while (0)
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && pt == 3 && !p.tail) LAUNCH_BWD_SPLIT_NOTAIL(GOPS_ENV_VEH3DOFCONTI, 3);
        else if (p.env.kind == GOPS_ENV_VEH3DOFCONTI && pt == 4 && !p.tail) LAUNCH_BWD_SPLIT_NOTAIL(GOPS_ENV_VEH3DOFCONTI, 4);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    int sk[2];
    rollout_variant(p, sk, true);
    if (sk[1] > 0) lds += sizeof(float) * 2 * (2 * TB * 256 + TB * ENV_STASH + TB * 8);   // two staging halves
    const int key = sk[0] * 100 + sk[1];
    if (p.f16) {
        switch (p.env.kind) {
            case GOPS_ENV_NONE: LAUNCH_BWD_H(GOPS_ENV_NONE); break;
            case GOPS_ENV_LQ: LAUNCH_BWD_H(GOPS_ENV_LQ); break;
            case GOPS_ENV_IDPENDULUM: LAUNCH_BWD_H(GOPS_ENV_IDPENDULUM); break;
            case GOPS_ENV_VEH3DOFCONTI: LAUNCH_BWD_H(GOPS_ENV_VEH3DOFCONTI); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (p.env.kind) {
        case GOPS_ENV_NONE: LAUNCH_BWD_PLAIN(GOPS_ENV_NONE); break;
        case GOPS_ENV_LQ:
            if (key == 1616) LAUNCH_BWD(GOPS_ENV_LQ, 16, 16); else LAUNCH_BWD_PLAIN(GOPS_ENV_LQ);
            break;
        case GOPS_ENV_IDPENDULUM:
            if (key == 1616) LAUNCH_BWD(GOPS_ENV_IDPENDULUM, 16, 16); else LAUNCH_BWD_PLAIN(GOPS_ENV_IDPENDULUM);
            break;
        case GOPS_ENV_VEH3DOFCONTI:
            if (key == 1216) LAUNCH_BWD2(GOPS_ENV_VEH3DOFCONTI, 12, 16, 2);
            else if (key == 16) LAUNCH_BWD(GOPS_ENV_VEH3DOFCONTI, 0, 16);
            else LAUNCH_BWD_PLAIN(GOPS_ENV_VEH3DOFCONTI);
            break;
        case GOPS_ENV_VEH3DOF_SURR: LAUNCH_BWD_PLAIN(GOPS_ENV_VEH3DOF_SURR); break;
        case GOPS_ENV_CARTPOLE: LAUNCH_BWD_PLAIN(GOPS_ENV_CARTPOLE); break;
        case GOPS_ENV_PENDULUM: LAUNCH_BWD_PLAIN(GOPS_ENV_PENDULUM); break;
        case GOPS_ENV_VEH2DOF: LAUNCH_BWD_PLAIN(GOPS_ENV_VEH2DOF); break;
        case GOPS_ENV_MOBILEROBOT: LAUNCH_BWD_PLAIN(GOPS_ENV_MOBILEROBOT); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
}
