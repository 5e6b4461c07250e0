// Half-precision rollout kernels with 64-TRAJECTORY tiles (GOPS_DTYPE_F16, BASELINE.json configs[4] "fp16 MFMA MLP path";
// round 4).  Same arithmetic, packings and stash contents as the 16-row half kernels (rollout_f16.h: transposed MFMAs - M =
// output feature, N = trajectory, K = input feature -, fp32 accumulation, half activations / deltas / stash, fp32 env model,
// head and adjoints), other work decomposition:
//   * a workgroup (4 waves) owns 64 trajectories; wave w owns the 64-feature quad w of every 256-wide layer for ALL 64 rows:
//     each weight fragment it pulls from L2 feeds FOUR MFMAs (one per 16-row group) instead of one - the weight stream per
//     trajectory is a quarter of the 16-row kernels' (there: 144 KB of fragments per 16.5 KB of stash traffic and tile-step);
//   * every per-step scalar / address / barrier cost is paid once per 64 rows, and the env phase runs on all 64 lanes of a
//     wave instead of 16;
//   * ONE hidden tile in LDS, rewritten in place behind a barrier (the whole GEMM result sits in 64 accumulator registers),
//     so two workgroups share a CU.
// Stash rows are STEP-major in 64-row tiles: row (t * tiles + tile) * 64 + m - at one step the resident workgroups read / write one
// contiguous window (tile-major: 3.5 % slower, address windows 320 KB apart).  The weight-gradient GEMMs contract over all rows
// and do not care about their order.
// Scope: env kinds GOPS_ENV_NONE (value / MLP batches) and GOPS_ENV_LQ, closed loop, every hidden layer 256 wide, at most 64
// padded inputs; everything else stays on the 16-row kernels (api.hip: h64_eligible).
#include "common.h"
#include "env_models.h"
#include "rollout_f16.h"

#define TB64 64
// stash stores are non-temporal (measured: plain stores cost the sweep 9 %)
#define H64_STORE(v, ptr) __builtin_nontemporal_store(v, ptr)
// (the env description is copied to LDS once per workgroup and the env phases read it there: common.h ENV_LDS_FLOATS)
#define H64_LD 264   // halfs per row of the hidden tile: 256 + 8 (16-byte row skew, conflict-free ds_read_b128)

// acc[jt][rg] (n-tile jt of quad q, 16-row group rg) += W_quad * act^T over kch chunks of 32 inputs
template <int RG>   // 16-row groups of the tile (4: 64 trajectories per workgroup, 2: 32)
__device__ __forceinline__ void gemm_quad_h64(const _Float16* act, int ld, int kch, const f16x8* Wp, int q, int lane, f32x4 (&acc)[4][RG]) {
    constexpr int PF = 2, NJ = 4;   // (measured at cfg5: ring depths 3 and 4 change nothing)
    const GLOBAL_AS f16x8* wb = gptr(Wp) + (size_t)q * 4 * kch * 64 + lane;
    const _Float16* brow = act + (lane & 15) * ld + 8 * (lane >> 4);
    f16x8 ring[PF][NJ];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int cd = d < kch ? d : kch - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) ring[d][j] = wb[((size_t)j * kch + cd) * 64];
    }
    for (int c0 = 0; c0 < kch; c0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int c = c0 + d;
            if (c < kch) {
                f16x8 b[RG], a[NJ];
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) b[rg] = ld8h(brow + 16 * rg * ld + 32 * c);
#pragma unroll
                for (int j = 0; j < NJ; ++j) a[j] = ring[d][j];
                const int cn = (c + PF < kch) ? c + PF : kch - 1;
#pragma unroll
                for (int j = 0; j < NJ; ++j) ring[d][j] = wb[((size_t)j * kch + cn) * 64];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg) acc[j][rg] = MFMA_F16(a[j], b[rg], acc[j][rg]);
            }
        }
    }
}

// fp32 tile xs [64][ldx] -> half tile x16 [64][ld16] (kp32 columns, zero padded) and the stash rows g16[(row0 + m) * kp32 ..]
__device__ __forceinline__ void convert_x_h64(const float* xs, int ldx, int kp, int kp32, _Float16* x16, int ld16, _Float16* g16, size_t row0, int tid) {
    const int upr = kp32 >> 3;
    for (int idx = tid; idx < TB64 * upr; idx += NTHREADS) {
        const int m = idx / upr, c = (idx - m * upr) << 3;
        f16x8 v = zero8h();
        if (c < kp) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xs + m * ldx + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(xs + m * ldx + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (_Float16)a[e]; v[4 + e] = (_Float16)b[e]; }
        }
        *reinterpret_cast<f16x8*>(x16 + m * ld16 + c) = v;
        if (g16 != nullptr) H64_STORE(v, gptr(reinterpret_cast<f16x8*>(g16 + (row0 + m) * kp32 + c)));
    }
}

// Hidden stack on the half tile x16 -> hbuf (the LAST hidden activation, [64][H64_LD]); activations (and act'(z) for GELU)
// of rows < stash_rows go to stash_h / stash_g rows row0 + m.  Ends with a barrier.
__device__ __forceinline__ void mlp_hidden_forward_h64(const MlpDev& M, const _Float16* x16, int ldx16, _Float16* hbuf, int tid, const float* s_bias,
                                                       int ldb, float* const* stash_h, float* const* stash_g, size_t row0, int stash_rows, DbgClock& dbg) {
    const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
    const int L = M.nl - 1;
    const int f0 = 64 * wave + 16 * g;
    for (int j = 0; j < L; ++j) {
        f32x4 acc[4][4] = {};
        gemm_quad_h64<4>(j == 0 ? x16 : hbuf, j == 0 ? ldx16 : H64_LD, M.kp32[j] >> 5, M.wph[j], wave, lane, acc);
        DBG_TICK(2 + 3 * (j > 0))
        if (j > 0) __syncthreads();   // every wave has read the tile it is about to overwrite
        f32x4 bv[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bv[jj] = *reinterpret_cast<const f32x4*>(s_bias + j * ldb + f0 + 4 * jj);
        const bool gelu = M.act == GOPS_ACT_GELU;
        // (gradient-free launches - INFADP's policy-evaluation rollout - keep no stash: the GELU epilogue then skips gelu'(z) and its
        // conversion, ~4 of its ~26 issue slots per element; the epilogues are what this kernel is bound by)
        auto epilogue = [&]<int ACT, bool WANT_DH>() {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = 16 * rg + m;
                f16x8 o[2], gd[2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 4 * jj + r;
                        const float z = acc[jj][rg][r] + bv[jj][r];
                        float h, dh = 0.f;
#ifdef H64_KO_MATH   // (knock-out build: what do the epilogue's transcendentals cost?)
                        h = z; dh = z;
#else
                        if (ACT == GOPS_ACT_GELU) {
                            gelu_pair_h(z, h, dh);
                            if (!WANT_DH) dh = 0.f;   // (dead: the compiler drops its two instructions and the conversion below)
                        } else h = act_fwd_t<ACT>(z);
#endif
                        o[e >> 3][e & 7] = (_Float16)h;
                        if (WANT_DH) gd[e >> 3][e & 7] = (_Float16)dh;
                    }
                *reinterpret_cast<f16x8*>(hbuf + row * H64_LD + f0) = o[0];
                *reinterpret_cast<f16x8*>(hbuf + row * H64_LD + f0 + 8) = o[1];
#ifdef H64_KO_STORE   // (knock-out build: what do the stash stores cost?)
                if (false) {
#else
                if (stash_h != nullptr && row < stash_rows) {
#endif
                    _Float16* hrow = reinterpret_cast<_Float16*>(stash_h[j + 1]) + (row0 + row) * 256;
                    H64_STORE(o[0], gptr(reinterpret_cast<f16x8*>(hrow + f0)));
                    H64_STORE(o[1], gptr(reinterpret_cast<f16x8*>(hrow + f0 + 8)));
                    if (WANT_DH && ACT == GOPS_ACT_GELU && gelu && stash_g != nullptr) {
                        _Float16* grow = reinterpret_cast<_Float16*>(stash_g[j + 1]) + (row0 + row) * 256;
                        H64_STORE(gd[0], gptr(reinterpret_cast<f16x8*>(grow + f0)));
                        H64_STORE(gd[1], gptr(reinterpret_cast<f16x8*>(grow + f0 + 8)));
                    }
                }
            }
        };
        act_dispatch(M.act, [&]<int ACT>() {
            if (ACT == GOPS_ACT_GELU && stash_h == nullptr) epilogue.template operator()<ACT, false>();
            else epilogue.template operator()<ACT, true>();
        });
        DBG_TICK(3 + 3 * (j > 0))
        __syncthreads();
        DBG_TICK(4 + 3 * (j > 0))
    }
}

size_t rollout_fwd_h64_lds_bytes(int ldx, int ldh) {
    const int ldx16 = (((ldx - 4) + 31) & ~31) + 8;
    return sizeof(float) * (size_t)(TB64 * ldx + TB64 * (4 + 4 + 1) + 16 + 32 + 4 * ldh + (GOPS_MAX_LAYERS - 1) * ldh + ENV_LDS_FLOATS) +
           sizeof(_Float16) * (size_t)(TB64 * ldx16 + TB64 * H64_LD);
}

template <int ENV, bool TAIL>
__global__ __launch_bounds__(NTHREADS, 2) void rollout_fwd_h64_kernel(const RolloutParams* __restrict__ pp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutParams& p = *pp;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, b0 = tile * TB64, nvalid = min(TB64, p.B - b0);
    const int O = p.env.obs_dim, A = p.env.act_dim;
    const int ldx = p.ldx, ldh = p.ldh;
    float* xs = smem;                       // [64][ldx] current observation (+ time column)
    float* s_act = xs + TB64 * ldx;         // [64][4] wrapped action
    float* s_th = s_act + TB64 * 4;         // [64][4] tanh(head) (ENV_NONE: raw head output)
    float* s_done = s_th + TB64 * 4;        // [64]
    float* s_bo = s_done + TB64;            // [16] head bias
    float* s_ac = s_bo + 16;                // [4][8] per-action constants
    float* s_wo = s_ac + 32;                // 4 ldh floats: the head weights as half-plane MFMA fragments (below)
    f16x8* s_woh = reinterpret_cast<f16x8*>(s_wo);
    float* s_bias = s_wo + 4 * ldh;         // [GOPS_MAX_LAYERS - 1][ldh]
    float* s_env = s_bias + (GOPS_MAX_LAYERS - 1) * ldh;   // GopsEnv copy
    const GopsEnv& env = *reinterpret_cast<const GopsEnv*>(s_env);
    for (int idx = tid; idx < (int)(sizeof(GopsEnv) / 4); idx += NTHREADS) s_env[idx] = gptr(reinterpret_cast<const float*>(&p.env))[idx];
    _Float16* x16 = reinterpret_cast<_Float16*>(s_env + ENV_LDS_FLOATS);
    const int ldx16 = (((p.ldx - 4) + 31) & ~31) + 8;
    _Float16* hbuf = x16 + TB64 * ldx16;    // [64][H64_LD]
    {
        const int Lh = p.pol.nl - 1, K = p.pol.dims[Lh], Ao = p.pol.dims[p.pol.nl];
        // head weights (fp32) as TWO half planes hi = half(w), lo = half(w - hi) (22 significant bits) in MFMA A-fragment order:
        // s_woh[(plane * kch + c) * 16 + 4 g + a] = W_o[a][32 c + 8 g .. + 7], a < 4 (fragment rows 4 .. 15 are zero: not stored)
        for (int idx = tid; idx < 2 * (K >> 5) * 16; idx += NTHREADS) {
            const int a = idx & 3, g = (idx >> 2) & 3, c = (idx >> 4) % (K >> 5), plane = idx / (16 * (K >> 5));
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = (a < Ao) ? gptr(p.pol.w[Lh])[a * K + 32 * c + 8 * g + e] : 0.f;
                const _Float16 hi = (_Float16)w;
                v[e] = plane ? (_Float16)(w - (float)hi) : hi;
            }
            s_woh[idx] = v;
        }
        if (tid < GOPS_MAX_ACT) s_bo[tid] = (tid < Ao) ? gptr(p.pol.b[Lh])[tid] : 0.f;
        stage_act_const(p.env, s_ac, tid);
        for (int j = 0; j < Lh; ++j)
            for (int n = tid; n < p.pol.dims[j + 1]; n += NTHREADS) s_bias[j * ldh + n] = gptr(p.pol.b[j])[n];
    }
    for (int idx = tid; idx < TB64 * ldx; idx += NTHREADS) {
        const int m = idx / ldx, c = idx - m * ldx;
        xs[idx] = (c < O && m < nvalid) ? gptr(p.in.obs)[(size_t)(b0 + m) * O + c] : 0.f;
    }
    if (tid < TB64) s_done[tid] = (tid < nvalid && p.in.done != nullptr && !p.env.no_mask_at_done && gptr(p.in.done)[b0 + tid] != 0.f) ? 1.f : 0.f;
    float v_acc = 0.f;
    __syncthreads();
    DbgClock dbg;   // phase counters of thread 0 (GOPS_DBG_BUILD + GOPS_DBG_TIMING=1, tools/dbg_run.py)
    dbg.init((p.dbg != nullptr) && blockIdx.x == 0 && tid == 0);
    for (int t = 0; t < p.H; ++t) {
        if (p.fh && tid < TB64) xs[tid * ldx + O] = (float)(t + 1);
        __syncthreads();
        DBG_TICK(0)
        const size_t row0 = ((size_t)t * gridDim.x + tile) * TB64;   // step-major stash tiles
        convert_x_h64(xs, ldx, p.pol.kp[0], p.pol.kp32[0], x16, ldx16, p.need_grad ? reinterpret_cast<_Float16*>(p.st.x) : nullptr, row0, tid);
        if (p.need_grad && tid < 2 * TB64)
            *gptr(reinterpret_cast<f32x4*>(p.st.xf + (row0 + (tid >> 1)) * 8 + 4 * (tid & 1))) =
                *reinterpret_cast<const f32x4*>(xs + (tid >> 1) * ldx + 4 * (tid & 1));
        __syncthreads();
        DBG_TICK(1)
        mlp_hidden_forward_h64(p.pol, x16, ldx16, hbuf, tid, s_bias, ldh, p.need_grad ? p.st.h : nullptr, p.need_grad ? p.st.z : nullptr, row0, TB64, dbg);
        {   // head on the matrix core: y[a][row] = sum_k W_o[a][k] h[row][k] as (hi + lo) planes of W_o (fragment rows = actions, zero
            // padded to 16) times the half tile; wave w takes rows 16 w .. + 15, lanes 0 .. 15 receive the (<= 4) outputs of
            // their trajectory; then squash + wrapper chain.  (The VALU form read W_o from LDS once per 16-row pass: 590 KB of LDS
            // reads per tile-step, 6.4 k of the step's 32 k cycles.)
            const int lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4, kch = p.pol.dims[p.pol.nl - 1] >> 5;
            const _Float16* brow = hbuf + (16 * wave + n) * H64_LD + 8 * g;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < kch; ++c) {
                const f16x8 b = ld8h(brow + 32 * c);
                f16x8 ahi = zero8h(), alo = zero8h();
                if (n < 4) { ahi = s_woh[c * 16 + 4 * g + n]; alo = s_woh[(kch + c) * 16 + 4 * g + n]; }
                acc = MFMA_F16(ahi, b, acc);
                acc = MFMA_F16(alo, b, acc);
            }
            if (lane < 16) {
                const int row = 16 * wave + lane;
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a)
                    if (a < A) {
                        const float ya = acc[a] + s_bo[a];
                        if (ENV == GOPS_ENV_NONE) {
                            s_th[row * 4 + a] = ya;
                        } else {
                            const ActC c = act_const(s_ac, a);
                            const float th = fast_tanh(ya);
                            s_th[row * 4 + a] = th;
                            s_act[row * 4 + a] = wrap_action(c, c.sc * th + c.of);
                        }
                    }
            }
        }
        DBG_TICK(8)
        __syncthreads();
        DBG_TICK(9)
        float r = 0.f;
        if (tid < TB64) {
            const int m = tid;
            if (p.need_grad) {   // env stash row: tanh outputs, done_t (the LQ state IS the observation: X stash)
                GLOBAL_AS f32x4* er = gptr(reinterpret_cast<f32x4*>(p.st.env + (row0 + m) * ENV_STASH));
                const f32x4 e0 = {s_th[m * 4 + 0], s_th[m * 4 + 1], s_th[m * 4 + 2], s_th[m * 4 + 3]};
                const f32x4 e1 = {s_done[m], 0.f, 0.f, 0.f};
                er[0] = e0;
                er[1] = e1;
            }
            if (ENV == GOPS_ENV_NONE) {
                r = s_th[m * 4];
            } else {   // GOPS_ENV_LQ (the arithmetic of rollout_fwd_kernel's LQ block); NS / NA: compile-time loop bounds
                auto lq_step = [&]<int NS, int NA>() {
                    constexpr bool EXACT = NS < GOPS_MAX_LQ_STATE;   // the dimensions ARE (NS, NA): no run-time guards
                    float x[GOPS_MAX_LQ_STATE], xn[GOPS_MAX_LQ_STATE], u[GOPS_MAX_ACT];
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { x[i] = (i < NS && (EXACT || i < O)) ? obs_unscale(env, i, xs[m * ldx + i]) : 0.f; xn[i] = 0.f; }
#pragma unroll
                    for (int j = 0; j < GOPS_MAX_ACT; ++j) u[j] = (j < NA && (EXACT || j < A)) ? s_act[m * 4 + j] : 0.f;
                    const bool frozen = s_done[m] != 0.f;
                    lq_forward<NS, NA>(env, x, u, xn, r);
                    if (!frozen || env.clip_obs || env.scale_obs) {
#pragma unroll
                        for (int i = 0; i < NS; ++i)
                            if (EXACT || i < O) {
                                const float v = obs_rescale(env, i, sel_reg(frozen, x[i], xn[i]));
                                xs[m * ldx + i] = env.clip_obs ? clampf(v, env.obs_low[i], env.obs_high[i]) : v;
                            }
                    }
                };
                if (O == 4 && A == 2) lq_step.template operator()<4, 2>();   // BASELINE configs[4] (lq s4a2)
                else lq_step.template operator()<GOPS_MAX_LQ_STATE, GOPS_MAX_ACT>();
            }
            const float d = s_done[m];
            float rr = (d != 0.f) ? 0.f : r;
            if (ENV != GOPS_ENV_NONE && env.shaping) rr = (rr + env.reward_shift) * env.reward_scale;
            v_acc += rr * p.gpow[t];
            if (p.out.rewards != nullptr && m < nvalid) gptr(p.out.rewards)[(size_t)t * p.B + b0 + m] = rr;
            // (pyth_lq never terminates: done_m == false; the done flags handed in stay as they are)
        }
        DBG_TICK(10)
    }
    dbg.dump(p.dbg);
    __syncthreads();
    if (TAIL) {   // v += (~done_H) gamma^H V_target(obs_H)
        if (p.fh && tid < TB64) xs[tid * ldx + O] = 0.f;
        for (int j = 0; j < p.val.nl - 1; ++j)
            for (int n = tid; n < p.val.dims[j + 1]; n += NTHREADS) s_bias[j * ldh + n] = gptr(p.val.b[j])[n];
        __syncthreads();
        convert_x_h64(xs, ldx, p.val.kp[0], p.val.kp32[0], x16, ldx16, nullptr, 0, tid);
        __syncthreads();
        mlp_hidden_forward_h64(p.val, x16, ldx16, hbuf, tid, s_bias, ldh, p.need_grad ? p.st.tail_h : nullptr, p.need_grad ? p.st.tail_z : nullptr,
                               (size_t)b0, nvalid, dbg);
        const int Lv = p.val.nl - 1;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            float y[GOPS_MAX_ACT];
            mlp_head_h<false>(gptr(p.val.w[Lv]), p.val.dims[Lv], gptr(p.val.b[Lv]), p.val.dims[Lv], 1, hbuf + 16 * pass * H64_LD, H64_LD, tid, y);
            if ((tid & 15) == 0) s_th[(16 * pass + (tid >> 4)) * 4] = y[0];
        }
        __syncthreads();
        if (tid < TB64) v_acc += ((p.tail_unmasked ? 1.f : 1.f - s_done[tid]) * p.gpow[p.H]) * s_th[tid * 4];
    }
    if (tid < nvalid) {
        gptr(p.out.v_pi)[b0 + tid] = v_acc;
        if (p.out.final_done != nullptr) gptr(p.out.final_done)[b0 + tid] = s_done[tid];
        if (p.need_grad && p.st.tail_done != nullptr) gptr(p.st.tail_done)[b0 + tid] = s_done[tid];
    }
    if (p.out.final_obs != nullptr) {
        for (int idx = tid; idx < TB64 * O; idx += NTHREADS) {
            const int m = idx / O, c = idx - m * O;
            if (m < nvalid) gptr(p.out.final_obs)[(size_t)(b0 + m) * O + c] = xs[m * ldx + c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward sweep, 64-row tiles.  delta_y (fp32, in the launch's scaled units) in s_gy[64][4] -> hidden deltas (half; stash
// st_d[j] when non-null) and, if want_gx, G[row][n] += (delta_1 W_0)[row][n] for n < ncols.  ONE delta tile `dbuf`, rewritten
// in place behind a barrier.  Ends without a barrier after the g_x update (the caller's end-of-step barrier follows).
template <int RG, class WP, class Hook>
__device__ __forceinline__ void mlp_backward_h64(const MlpDev& M, WP Wo, int ldw, const float* s_gy, _Float16* dbuf, float* G, int ldg, int tid,
                                                 float* const* st_h, float* const* st_z, float* const* st_d, float* stash_dy, size_t row0,
                                                 int nvalid, bool want_gx, int ncols, DbgClock& dbg, Hook&& after_head, bool skip_d1 = false) {
    const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
    const int L = M.nl - 1, A = M.dims[M.nl];
    const bool gelu = (M.act == GOPS_ACT_GELU);
    {   // head: delta_L[row][k] = (sum_a gy[row][a] Wo[a][k]) * act'_L[row][k]; thread (hm, hp) walks 4 row groups x 2 column blocks
        const int hm = tid >> 4, hp = tid & 15;
        // all act' operands of the phase are requested before the first delta store (the compiler does not move a load
        // across a store it cannot prove disjoint: fetched inside the loop, every pass waited for its own HBM round trip)
        f16x8 hv[2][RG];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pass = 0; pass < RG; ++pass) {
                const int row = 16 * pass + hm;
                hv[kb][pass] = zero8h();
                if (row < nvalid)
                    hv[kb][pass] = ld8h(gptr(reinterpret_cast<const _Float16*>(gelu ? st_z[L] : st_h[L]) + (row0 + row) * 256 + 8 * hp + 128 * kb));
            }
        act_dispatch(M.act, [&]<int ACT>() {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int k = 8 * hp + 128 * kb;
                f32x4 w0[GOPS_MAX_ACT], w1[GOPS_MAX_ACT];
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    w0[a] = (a < A) ? ld4(Wo + a * ldw + k) : z;
                    w1[a] = (a < A) ? ld4(Wo + a * ldw + k + 4) : z;
                }
#pragma unroll
                for (int pass = 0; pass < RG; ++pass) {
                    const int row = 16 * pass + hm;
                    const bool ok = row < nvalid;
                    const f32x4 gy = *reinterpret_cast<const f32x4*>(s_gy + row * 4);
                    f16x8 dv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float acc = 0.f;
#pragma unroll
                        for (int a = 0; a < GOPS_MAX_ACT; ++a) acc += gy[a] * (e < 4 ? w0[a][e] : w1[a][e - 4]);   // (rows a >= A: zero weights)
                        const float hf = (float)hv[kb][pass][e];
                        const float d = (ACT == GOPS_ACT_GELU) ? hf : act_bwd_t<ACT>(hf, hf);
                        dv[e] = sat_h(ok ? acc * d : 0.f);
                    }
                    *reinterpret_cast<f16x8*>(dbuf + row * H64_LD + k) = dv;
                    if (st_d != nullptr && !(skip_d1 && L == 1))
                        H64_STORE(dv, gptr(reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(st_d[L]) + (row0 + row) * 256 + k)));
                }
            }
        });
        if (stash_dy != nullptr && tid < 16 * RG) {
            f32x4 v = *reinterpret_cast<const f32x4*>(s_gy + tid * 4);
#pragma unroll
            for (int a = 0; a < GOPS_MAX_ACT; ++a)
                if (a >= A || tid >= nvalid) v[a] = 0.f;
            *gptr(reinterpret_cast<f32x4*>(stash_dy + (row0 + tid) * 4)) = v;
        }
    }
    DBG_TICK(2)
    __syncthreads();
    DBG_TICK(3)
    after_head();   // the caller's prefetches: they travel together with the layer step's act' operands, in front of its GEMM
    const int f0 = 64 * wave + 16 * g;
    for (int j = L - 1; j >= 1; --j) {   // delta_j = (delta_{j+1} W_j) * act'_j
        f16x8 hv[RG][2];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {   // act' operands: in flight during the GEMM
            const int row = 16 * rg + m;
            hv[rg][0] = hv[rg][1] = zero8h();
            if (row < nvalid) {
                const GLOBAL_AS _Float16* src = gptr(reinterpret_cast<const _Float16*>(gelu ? st_z[j] : st_h[j]) + (row0 + row) * 256 + f0);
                hv[rg][0] = ld8h(src);
                hv[rg][1] = ld8h(src + 8);
            }
        }
        f32x4 acc[4][RG] = {};
        gemm_quad_h64<RG>(dbuf, H64_LD, M.dims[j + 1] >> 5, M.wpth[j], wave, lane, acc);
        DBG_TICK(4)
        __syncthreads();   // every wave has read the delta tile it is about to overwrite
        DBG_TICK(5)
        act_dispatch(M.act, [&]<int ACT>() {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const int row = 16 * rg + m;
                f16x8 o[2];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 4 * jj + r;
                        const float hf = (float)hv[rg][e >> 3][e & 7];
                        const float d = (ACT == GOPS_ACT_GELU) ? hf : act_bwd_t<ACT>(hf, hf);
                        o[e >> 3][e & 7] = sat_h((row < nvalid) ? acc[jj][rg][r] * d : 0.f);
                    }
                *reinterpret_cast<f16x8*>(dbuf + row * H64_LD + f0) = o[0];
                *reinterpret_cast<f16x8*>(dbuf + row * H64_LD + f0 + 8) = o[1];
                if (st_d != nullptr && !(skip_d1 && j == 1)) {   // (skip_d1: the caller forms the first layer's gradient from the LDS tile)
                    _Float16* dst = reinterpret_cast<_Float16*>(st_d[j]) + (row0 + row) * 256 + f0;
                    H64_STORE(o[0], gptr(reinterpret_cast<f16x8*>(dst)));
                    H64_STORE(o[1], gptr(reinterpret_cast<f16x8*>(dst + 8)));
                }
            }
        });
        DBG_TICK(6)
        __syncthreads();
        DBG_TICK(7)
    }
    if (want_gx) {   // g_x = delta_1 W_0: 16-feature tiles over the (16-padded) inputs; wave w takes row group w % RG, n-tiles w / RG, ...
        const int kch = M.dims[1] >> 5, nt_tot = M.kp[0] >> 4;
        const int rgw = wave % RG;
        const _Float16* brow = dbuf + (16 * rgw + m) * H64_LD + 8 * g;
        for (int nt = wave / RG; nt < nt_tot; nt += 4 / RG) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const GLOBAL_AS f16x8* wb = gptr(M.wpth[0]) + (size_t)nt * kch * 64 + lane;
            for (int c = 0; c < kch; c += 2) {
                const f16x8 a0 = wb[(size_t)c * 64], a1 = wb[(size_t)(c + 1) * 64];
                acc0 = MFMA_F16(a0, ld8h(brow + 32 * c), acc0);
                acc1 = MFMA_F16(a1, ld8h(brow + 32 * (c + 1)), acc1);
            }
            const int f = 16 * nt + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (f + r < ncols) G[(16 * rgw + m) * ldg + f + r] += acc0[r] + acc1[r];
        }
    }
    DBG_TICK(8)
}

// The sweep walks the forward's 64-row stash tiles in tiles of H64_BWD_RG x 16 rows.  Measured at cfg5 (round 4): 64 rows per
// workgroup (two workgroups = 8 waves per CU, 235 registers) 0.491 ms; 32 rows (three workgroups = 12 waves per CU, 143
// registers) 0.587 ms - the weight stream per trajectory doubles and costs more than the extra waves hide.
#ifndef H64_BWD_RG
#define H64_BWD_RG 4
#endif
#define H64_BWD_WGS (H64_BWD_RG == 4 ? 2 : 3)
// (+ the fused first-layer gradient: the policy-input rows of two steps [2][TBW][8] and the workgroup's accumulators [256][9])
#define H64_W0_COLS 8
size_t rollout_bwd_h64_lds_bytes(int ldx, int ldh) {
    return sizeof(float) * (size_t)(16 * H64_BWD_RG * ldx + 16 * H64_BWD_RG * 4 + 4 * ldh + ENV_LDS_FLOATS + 2 * 16 * H64_BWD_RG * H64_W0_COLS + 256 * (H64_W0_COLS + 1)) +
           sizeof(_Float16) * (size_t)(16 * H64_BWD_RG * H64_LD);
}

template <int ENV, bool TAIL, int RG>
__global__ __launch_bounds__(NTHREADS, H64_BWD_WGS) void rollout_bwd_h64_kernel(const RolloutParams* __restrict__ pp, const BwdPatch q) {
    constexpr int TBW = 16 * RG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutParams& p = *pp;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, b0 = tile * TBW, nvalid = min(TBW, p.B - b0);
    const int ftile = b0 / TB64, fsub = b0 % TB64;   // the forward's 64-row stash tile this tile is part of
    const int O = p.env.obs_dim, A = p.env.act_dim;
    const int ldx = p.ldx, ldh = p.ldh;
#ifdef H64_SKEW_US   // experiment: every second wave of workgroups starts late, so that co-resident workgroups run different phases
    if ((blockIdx.x >> 8) & 1) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (long long)(H64_SKEW_US) * 100) __builtin_amdgcn_s_sleep(32);
    }
#endif
    float* G = smem;                  // [64][ldx] adjoint of obs_{t+1}
    float* s_gy = G + TBW * ldx;      // [TBW][4]
    float* s_wo = s_gy + TBW * 4;     // [4][ldh] head weights
    float* s_env = s_wo + 4 * ldh;    // GopsEnv copy
    const GopsEnv& env = *reinterpret_cast<const GopsEnv*>(s_env);
    for (int idx = tid; idx < (int)(sizeof(GopsEnv) / 4); idx += NTHREADS) s_env[idx] = gptr(reinterpret_cast<const float*>(&p.env))[idx];
    float* s_x = s_env + ENV_LDS_FLOATS;                 // [2][TBW][8]  policy-input rows of this step / the step before (fused first-layer gradient)
    float* s_w0 = s_x + 2 * TBW * H64_W0_COLS;           // [256][9]     this workgroup's sums of delta_1^T x and of delta_1, by feature
    _Float16* dbuf = reinterpret_cast<_Float16*>(s_w0 + 256 * (H64_W0_COLS + 1));   // [64][H64_LD]
    // The first layer's weight / bias gradient formed here (BwdPatch::w0_part; ENV_LQ, <= 8 policy inputs): delta_1 never goes to the
    // stash (a third of the sweep's HBM writes) and the layer's GEMM launch (which read it back: 0.38 GB at cfg5) is gone.  The sums
    // of step t are formed by waves 1 .. 3 during the env phase of step t - 1, which occupies wave 0 only (delta_1 of step t stays in
    // `dbuf` until the network sweep of step t - 1 starts behind that phase's barrier); step 0's by everybody behind the loop.
    const bool fuse_w0 = q.w0_part != nullptr;   // (pyth_lq rollouts and plain value batches: api.hip h64_fuses_dw0)
    const int K0 = p.pol.dims[0];
    auto w0_accumulate = [&](int feature, const float* xrows) {
        const _Float16* dcol = dbuf + feature;
        float* acc = s_w0 + feature * (H64_W0_COLS + 1);
        if (K0 <= 4) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, ab = 0.f;
#pragma unroll 8
            for (int row = 0; row < TBW; ++row) {
                const float d = (float)dcol[row * H64_LD];
                const f32x4 x = *reinterpret_cast<const f32x4*>(xrows + row * H64_W0_COLS);
                a0 = fmaf(d, x[0], a0); a1 = fmaf(d, x[1], a1); a2 = fmaf(d, x[2], a2); a3 = fmaf(d, x[3], a3);
                ab += d;
            }
            acc[0] += a0; acc[1] += a1; acc[2] += a2; acc[3] += a3; acc[H64_W0_COLS] += ab;
        } else {
            float a[H64_W0_COLS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ab = 0.f;
#pragma unroll 4
            for (int row = 0; row < TBW; ++row) {
                const float d = (float)dcol[row * H64_LD];
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xrows + row * H64_W0_COLS), x1 = *reinterpret_cast<const f32x4*>(xrows + row * H64_W0_COLS + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { a[k] = fmaf(d, x0[k], a[k]); a[4 + k] = fmaf(d, x1[k], a[4 + k]); }
                ab += d;
            }
#pragma unroll
            for (int k = 0; k < H64_W0_COLS; ++k) acc[k] += a[k];
            acc[H64_W0_COLS] += ab;
        }
    };
    {
        const int Lh = p.pol.nl - 1, K = p.pol.dims[Lh], Ao = p.pol.dims[p.pol.nl];
        for (int idx = tid; idx < Ao * K; idx += NTHREADS) {
            const int a = idx / K, k = idx - a * K;
            s_wo[a * ldh + k] = gptr(p.pol.w[Lh])[idx];
        }
    }
    if (fuse_w0)
        for (int idx = tid; idx < 256 * (H64_W0_COLS + 1); idx += NTHREADS) s_w0[idx] = 0.f;
    for (int idx = tid; idx < TBW * ldx; idx += NTHREADS) G[idx] = 0.f;
    DbgClock dbg;   // phase counters of thread 0 (GOPS_DBG_BUILD + GOPS_DBG_TIMING=1, tools/dbg_run.py)
    dbg.init(false);
    auto step_row0 = [&](int t) -> size_t {
        return ((size_t)t * ((p.B + TB64 - 1) / TB64) + ftile) * TB64 + fsub;   // step-major stash tiles (as the forward wrote them)
    };
    // env-stash row and policy input of a trajectory, requested one step ahead (from inside the previous step's network sweep:
    // read at the top of the env phase they cost that phase a full HBM round trip on ONE wave while three wait at the barrier)
    struct EnvRow { f32x4 e0, xa, xb; float dflag; };
    auto fetch_env = [&](int t, EnvRow& r) {
        r.e0 = r.xa = r.xb = f32x4{0.f, 0.f, 0.f, 0.f};
        r.dflag = 1.f;
        if ((ENV != GOPS_ENV_NONE || fuse_w0) && t >= 0 && tid < nvalid) {
            const size_t row = step_row0(t) + tid;
            const GLOBAL_AS f32x4* xr = gptr(reinterpret_cast<const f32x4*>(p.st.xf + row * 8));
            if (ENV != GOPS_ENV_NONE) {
                const GLOBAL_AS f32x4* er = gptr(reinterpret_cast<const f32x4*>(p.st.env + row * ENV_STASH));
                r.e0 = er[0];
                r.dflag = er[1][0];
            }
            r.xa = xr[0];
            r.xb = xr[1];
        }
    };
    EnvRow cur, nxt;
    float gv = (tid < nvalid) ? gptr(q.grad_v)[b0 + tid] : 0.f;
    gv *= f16_grad_scale(gptr(p.gscale)[0]);
    if (TAIL) {
        if (tid < TBW) {
            const float dH = (tid < nvalid) ? gptr(p.st.tail_done)[b0 + tid] : 1.f;
            s_gy[tid * 4 + 0] = gv * ((p.tail_unmasked ? 1.f : 1.f - dH) * p.gpow[p.H]);
            s_gy[tid * 4 + 1] = s_gy[tid * 4 + 2] = s_gy[tid * 4 + 3] = 0.f;
        }
        __syncthreads();
        mlp_backward_h64<RG>(p.val, gptr(p.val.w[p.val.nl - 1]), p.val.dims[p.val.nl - 1], s_gy, dbuf, G, ldx, tid, p.st.tail_h, p.st.tail_z, nullptr,
                         nullptr, (size_t)b0, nvalid, true, O, dbg, [&] { fetch_env(p.H - 1, cur); });
    } else {
        fetch_env(p.H - 1, cur);
    }
    __syncthreads();
    dbg.init((q.dbg != nullptr) && blockIdx.x == 0 && tid == 0);
    for (int t = p.H - 1; t >= 0; --t) {
        const size_t row0 = step_row0(t);
        float g_r = gv * p.gpow[t];
        if (ENV != GOPS_ENV_NONE && env.shaping) g_r *= env.reward_scale;
        if (fuse_w0 && tid >= TBW && t < p.H - 1) {   // waves 1 .. 3, while wave 0 steps the env adjoint: the sums of step t + 1
            const int u = tid - TBW;                   // 192 threads, 256 features: the first 64 take two
            const float* xrows = s_x + ((t + 1) & 1) * (TBW * H64_W0_COLS);
            w0_accumulate(u, xrows);
            if (u < 256 - (NTHREADS - TBW)) w0_accumulate(u + (NTHREADS - TBW), xrows);
        }
        if (tid < TBW) {
            const int m = tid;
            if (fuse_w0) {   // the row's policy input as the GEMM would read it from the half X stash (rows beyond the batch: zeros)
                f32x4 xa = cur.xa, xb = cur.xb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xa[i] = (m < nvalid && i < K0) ? (float)(_Float16)xa[i] : 0.f;
                    xb[i] = (m < nvalid && 4 + i < K0) ? (float)(_Float16)xb[i] : 0.f;
                }
                float* xr = s_x + (t & 1) * (TBW * H64_W0_COLS) + m * H64_W0_COLS;
                *reinterpret_cast<f32x4*>(xr) = xa;
                *reinterpret_cast<f32x4*>(xr + 4) = xb;
            }
            if (ENV == GOPS_ENV_NONE) {
                s_gy[m * 4 + 0] = g_r;
                s_gy[m * 4 + 1] = s_gy[m * 4 + 2] = s_gy[m * 4 + 3] = 0.f;
            } else {   // GOPS_ENV_LQ (the arithmetic of rollout_bwd_kernel's LQ block); NS / NA: compile-time loop bounds
                auto lq_adjoint = [&]<int NS, int NA>() {
                    constexpr bool EXACT = NS < GOPS_MAX_LQ_STATE;   // the dimensions ARE (NS, NA): no run-time guards
                    const float th[GOPS_MAX_ACT] = {cur.e0[0], cur.e0[1], cur.e0[2], cur.e0[3]}, dflag = cur.dflag;
                    float x[GOPS_MAX_LQ_STATE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (m < nvalid) {
#pragma unroll
                        for (int i = 0; i < NS; ++i)
                            if (EXACT || i < O) x[i] = obs_unscale(env, i, i < 4 ? cur.xa[i & 3] : cur.xb[i & 3]);   // the stash holds the (scaled) policy input
                    }
                    float abar[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, u[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f}, sc[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f},
                          gu[GOPS_MAX_ACT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        sc[a] = (env.policy_high[a] - env.policy_low[a]) / 2.f;
                        abar[a] = sc[a] * th[a] + (env.policy_high[a] + env.policy_low[a]) / 2.f;
                        u[a] = (EXACT || a < A) ? wrap_action(env, a, abar[a]) : 0.f;
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
                    if (env.scale_obs) {
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
                        if (EXACT || i < O) G[m * ldx + i] = env.scale_obs ? gx[i] / env.obs_scale[i] : gx[i];
#pragma unroll
                    for (int a = 0; a < GOPS_MAX_ACT; ++a)
                        s_gy[m * 4 + a] = (a < NA && (EXACT || a < A)) ? wrap_action_bwd(env, a, abar[a], gu[a]) * sc[a] * (1.f - th[a] * th[a]) : 0.f;
                };
                if (O == 4 && A == 2) lq_adjoint.template operator()<4, 2>();   // BASELINE configs[4] (lq s4a2)
                else lq_adjoint.template operator()<GOPS_MAX_LQ_STATE, GOPS_MAX_ACT>();
            }
        }
        DBG_TICK(0)
        __syncthreads();
        DBG_TICK(1)
        mlp_backward_h64<RG>(p.pol, s_wo, ldh, s_gy, dbuf, G, ldx, tid, p.st.h, p.st.z, p.st.d, p.st.dy, row0, nvalid,
                         /*want_gx=*/t > 0 && ENV != GOPS_ENV_NONE, O, dbg, [&] { fetch_env(t - 1, nxt); }, fuse_w0);
        cur = nxt;
        __syncthreads();
        DBG_TICK(9)
    }
    dbg.dump(q.dbg);
    if (fuse_w0) {   // step 0's sums (thread n <-> feature n; the loop's closing barrier published delta_1), then this workgroup's slab
        w0_accumulate(tid, s_x);   // (step 0: buffer 0)
        const float* acc = s_w0 + tid * (H64_W0_COLS + 1);
        f32x4 lo = {acc[0], acc[1], acc[2], acc[3]}, hi = {acc[4], acc[5], acc[6], acc[7]};
        float* dst = q.w0_part + ((size_t)blockIdx.x * 256 + tid) * H64_W0_COLS;
        *gptr(reinterpret_cast<f32x4*>(dst)) = lo;
        *gptr(reinterpret_cast<f32x4*>(dst + 4)) = hi;
        gptr(q.w0_part_b)[(size_t)blockIdx.x * 256 + tid] = acc[H64_W0_COLS];
    }
    if (q.ad_st != nullptr && blockIdx.x == 0 && threadIdx.x == 0) adam_snapshot(q.ad_st, q.ad_snap, q.ad_b1, q.ad_b2);   // (gops_rollout_backward_update)
}

hipError_t launch_rollout_fwd_h64(const RolloutParams& p, const RolloutParams* dp, hipStream_t stream) {
    const dim3 grid((p.B + TB64 - 1) / TB64), block(NTHREADS);
    size_t lds = rollout_fwd_h64_lds_bytes(p.ldx, p.ldh);
    if (p.dbg != nullptr) lds = max(lds, (size_t)84 * 1024);   // phase counters: ONE workgroup per CU, so that the phases of thread 0 add up
    if (p.env.kind == GOPS_ENV_LQ) {
        if (p.tail) launch_with_lds(rollout_fwd_h64_kernel<GOPS_ENV_LQ, true>, grid, block, lds, stream, dp);
        else launch_with_lds(rollout_fwd_h64_kernel<GOPS_ENV_LQ, false>, grid, block, lds, stream, dp);
    } else if (p.env.kind == GOPS_ENV_NONE) {
        if (p.tail) return hipErrorInvalidValue;
        launch_with_lds(rollout_fwd_h64_kernel<GOPS_ENV_NONE, false>, grid, block, lds, stream, dp);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_rollout_bwd_h64(const RolloutParams& p, const RolloutParams* dp, const BwdPatch& q, hipStream_t stream) {
    constexpr int RG = H64_BWD_RG, TBW = 16 * RG;
    const dim3 grid((p.B + TBW - 1) / TBW), block(NTHREADS);
    size_t lds = rollout_bwd_h64_lds_bytes(p.ldx, p.ldh);
    if (q.dbg != nullptr) lds = max(lds, (size_t)84 * 1024);   // (as in the forward)
    if (p.env.kind == GOPS_ENV_LQ) {
        if (p.tail) launch_with_lds(rollout_bwd_h64_kernel<GOPS_ENV_LQ, true, RG>, grid, block, lds, stream, dp, q);
        else launch_with_lds(rollout_bwd_h64_kernel<GOPS_ENV_LQ, false, RG>, grid, block, lds, stream, dp, q);
    } else if (p.env.kind == GOPS_ENV_NONE) {
        if (p.tail) return hipErrorInvalidValue;
        launch_with_lds(rollout_bwd_h64_kernel<GOPS_ENV_NONE, false, RG>, grid, block, lds, stream, dp, q);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Launches whose first-layer weight gradient the sweep forms itself (BwdPatch::w0_part): pyth_lq policies and value nets with <= 8 inputs
// (GOPS_VF_NO_FUSED_DW0: the GEMM path, for A/B)
int h64_sweep_grid(const RolloutParams& p) { return (p.B + 16 * H64_BWD_RG - 1) / (16 * H64_BWD_RG); }
bool h64_fuses_dw0(const RolloutParams& p) {
    return p.f16 && p.h64 && (p.env.kind == GOPS_ENV_LQ || p.env.kind == GOPS_ENV_NONE) && p.pol.dims[0] <= H64_W0_COLS && p.pol.dims[1] == 256 &&
           p.need_grad &&
           !(p.vflags & GOPS_VF_NO_FUSED_DW0);
}

// The launches the 64-row half kernels take (api.hip build_plan; GOPS_VF_NO_HALF_TILE64 keeps the 16-row kernels)
bool h64_eligible(const RolloutParams& p) {
    if (!p.f16 || p.open_loop || p.ext || p.env.repeat_num > 1) return false;
    if (p.vflags & GOPS_VF_NO_HALF_TILE64) return false;
    if (p.env.kind != GOPS_ENV_LQ && p.env.kind != GOPS_ENV_NONE) return false;
    if (p.env.kind == GOPS_ENV_NONE && p.tail) return false;
    auto net_ok = [](const MlpDev& M) {
        if (M.nl < 2 || M.kp32[0] > 64) return false;
        for (int j = 1; j < M.nl; ++j)
            if (M.dims[j] != 256) return false;
        return true;
    };
    if (!net_ok(p.pol) || (p.tail && !net_ok(p.val))) return false;
    return rollout_fwd_h64_lds_bytes(p.ldx, p.ldh) <= 80 * 1024 && rollout_bwd_h64_lds_bytes(p.ldx, p.ldh) <= 80 * 1024;
}
