// Forward horizon rollout: one workgroup (4 wavefronts) owns a tile of 16 trajectories and walks
// all H timesteps without leaving the CU.  Per step: policy MLP (hidden layers on
// v_mfma_f32_16x16x4_f32 with the activations staged in LDS and the weights streamed from L2 in
// MFMA-fragment order), tanh head + wrapper chain, env model step, masked/shaped reward into the
// discounted return.  Replaces the Python loop of fhadp.py:117-120 / infadp.py:171-180,198-208.
#include "common.h"
#include "env_models.h"

// Hidden layers of `M` applied to the LDS tile `in` (TB x kp[0], leading dim ld_in).  Returns the
// LDS buffer that holds the last hidden activation.  When stash_h is non-null the activations
// (and GELU pre-activations) of the tile are written to stash_h[j] + row0 * dims[j].
__device__ __forceinline__ float* mlp_hidden_forward(const MlpDev& M, const float* in, int ld_in,
                                                     float* ha, float* hb, int ldh, int tid,
                                                     float* const* stash_h, float* const* stash_z,
                                                     size_t row0, int nvalid) {
    const int lane = tid & 63;
    const int L = M.nl - 1;
    const float* cur = in;
    int ldc = ld_in;
    float* out = ha;
    for (int j = 0; j < L; ++j) {
        const int N = M.dims[j + 1], kch = M.kp[j] >> 4, nt_tot = N >> 4;
        const float* bias = M.b[j];
        const bool save_z = (stash_z != nullptr) && (M.act == GOPS_ACT_GELU);
        float* zrow = save_z ? stash_z[j + 1] + row0 * N : nullptr;
        gemm_layer(cur, ldc, kch, nt_tot, M.wp[j], tid, [&](const f32x4& acc, int ntile) {
            const int n = (ntile << 4) + (lane & 15);
            const float bn = bias[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = ((lane >> 4) << 2) + r;
                const float z = acc[r] + bn;
                out[m * ldh + n] = act_fwd(M.act, z);
                if (save_z && m < nvalid) zrow[(size_t)m * N + n] = z;
            }
        });
        __syncthreads();
        if (stash_h != nullptr) stash_tile(out, ldh, N, stash_h[j + 1], row0, nvalid, tid);
        cur = out;
        ldc = ldh;
        out = (out == ha) ? hb : ha;
    }
    return const_cast<float*>(cur);
}

// Output layer (width A <= 4) on the VALU: thread (hm = tid>>4, hp = tid&15) strides over k.
// Result y[a] valid in the lanes with hp == 0.
__device__ __forceinline__ void mlp_head(const MlpDev& M, const float* hcur, int ldh, int tid,
                                         float (&y)[GOPS_MAX_ACT]) {
    const int L = M.nl - 1, K = M.dims[L], A = M.dims[M.nl];
    const int hm = tid >> 4, hp = tid & 15;
    const float* Wo = M.w[L];
#pragma unroll
    for (int a = 0; a < GOPS_MAX_ACT; ++a) y[a] = 0.f;
    for (int k = hp; k < K; k += 16) {
        const float hv = hcur[hm * ldh + k];
#pragma unroll
        for (int a = 0; a < GOPS_MAX_ACT; ++a)
            if (a < A) y[a] += hv * Wo[a * K + k];
    }
#pragma unroll
    for (int a = 0; a < GOPS_MAX_ACT; ++a) {
        y[a] += __shfl_xor(y[a], 1);
        y[a] += __shfl_xor(y[a], 2);
        y[a] += __shfl_xor(y[a], 4);
        y[a] += __shfl_xor(y[a], 8);
        if (a < A) y[a] += M.b[L][a];
    }
}

template <int ENV>
__global__ __launch_bounds__(NTHREADS) void rollout_fwd_kernel(const RolloutParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * TB;
    const int nvalid = min(TB, p.B - b0);
    const int O = p.env.obs_dim, A = p.env.act_dim;
    const int ldx = p.ldx, ldh = p.ldh;
    float* xs = smem;                       // [TB][ldx]  current observation (+ time column)
    float* ha = xs + TB * ldx;              // [TB][ldh]
    float* hb = ha + TB * ldh;              // [TB][ldh]
    float* s_state = hb + TB * ldh;         // [TB][8]
    float* s_act = s_state + TB * 8;        // [TB][4] wrapped action
    float* s_th = s_act + TB * 4;           // [TB][4] tanh(head) (ENV_NONE: raw head output)
    float* s_done = s_th + TB * 4;          // [TB]

    for (int idx = tid; idx < TB * ldx; idx += NTHREADS) {
        const int m = idx / ldx, c = idx - m * ldx;
        xs[idx] = (c < O && m < nvalid) ? p.in.obs[(size_t)(b0 + m) * O + c] : 0.f;
    }
    if (tid < TB) s_done[tid] = (tid < nvalid && p.in.done != nullptr && p.in.done[b0 + tid] != 0.f) ? 1.f : 0.f;
    if (ENV == GOPS_ENV_VEH3DOFCONTI) {
        if (tid < TB * 6) {
            const int m = tid / 6, c = tid - m * 6;
            s_state[m * 8 + c] = (m < nvalid) ? p.in.state[(size_t)(b0 + m) * 6 + c] : (c == 3 ? 1.f : 0.f);
        }
    }
    float v_acc = 0.f;
    const IdpConst IC = idp_const();
    const VehConst VC = veh_const();
    const int TL = p.env.pre_horizon + 1 + p.H;   // reference-table points per trajectory

    for (int t = 0; t < p.H; ++t) {
        if (p.fh && tid < TB) xs[tid * ldx + O] = (float)(t + 1);
        __syncthreads();
        const size_t row0 = (size_t)t * p.B + b0;
        if (p.need_grad) stash_tile(xs, ldx, p.pol.kp[0], p.st.x, row0, nvalid, tid);
        float* hcur = mlp_hidden_forward(p.pol, xs, ldx, ha, hb, ldh, tid,
                                         p.need_grad ? p.st.h : nullptr, p.need_grad ? p.st.z : nullptr,
                                         row0, nvalid);
        {
            float y[GOPS_MAX_ACT];
            mlp_head(p.pol, hcur, ldh, tid, y);
            if ((tid & 15) == 0) {
                const int hm = tid >> 4;
#pragma unroll
                for (int a = 0; a < GOPS_MAX_ACT; ++a) {
                    if (a < A) {
                        if (ENV == GOPS_ENV_NONE) {
                            s_th[hm * 4 + a] = y[a];
                        } else {
                            const float th = tanhf(y[a]);
                            const float sc = (p.env.policy_high[a] - p.env.policy_low[a]) / 2.f;
                            const float of = (p.env.policy_high[a] + p.env.policy_low[a]) / 2.f;
                            s_th[hm * 4 + a] = th;
                            s_act[hm * 4 + a] = wrap_action(p.env, a, sc * th + of);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (p.need_grad && tid < nvalid) {   // env stash row: tanh outputs, done_t, state_t
            float* er = p.st.env + (row0 + tid) * ENV_STASH;
            f32x4 e0 = {s_th[tid * 4 + 0], s_th[tid * 4 + 1], s_th[tid * 4 + 2], s_th[tid * 4 + 3]};
            f32x4 e1 = {s_done[tid], s_state[tid * 8 + 0], s_state[tid * 8 + 1], s_state[tid * 8 + 2]};
            f32x4 e2 = {s_state[tid * 8 + 3], s_state[tid * 8 + 4], s_state[tid * 8 + 5], 0.f};
            reinterpret_cast<f32x4*>(er)[0] = e0;
            reinterpret_cast<f32x4*>(er)[1] = e1;
            reinterpret_cast<f32x4*>(er)[2] = e2;
        }

        // ---------------- env model step + MaskAtDone / ShapingReward / ClipObservation ---------
        float r = 0.f;          // raw model reward (threads tid < TB)
        bool done_m = false;    // done flag from the base model
        if (ENV == GOPS_ENV_NONE) {
            if (tid < TB) r = s_th[tid * 4];
        } else if (ENV == GOPS_ENV_LQ) {
            if (tid < TB) {
                const int m = tid;
                float x[GOPS_MAX_LQ_STATE], xn[GOPS_MAX_LQ_STATE], u[GOPS_MAX_ACT];
#pragma unroll
                for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i) { x[i] = (i < O) ? xs[m * ldx + i] : 0.f; xn[i] = 0.f; }
#pragma unroll
                for (int j = 0; j < GOPS_MAX_ACT; ++j) u[j] = (j < A) ? s_act[m * 4 + j] : 0.f;
                lq_forward(p.env, x, u, xn, r);
                if (s_done[m] == 0.f) {
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                        if (i < O) xs[m * ldx + i] = p.env.clip_obs ? clampf(xn[i], p.env.obs_low[i], p.env.obs_high[i]) : xn[i];
                } else if (p.env.clip_obs) {
#pragma unroll
                    for (int i = 0; i < GOPS_MAX_LQ_STATE; ++i)
                        if (i < O) xs[m * ldx + i] = clampf(x[i], p.env.obs_low[i], p.env.obs_high[i]);
                }
            }
        } else if (ENV == GOPS_ENV_IDPENDULUM) {
            if (tid < TB) {
                const int m = tid;
                float s[6], sn[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) s[i] = xs[m * ldx + i];
                const float a = s_act[m * 4];
                const float u = 500.f * a;
                IdpSub w;
#pragma unroll 1
                for (int k = 0; k < 5; ++k) {
                    idp_substep(IC, s, u, 0.002f, sn, w);
#pragma unroll
                    for (int i = 0; i < 6; ++i) s[i] = sn[i];
                }
                r = idp_reward(s, a);
                done_m = idp_done(IC, s);
                if (s_done[m] == 0.f) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) xs[m * ldx + i] = s[i];
                }
            }
        } else {   // GOPS_ENV_VEH3DOFCONTI: all 256 threads, thread = (trajectory m, part)
            const int m = tid & 15, part = tid >> 4;
            const int P = p.env.pre_horizon;
            float s[6], sn[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) s[i] = s_state[m * 8 + i];
            const float steer = s_act[m * 4 + 0], ax = s_act[m * 4 + 1];
            const float dflag = s_done[m];
            VehStep w;
            veh_f_xu(VC, s, steer, ax, sn, w);
            if (part == 0) {
                float o[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) o[i] = xs[m * ldx + i];
                r = veh_reward(o, steer, ax);
            }
            __syncthreads();   // every read of the old obs / state is done
            float cn, snn;
            sincosf(-sn[2], &snn, &cn);
            const f32x4* tbl = reinterpret_cast<const f32x4*>(p.ref_table) + (size_t)(b0 + m) * TL + (t + 1);
            for (int j = part; j <= P; j += 16) {
                f32x4 rp = {0.f, 0.f, 0.f, 0.f};
                if (m < nvalid) rp = tbl[j];
                const float dx = rp[0] - sn[0], dy = rp[1] - sn[1];
                const float xtf = dx * cn - dy * snn;
                const float ytf = dx * snn + dy * cn;
                const float ptf = angle_normalize(rp[2] - sn[2]);
                const float utf = rp[3] - sn[3];
                if (j == 0) {
                    done_m = (fabsf(xtf) > 10.f) || (fabsf(ytf) > 10.f) || (fabsf(ptf) > 3.14159265358979323846f);
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
        }
        if (tid < TB) {
            const float d = s_done[tid];
            float rr = (d != 0.f) ? 0.f : r;
            if (ENV != GOPS_ENV_NONE && p.env.shaping) rr = (rr + p.env.reward_shift) * p.env.reward_scale;
            v_acc += rr * p.gpow[t];
            if (p.out.rewards != nullptr && tid < nvalid) p.out.rewards[(size_t)t * p.B + b0 + tid] = rr;
            if (done_m) s_done[tid] = 1.f;
        }
    }
    __syncthreads();

    if (p.tail) {   // v += (~done_H) * gamma^H * V_target(obs_H)   (infadp.py:182-184, 210)
        if (p.fh && tid < TB) xs[tid * ldx + O] = 0.f;
        __syncthreads();
        float* hcur = mlp_hidden_forward(p.val, xs, ldx, ha, hb, ldh, tid,
                                         p.need_grad ? p.st.tail_h : nullptr,
                                         p.need_grad ? p.st.tail_z : nullptr, (size_t)b0, nvalid);
        float y[GOPS_MAX_ACT];
        mlp_head(p.val, hcur, ldh, tid, y);
        if ((tid & 15) == 0) s_th[(tid >> 4) * 4] = y[0];
        __syncthreads();
        if (tid < TB) v_acc += ((1.f - s_done[tid]) * p.gpow[p.H]) * s_th[tid * 4];
    }

    if (tid < nvalid) {
        p.out.v_pi[b0 + tid] = v_acc;
        if (p.out.final_done != nullptr) p.out.final_done[b0 + tid] = s_done[tid];
        if (p.need_grad && p.st.tail_done != nullptr) p.st.tail_done[b0 + tid] = s_done[tid];
    }
    if (p.out.final_obs != nullptr) {
        for (int idx = tid; idx < TB * O; idx += NTHREADS) {
            const int m = idx / O, c = idx - m * O;
            if (m < nvalid) p.out.final_obs[(size_t)(b0 + m) * O + c] = xs[m * ldx + c];
        }
    }
    if (ENV == GOPS_ENV_VEH3DOFCONTI && p.out.final_state != nullptr && tid < TB * 6) {
        const int m = tid / 6, c = tid - m * 6;
        if (m < nvalid) p.out.final_state[(size_t)(b0 + m) * 6 + c] = s_state[m * 8 + c];
    }
}

size_t rollout_fwd_lds_bytes(int ldx, int ldh) {
    return sizeof(float) * (size_t)(TB * ldx + 2 * TB * ldh + TB * (8 + 4 + 4 + 1));
}

hipError_t launch_rollout_fwd(const RolloutParams& p, hipStream_t stream) {
    const dim3 grid((p.B + TB - 1) / TB), block(NTHREADS);
    const size_t lds = rollout_fwd_lds_bytes(p.ldx, p.ldh);
    switch (p.env.kind) {
        case GOPS_ENV_NONE: hipLaunchKernelGGL(rollout_fwd_kernel<GOPS_ENV_NONE>, grid, block, lds, stream, p); break;
        case GOPS_ENV_LQ: hipLaunchKernelGGL(rollout_fwd_kernel<GOPS_ENV_LQ>, grid, block, lds, stream, p); break;
        case GOPS_ENV_IDPENDULUM: hipLaunchKernelGGL(rollout_fwd_kernel<GOPS_ENV_IDPENDULUM>, grid, block, lds, stream, p); break;
        case GOPS_ENV_VEH3DOFCONTI: hipLaunchKernelGGL(rollout_fwd_kernel<GOPS_ENV_VEH3DOFCONTI>, grid, block, lds, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
