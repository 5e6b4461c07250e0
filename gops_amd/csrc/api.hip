// Host side of the C ABI (include/gops_hip.h): argument checking, workspace carving, launch
// sequencing.  Nothing here allocates device memory or synchronises the stream.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"

hipError_t launch_rollout_fwd(const RolloutParams& p, const RolloutParams* dp, hipStream_t stream);
hipError_t launch_rollout_bwd(const RolloutParams& p, const RolloutParams* dp, const BwdPatch& q, hipStream_t stream);
size_t rollout_fwd_lds_bytes(int ldx, int ldh, int ref_points, bool f16, int split_k0, bool ss = false);
size_t rollout_bwd_lds_bytes(int ldx, int ldh, int ref_points, bool f16, bool split, bool ssb = false);
bool split_eligible(const RolloutParams& p);
bool h64_fuses_dw0(const RolloutParams& p);   // rollout_h64.hip
int h64_sweep_grid(const RolloutParams& p);
int split_grid_limit();
int ssb_grid_limit();     // rollout_bwd.hip
bool ssb_fuses_out(const RolloutParams& p);
void rollout_variant(const RolloutParams& p, int sk[2], bool backward);
bool h64_eligible(const RolloutParams& p);   // rollout_h64.hip
hipError_t launch_upload_params(const RolloutParams& p, RolloutParams* dst, hipStream_t s);
hipError_t launch_prologue(const RolloutParams& p, RolloutParams* dst, int P, float pdt, hipStream_t s);
hipError_t launch_dw_gemm(const float* D, int N, const float* X, int Kp, long long S, int splits,
                          int chunks_per_split, float* part, float* part_b, bool big, hipStream_t s, const float* dscale, unsigned vflags);
bool dw_skinny_ok(int N, int Kp, unsigned vflags);   // aux_kernels.hip
hipError_t launch_dw_gemm_f16(const void* D, int N, const void* X, int Kp, long long S, int splits,
                              int chunks_per_split, float* part, float* part_b, hipStream_t s);
hipError_t launch_dw_out(const float* dy, const float* h, bool h_is_half, int K, int A, long long S, int splits,
                         float* part, float* part_b, hipStream_t s);
void reduce_jobs_add(ReduceJobs& jobs, const float* part, int splits, int rows, int cols, int ld, float* out, int slab_rows = 0);
hipError_t launch_linear_out_fwd(const float* h, int K, const float* Wo, const float* bo, int W, int B, float* y, hipStream_t s);
hipError_t launch_linear_out_bwd(const float* gy, int W, int Wp, const float* Wo, int K, int B, long long S, float* gh,
                                 float* gyp, hipStream_t s);
hipError_t launch_reduce(const ReduceJobs& jobs, hipStream_t s);
hipError_t launch_fill_zero(float* p, size_t n, hipStream_t s);
hipError_t launch_env_step(const GopsEnv& env, int B, const GopsStepIO& io, float pdt, hipStream_t s);
hipError_t launch_env_constraint(const GopsEnv& env, int B, const GopsStepIO& io, hipStream_t s);
bool ss_eligible(const RolloutParams& p);   // rollout_fwd.hip
bool ss_tail_exact(const RolloutParams& p); // rollout_fwd.hip
bool ssb_eligible(const RolloutParams& p);  // rollout_bwd.hip
hipError_t launch_polyak(const GopsAdamTensors& T, float omt, float tau, hipStream_t s);
hipError_t launch_batch_loss(const float* a, const float* b, int n, float gsc, float sc0, float* grad, float* stats, hipStream_t s);
hipError_t launch_adam(const GopsAdamTensors& T, GopsAdamState* st, double beta1, double beta2, float eps,
                       hipStream_t s);

// One hipFuncSetAttribute per kernel (common.h: launch_with_lds).
bool lds_attr_needed(const void* kernel) {
    static std::mutex mu;
    static std::vector<const void*> seen;
    std::lock_guard<std::mutex> lk(mu);
    for (const void* k : seen)
        if (k == kernel) return false;
    seen.push_back(kernel);
    return true;
}

namespace {

inline int pad16(int x) { return (x + 15) & ~15; }
inline int pad32(int x) { return (x + 31) & ~31; }
constexpr size_t kAlign = 256;   // bytes

// ---- opt-in kernel timing (bench.py) ---------------------------------------------------------
struct ProfState {
    std::mutex mu;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[6];   // 0-2: rollout fwd / bwd / dW; 3-5: the same for plain MLP batches (value net)
} g_prof;

struct ProfScope {
    int id;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(int id_, hipStream_t s_) : id(id_), s(s_) {
        if (g_prof.on) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (a != nullptr) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof.mu);
            g_prof.ev[id].emplace_back(a, b);
        }
    }
};

// ---- workspace plan ----------------------------------------------------------------------------
// GopsEnv.ref_c of the default MultiRefTrajModel parameters (ref_traj_data.py:18-37), folded like the reference folds
// its Python scalars (double arithmetic, one rounding to fp32)
void fill_ref_defaults(GopsEnv& e) {
    if (e.ref_custom) return;
    const double PI = 3.14159265358979323846, w = 2.0 * PI / 10.0;
    const double c[24] = {-1.0 / w, w, 0.0, 5.0, 1.0 / w * 1.0, 1.0,                  // sine speed: A = 1, omega = 2 pi / 10, phi = 0, b = 5
                          5.0,                                                      // constant speed
                          1.5, w, 0.0,                                              // sine path
                          5.0, 9.0, 14.0, 18.0, 0.0, 3.5, 3.5 / 4.0, -3.5 / 4.0,    // double lane
                          10.0, 2.0 * 3.0 / 10.0, -2.0 * 3.0 / 10.0, 10.0 / 2.0,    // triangle: A = 3, T = 10
                          100.0, 0.0};                                              // circle
    for (int i = 0; i < 24; ++i) e.ref_c[i] = (float)c[i];
}

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    float* take(size_t nfloats) {
        off = (off + kAlign - 1) / kAlign * kAlign;
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += nfloats * sizeof(float);
        return p;
    }
};

struct DwPlan {
    int splits, chunks_per_split;
    bool big;
};

// ---- debug override of the variant flags from the process environment ------------------------------------------------
// Kernel variants are selected by GopsRolloutDesc.variant_flags / GopsMlp.variant_flags (ABI v10).  The environment knobs of
// ABI v9 survive as a DEBUG override only: read ONCE, when the first call reaches the library, OR-ed into every description.
struct EnvOverride {
    unsigned flags = 0;
    int l2_warmup = 0;      // as GopsRolloutDesc.l2_warmup (GOPS_TOUCH = mode -> mode + 1)
    int dw_wgs = 0;
    bool dbg_timing = false;
};
const EnvOverride& env_override() {
    static const EnvOverride o = [] {
        struct Knob { const char* name; char match; unsigned bit; int kind; };   // kind 0: flag when set (match 0) / when value[0] == match
        static const Knob knobs[] = {
            {"GOPS_SPLIT", '0', GOPS_VF_NO_STATIONARY_SPLIT, 0}, {"GOPS_SS", '0', GOPS_VF_NO_STREAMED_SPLIT_FWD, 0},
            {"GOPS_SSB", '0', GOPS_VF_NO_STREAMED_SPLIT_BWD, 0}, {"GOPS_SS_VALUE", '0', GOPS_VF_NO_STREAMED_SPLIT_VALUE, 0},
            {"GOPS_SPLIT_STREAM0", '0', GOPS_VF_NO_SPLIT_STREAM0, 0}, {"GOPS_SPLIT_TAIL_MULTI", 0, GOPS_VF_SPLIT_TAIL_MULTI, 0},
            {"GOPS_DW_EXACT", 0, GOPS_VF_DW_EXACT, 0}, {"GOPS_DW_F32", 0, GOPS_VF_DW_F32, 0}, {"GOPS_DW_NOGUARD", 0, GOPS_VF_DW_NO_GUARD, 0}, {"GOPS_NO_FUSED_DW0", 0, GOPS_VF_NO_FUSED_DW0, 0},
            {"GOPS_DW_SKINNY", '0', GOPS_VF_DW_NO_SKINNY, 0}, {"GOPS_DW_SPEC", '0', GOPS_VF_DW_NO_SPEC, 0}, {"GOPS_DW_DIRECT", 0, GOPS_VF_DW_DIRECT, 0},
            {"GOPS_NO_FUSED_DWOUT", 0, GOPS_VF_NO_FUSED_DWOUT, 0}, {"GOPS_H64", '0', GOPS_VF_NO_HALF_TILE64, 0}, {"GOPS_NARROW", '0', GOPS_VF_NO_NARROW_LDS, 0}, {"GOPS_N64", '0', GOPS_VF_NO_NARROW_N64, 0}, {"GOPS_BWD_UPLOAD", 0, GOPS_VF_BWD_UPLOAD, 0},
            {"GOPS_SK", 0, 0, 1}, {"GOPS_TOUCH", 0, 0, 2}, {"GOPS_DW_WGS", 0, 0, 3}, {"GOPS_DBG_TIMING", 0, 0, 4}};
        EnvOverride r;
        for (const Knob& k : knobs) {
            const char* e = getenv(k.name);
            if (e == nullptr) continue;
            if (k.kind == 0) { if (k.match == 0 || e[0] == k.match) r.flags |= k.bit; }
            else if (k.kind == 1) {   // GOPS_SK="a,b": 0,0 = plain streamed kernels, 0,16 = layer 0 streamed; any value: stationary at any batch
                int a = -1, b = -1;
                r.flags |= GOPS_VF_STATIONARY_ANY_BATCH;
                if (sscanf(e, "%d,%d", &a, &b) == 2) { if (b == 0) r.flags |= GOPS_VF_STREAMED_FP32; else if (a == 0) r.flags |= GOPS_VF_STREAM_LAYER0; }
                else if (e[0] == '0') r.flags |= GOPS_VF_STREAMED_FP32;
            }
            else if (k.kind == 2) r.l2_warmup = atoi(e) + 1;
            else if (k.kind == 3) r.dw_wgs = atoi(e);
            else r.dbg_timing = true;
        }
        return r;
    }();
    return o;
}

DwPlan plan_dw(int N, int Kp, long long S, bool f16 = false, unsigned vflags = 0, int wg_target = 512) {
    DwPlan d;
    d.big = f16 || (N >= 128 && Kp >= 128);   // the half-precision GEMM has one tile size (128) and 64-sample chunks
    const int T = d.big ? 128 : 64;
    int tiles = ((N + T - 1) / T) * ((Kp + T - 1) / T);
    if (!f16 && dw_skinny_ok(N, Kp, vflags)) tiles = (N + 255) / 256;   // dw_skinny_kernel: one workgroup per 256 features and split
    const int sc = f16 ? 64 : DW_SC_HOST;
    const long long chunks = (S + sc - 1) / sc;
    if (wg_target < 1) wg_target = 512;
    long long splits = (wg_target + tiles - 1) / tiles;
    if (splits > chunks) splits = chunks;
    if (splits < 1) splits = 1;
    d.chunks_per_split = (int)((chunks + splits - 1) / splits);
    d.splits = (int)((chunks + d.chunks_per_split - 1) / d.chunks_per_split);
    return d;
}

int check_mlp(const GopsMlp& m, int in_dim, int out_dim, bool f16) {
    if (m.n_layers < 2 || m.n_layers > GOPS_MAX_LAYERS) return GOPS_ERR_UNSUPPORTED;
    if (m.sizes[0] != in_dim || m.sizes[m.n_layers] != out_dim) return GOPS_ERR_BAD_ARG;
    if (out_dim < 1 || out_dim > GOPS_MAX_ACT) return GOPS_ERR_UNSUPPORTED;
    for (int j = 1; j < m.n_layers; ++j)
        if (m.sizes[j] < 16 || (m.sizes[j] & (f16 ? 63 : 15))) return GOPS_ERR_UNSUPPORTED;
    if (m.hidden_act < GOPS_ACT_LINEAR || m.hidden_act > GOPS_ACT_TANH) return GOPS_ERR_BAD_ARG;
    for (int j = 0; j < m.n_layers; ++j)
        if (m.weight[j] == nullptr || m.bias[j] == nullptr) return GOPS_ERR_BAD_ARG;
    return GOPS_OK;
}

void fill_mlp(MlpDev& d, const GopsMlp& m) {
    memset(&d, 0, sizeof(d));
    d.nl = m.n_layers;
    d.act = m.hidden_act;
    for (int j = 0; j <= m.n_layers; ++j) d.dims[j] = m.sizes[j];
    for (int j = 0; j < m.n_layers; ++j) {
        d.kp[j] = pad16(m.sizes[j]);
        d.kp32[j] = pad32(m.sizes[j]);
        d.w[j] = m.weight[j];
        d.b[j] = m.bias[j];
    }
}

void carve_packs(Carver& c, MlpDev& d, bool f16) {
    for (int j = 0; j < d.nl - 1; ++j) {
        if (f16) {   // half fragments: N x kp32 elements each (2 per float)
            const size_t n = ((size_t)d.dims[j + 1] * d.kp32[j] + 1) / 2;
            d.wph[j] = reinterpret_cast<const f16x8*>(c.take(n));
            d.wpth[j] = reinterpret_cast<const f16x8*>(c.take(n));
            continue;
        }
        const size_t n = (size_t)d.dims[j + 1] * d.kp[j];
        d.wp[j] = reinterpret_cast<const f32x4*>(c.take(n));
        d.wpt[j] = reinterpret_cast<const f32x4*>(c.take(n));
    }
}

struct Plan {
    RolloutParams p;
    RolloutParams* dev_params = nullptr;   // device copy read by the rollout kernels
    float* dw_part[GOPS_MAX_LAYERS] = {};     // split-K partial slabs, one region per Linear layer
    float* dw_part_b[GOPS_MAX_LAYERS] = {};
    float* dummy = nullptr;                   // open loop: stand-in policy weights
    size_t dummy_floats = 0;
    size_t bytes = 0;
};

// Builds kernel parameters and carves the workspace.  ws == nullptr: size query only.
int build_plan(const GopsRolloutDesc& desc, void* ws, Plan& plan) {
    RolloutParams& p = plan.p;
    memset(&p, 0, sizeof(p));
    const GopsEnv& e = desc.env;
    if (desc.batch < 1 || desc.horizon < 1 || desc.horizon > GOPS_MAX_HORIZON) return GOPS_ERR_BAD_ARG;
    if (e.kind < GOPS_ENV_NONE || e.kind > GOPS_ENV_MOBILEROBOT) return GOPS_ERR_BAD_ARG;
    if (e.obs_dim < 1 || e.data_env) return GOPS_ERR_BAD_ARG;   // data-env semantics exist for gops_env_step only
    const int pol_out = (e.kind == GOPS_ENV_NONE) ? 1 : e.act_dim;
    if (desc.dtype != GOPS_DTYPE_F32 && desc.dtype != GOPS_DTYPE_F16) return GOPS_ERR_BAD_ARG;
    const bool f16 = desc.dtype == GOPS_DTYPE_F16;
    int rc = GOPS_OK;
    if (desc.open_loop) {   // no policy inside the rollout (FHADP2)
        if (e.kind == GOPS_ENV_NONE || desc.tail_value || desc.finite_horizon || f16) return GOPS_ERR_BAD_ARG;
    } else if ((rc = check_mlp(desc.policy, e.obs_dim + (desc.finite_horizon ? 1 : 0), pol_out, f16)) != GOPS_OK) {
        return rc;
    }
    if (desc.tail_value && (rc = check_mlp(desc.value, e.obs_dim, 1, f16)) != GOPS_OK) return rc;
    if (e.kind == GOPS_ENV_NONE && (desc.horizon != 1 || desc.tail_value || desc.finite_horizon)) return GOPS_ERR_BAD_ARG;
    if (e.scale_obs && e.kind != GOPS_ENV_LQ && e.kind != GOPS_ENV_IDPENDULUM && e.kind < GOPS_ENV_CARTPOLE) return GOPS_ERR_UNSUPPORTED;   // obs_dim <= 8 only
    if (e.kind == GOPS_ENV_CARTPOLE && (e.obs_dim != 4 || e.act_dim != 1)) return GOPS_ERR_BAD_ARG;
    if (e.kind == GOPS_ENV_PENDULUM && (e.obs_dim != 3 || e.act_dim != 1)) return GOPS_ERR_BAD_ARG;
    if (e.kind == GOPS_ENV_VEH2DOF && (e.act_dim != 1 || e.pre_horizon < 1 || e.obs_dim != 4 + e.pre_horizon || e.clip_obs ||
                                        (e.cstr_err && e.n_constraint != 1))) return GOPS_ERR_BAD_ARG;
    if (e.kind == GOPS_ENV_MOBILEROBOT && (e.obs_dim != MOB_OBS || e.act_dim != 2 || e.n_constraint != 1 || e.scale_obs)) return GOPS_ERR_BAD_ARG;
    if (e.kind >= GOPS_ENV_CARTPOLE && f16) return GOPS_ERR_UNSUPPORTED;   // the half-precision kernels are built for the BASELINE envs
    if (e.kind == GOPS_ENV_LQ && (e.obs_dim > GOPS_MAX_LQ_STATE || e.act_dim > GOPS_MAX_ACT)) return GOPS_ERR_UNSUPPORTED;
    if (e.kind == GOPS_ENV_IDPENDULUM && (e.obs_dim != 6 || e.act_dim != 1 || e.clip_obs)) return GOPS_ERR_BAD_ARG;
    if (e.kind == GOPS_ENV_VEH3DOFCONTI &&
        (e.act_dim != 2 || e.pre_horizon < 1 || e.obs_dim != 6 + 4 * e.pre_horizon || e.clip_obs))
        return GOPS_ERR_BAD_ARG;
    if (e.kind == GOPS_ENV_VEH3DOF_SURR &&
        (e.act_dim != 2 || e.pre_horizon < 1 || e.n_surr < (e.cstr_err ? 0 : 1) || e.n_surr > (e.cstr_err ? 0 : GOPS_MAX_SURR) ||
         (e.cstr_err ? e.n_constraint != 2 : (e.n_constraint != 1 && e.n_constraint != 3)) ||
         e.obs_dim != 6 + 4 * e.pre_horizon + 4 * e.n_surr || e.clip_obs ||
         (e.surr_penalty && (e.n_surr != 1 || e.n_constraint != 1)) ||
         desc.open_loop == 1 || desc.dtype != GOPS_DTYPE_F32))   // (open_loop 2: OptController's raw-action rollouts)
        return GOPS_ERR_BAD_ARG;
    if (e.clip_obs && e.obs_dim > (e.kind == GOPS_ENV_MOBILEROBOT ? GOPS_MAX_CLIP_OBS : 8)) return GOPS_ERR_UNSUPPORTED;
    if (e.repeat_num < 0 || e.repeat_num > GOPS_MAX_REPEAT) return GOPS_ERR_BAD_ARG;
    if (e.repeat_num > 1 && (f16 || (e.kind != GOPS_ENV_LQ && e.kind != GOPS_ENV_IDPENDULUM && e.kind != GOPS_ENV_CARTPOLE &&
                                     e.kind != GOPS_ENV_PENDULUM)))
        return GOPS_ERR_UNSUPPORTED;   // ActionRepeatModel: obs == state models, fp32

    p.B = desc.batch;
    p.H = desc.horizon;
    p.fh = desc.finite_horizon ? 1 : 0;
    p.need_grad = desc.need_grad ? 1 : 0;
    p.tail = desc.tail_value ? 1 : 0;
    p.tail_unmasked = (desc.tail_value && desc.tail_unmasked) ? 1 : 0;
    p.env = e;
    lq_pad_env(p.env);   // (env_models.h: the LQ matrices with compile-time strides)
    fill_ref_defaults(p.env);
    p.open_loop = desc.open_loop == 2 ? 2 : (desc.open_loop ? 1 : 0);
    p.f16 = f16 ? 1 : 0;
    p.vflags = desc.variant_flags | env_override().flags;
    p.dw_wgs = desc.dw_workgroups > 0 ? desc.dw_workgroups : (env_override().dw_wgs > 0 ? env_override().dw_wgs : 512);
    if (p.open_loop) {
        // The kernels keep their tile / stash bookkeeping in terms of a policy: give them the
        // smallest one (obs -> 16 -> act, weights zeroed in the workspace); its layers are never
        // evaluated, only the observation stash (env adjoints read it) is live.
        GopsMlp dummy;
        memset(&dummy, 0, sizeof(dummy));
        dummy.n_layers = 2;
        dummy.sizes[0] = e.obs_dim; dummy.sizes[1] = 16; dummy.sizes[2] = e.act_dim;
        dummy.hidden_act = GOPS_ACT_RELU;
        fill_mlp(p.pol, dummy);
    } else {
        fill_mlp(p.pol, desc.policy);
    }
    if (p.tail) fill_mlp(p.val, desc.value);
    int kp0 = p.pol.kp[0], hmax = 16;
    if (p.tail && p.val.kp[0] > kp0) kp0 = p.val.kp[0];
    for (int j = 1; j < p.pol.nl; ++j) hmax = p.pol.dims[j] > hmax ? p.pol.dims[j] : hmax;
    if (p.tail)
        for (int j = 1; j < p.val.nl; ++j) hmax = p.val.dims[j] > hmax ? p.val.dims[j] : hmax;
    // L2 warm-up of the sweep (rollout_bwd.hip warm_up): pays while a CU holds one tile (nothing else hides the HBM latency of
    // the next step's stash rows); with more tiles than CUs the co-resident workgroups hide it, and the warm-up lines are
    // evicted before their use - measured at cfg5 (4096 tiles): 3.8 GB fetched per sweep with it, 1.55 GB without, 1.39 -> 1.24 ms
    p.touch_mode = ((p.B + TB - 1) / TB > split_grid_limit()) ? 0 : 2;
    if (const int lw = desc.l2_warmup > 0 ? desc.l2_warmup : env_override().l2_warmup) p.touch_mode = lw - 1;   // tuning knob
    p.ldx = kp0 + 4;
    p.ldh = hmax + 4;
    const bool veh = env_has_ref_table(e.kind);
    const int ref_pts = veh ? e.pre_horizon + 1 + desc.horizon
                                                          : (e.kind == GOPS_ENV_IDPENDULUM ? IDP_POINTS(false) : 0);
    if (rollout_fwd_lds_bytes(p.ldx, p.ldh, ref_pts, f16, 0) > 160 * 1024 || rollout_bwd_lds_bytes(p.ldx, p.ldh, ref_pts, f16, false) > 160 * 1024)
        return GOPS_ERR_UNSUPPORTED;
    p.sp.on = split_eligible(p) ? 1 : 0;
    p.h64 = h64_eligible(p) ? 1 : 0;   // half precision: 64-trajectory tiles (stash rows in 64-row tiles)
    const int tile_rows = p.h64 ? 64 : TB;
    for (int t = 0; t <= p.H; ++t) p.gpow[t] = (float)pow(desc.gamma, (double)t);

    Carver c(ws);
    plan.dev_params = reinterpret_cast<RolloutParams*>(c.take((sizeof(RolloutParams) + 3) / 4));
    if (p.open_loop) {   // zeroed stand-in weights (run_forward clears them)
        plan.dummy_floats = (size_t)16 * p.pol.dims[0] + 16 + (size_t)p.pol.dims[2] * 16 + p.pol.dims[2];
        float* z = c.take(plan.dummy_floats);
        plan.dummy = z;
        p.pol.w[0] = z; p.pol.b[0] = z + (size_t)16 * p.pol.dims[0];
        p.pol.w[1] = p.pol.b[0] + 16; p.pol.b[1] = p.pol.w[1] + (size_t)p.pol.dims[2] * 16;
    }
    carve_packs(c, p.pol, f16);
    if (p.tail) carve_packs(c, p.val, f16);
    if (p.sp.on) {   // plane-split operands (2 bytes per element and plane) + one scale per n-tile
        SplitDev& sp = p.sp;
        const int n1 = p.pol.dims[1], n2 = p.pol.dims[2];
        sp.kc[0] = p.pol.kp32[0] >> 5;
        sp.kc[1] = n1 >> 5;
        auto take_pair = [&](size_t elems, const bf16x8*& w1, const f16x8*& r) {
            w1 = reinterpret_cast<const bf16x8*>(c.take((elems + 1) / 2));
            r = reinterpret_cast<const f16x8*>(c.take((elems + 1) / 2));
        };
        take_pair((size_t)n1 * 32 * sp.kc[0], sp.w1[0], sp.r[0]);
        take_pair((size_t)n2 * 32 * sp.kc[1], sp.w1[1], sp.r[1]);
        take_pair((size_t)n1 * n2, sp.w1t[1], sp.rt[1]);                 // tiles over the n1 inputs of layer 1, slots over its n2 outputs
        take_pair((size_t)p.pol.kp[0] * n1, sp.w1t[0], sp.rt[0]);        // tiles over the (16-padded) inputs of layer 0, slots over its n1 outputs
        sp.inv[0] = c.take(n1 >> 4);
        sp.inv[1] = c.take(n2 >> 4);
        sp.invt[1] = c.take(n1 >> 4);
        sp.invt[0] = c.take(p.pol.kp[0] >> 4);
    }
    p.ss = (!p.sp.on && ss_eligible(p)) ? 1 : 0;
    p.tail_fp32 = (p.ss && ss_tail_exact(p)) ? 1 : 0;
    if (p.ss) {   // streamed-split forward: planes of every hidden layer of the policy (and of the tail value net)
        for (int m = 0; m < (p.tail ? 2 : 1); ++m) {
            const MlpDev& d = m ? p.val : p.pol;
            SplitNetDev& sn = m ? p.ssv : p.ssp;
            for (int j = 0; j < d.nl - 1; ++j) {
                sn.kc[j] = (j == 0) ? ss_kc0(d.kp32[0]) : d.dims[j] >> 5;
                const size_t elems = (size_t)d.dims[j + 1] * 32 * sn.kc[j];
                sn.w1[j] = reinterpret_cast<const bf16x8*>(c.take((elems + 1) / 2));
                sn.r[j] = reinterpret_cast<const f16x8*>(c.take((elems + 1) / 2));
                sn.inv[j] = c.take(d.dims[j + 1] >> 4);
            }
        }
    }
    // the sweep of the same launch on the streamed-split kernel too: transposed planes (n-tiles over a layer's inputs)
    p.ssb = (p.need_grad && ssb_eligible(p)) ? 1 : 0;
    if (p.ssb) {
        for (int m = 0; m < (p.tail ? 2 : 1); ++m) {
            const MlpDev& d = m ? p.val : p.pol;
            SplitNetDev& sn = m ? p.ssvt : p.sspt;
            for (int j = 0; j < d.nl - 1; ++j) {
                const int nin = (j == 0) ? d.kp[0] : d.dims[j];   // (16-padded) inputs of the layer = columns of the transposed operand
                sn.kc[j] = d.dims[j + 1] >> 5;
                const size_t elems = (size_t)nin * d.dims[j + 1];
                sn.w1[j] = reinterpret_cast<const bf16x8*>(c.take((elems + 1) / 2));
                sn.r[j] = reinterpret_cast<const f16x8*>(c.take((elems + 1) / 2));
                sn.inv[j] = c.take(nin >> 4);
            }
        }
    }
    // narrow nets on the plain streamed fp32 kernels (neither plane-split nor register-stationary nor half): the policy's packed
    // hidden-layer weights live in LDS for the whole launch (common.h gemm_layer_lds) - the shapes of the reference's example scripts
    p.narrow = 0;
    if (!f16 && !p.sp.on && !p.ss && !p.ssb && !p.h64 && !p.open_loop && !(p.vflags & GOPS_VF_NO_NARROW_LDS)) {
        int skf[2], skb[2], nf = 0;
        rollout_variant(p, skf, false);
        rollout_variant(p, skb, true);
        for (int j = 0; j < p.pol.nl - 1; ++j) nf += p.pol.kp[j] * p.pol.dims[j + 1];
        const size_t lf = (rollout_fwd_lds_bytes(p.ldx, p.ldh, veh ? ref_pts : 0, false, 0) + 15) & ~(size_t)15;
        const size_t lb = (rollout_bwd_lds_bytes(p.ldx, p.ldh, ref_pts, false, false) + 15) & ~(size_t)15;
        if (skf[0] == 0 && skf[1] == 0 && skb[0] == 0 && skb[1] == 0 && nf > 0 && nf <= NARROW_MAX_FLOATS && (nf & 3) == 0 &&
            std::max(lf, lb) + 4 * (size_t)nf <= 52 * 1024) {   // three workgroups per CU stay resident
            p.narrow = 1;
            // obs -> 64 -> 64 -> act (<= 64 padded inputs): the kernels' form with these shapes as compile-time constants
            if (p.pol.nl == 3 && p.pol.dims[1] == 64 && p.pol.dims[2] == 64 && p.pol.kp[0] <= 64 && !(p.vflags & GOPS_VF_NO_NARROW_N64)) p.narrow = 2;
            p.narrow_floats = nf;
            p.narrow_off_fwd = (int)(lf / 4);
            p.narrow_off_bwd = (int)(lb / 4);
        }
    }
    p.gscale = c.take(8);   // [4 ..]: the fused Adam step's scalar factors (adam_snapshot); [0 .. 1]: max|grad_v| / max|delta_y| of a backward launch (f16 sweep scale; delta scale of the weight-gradient GEMM)
    // stash rows: every tile stores all 16 rows of every step (tile-major order)
    const long long S = (long long)((p.B + tile_rows - 1) / tile_rows) * tile_rows * p.H;
    if (veh) p.ref_table = c.take((size_t)p.B * (e.pre_horizon + 1 + p.H) * 4);
    if (e.kind == GOPS_ENV_VEH3DOF_SURR)
        p.surr_table = reinterpret_cast<const f32x4*>(c.take((size_t)p.B * (p.H + 1) * e.n_surr * 4));
    if (p.need_grad) {
        const bool gelu = p.pol.act == GOPS_ACT_GELU;
        const size_t el = f16 ? 2 : 1;   // stash elements per float of workspace (half: 2)
        p.st.x = c.take(((size_t)S * (f16 ? p.pol.kp32[0] : p.pol.kp[0]) + el - 1) / el);
        if (f16) p.st.xf = c.take((size_t)S * 8);
        for (int j = 1; j < p.pol.nl; ++j) {
            p.st.h[j] = c.take(((size_t)S * p.pol.dims[j] + el - 1) / el);
            p.st.d[j] = c.take(((size_t)S * p.pol.dims[j] + el - 1) / el);
            if (gelu) p.st.z[j] = c.take(((size_t)S * p.pol.dims[j] + el - 1) / el);
        }
        p.st.dy = c.take((size_t)S * 4);
        p.st.env = c.take((size_t)S * ENV_STASH);
        if (e.kind == GOPS_ENV_IDPENDULUM && e.repeat_num <= 1) {
            // the sub-step parking of the forward: read by the plane-split stationary sweep and by the sweeps that stage nothing
            // else (streamed fp32 - also its EXT form -, streamed-split); the exact-fp32 stationary sweep and the half kernels recompute
            bool park = p.sp.on != 0;
            if (!park && !f16) {
                int sk[2];
                rollout_variant(p, sk, true);
                park = p.ssb || sk[1] == 0;
            }
            if (park) p.st.idp = c.take((size_t)S * IDP_PARK);
        }
        if (p.tail) {
            for (int j = 1; j < p.val.nl; ++j) {
                const size_t rows = (size_t)((p.B + tile_rows - 1) / tile_rows) * tile_rows;   // whole tiles (FM stash tiles are written whole)
                p.st.tail_h[j] = c.take((rows * p.val.dims[j] + el - 1) / el);
                if (p.val.act == GOPS_ACT_GELU) p.st.tail_z[j] = c.take((rows * p.val.dims[j] + el - 1) / el);
            }
        }
        p.st.tail_done = c.take((size_t)p.B);
        // split-K partial slabs of the weight-gradient GEMMs: one region per layer so that all
        // partial sums can be reduced by a single launch at the end
        for (int j = 0; j < p.pol.nl - 1; ++j) {
            const int Kp = f16 ? p.pol.kp32[j] : p.pol.kp[j];
            const DwPlan d = plan_dw(p.pol.dims[j + 1], Kp, S, f16, p.vflags, p.dw_wgs);
            size_t nw = (size_t)d.splits * p.pol.dims[j + 1] * Kp, nb = (size_t)d.splits * p.pol.dims[j + 1];
            if (j == 0 && h64_fuses_dw0(p)) {   // one slab [256][8] / [256] per workgroup of the 64-row sweep instead (rollout_h64.hip)
                nw = std::max(nw, (size_t)h64_sweep_grid(p) * 256 * 8);
                nb = std::max(nb, (size_t)h64_sweep_grid(p) * 256);
            }
            plan.dw_part[j] = c.take(nw);
            plan.dw_part_b[j] = c.take(nb);
        }
        const int Lh = p.pol.nl - 1;
        plan.dw_part[Lh] = c.take((size_t)DW_OUT_SPLITS * GOPS_MAX_ACT * p.pol.dims[Lh]);
        plan.dw_part_b[Lh] = c.take((size_t)DW_OUT_SPLITS * GOPS_MAX_ACT);
    }
    plan.bytes = c.off + kAlign;
    return GOPS_OK;
}

#ifdef GOPS_DUMP
// debug build only (make variant V=dump VFLAGS=-DGOPS_DUMP): per-thread, per-step record buffer of the streamed-split forward
#define GOPS_DUMP_BYTES ((size_t)320 << 20)
float* gops_dump_buffer() {
    static float* buf = nullptr;
    if (buf == nullptr) { (void)hipMalloc(&buf, GOPS_DUMP_BYTES); (void)hipMemset(buf, 0, GOPS_DUMP_BYTES); }
    return buf;
}
#endif

float pdt_of(const GopsEnv& e) { return (float)((double)e.pre_horizon * 0.1); }

int run_forward(const GopsRolloutDesc& desc, const GopsRolloutIn& in, const GopsRolloutOut& out,
                void* ws, size_t ws_bytes, hipStream_t s) {
    Plan plan;
    int rc = build_plan(desc, ws, plan);
    if (rc != GOPS_OK) return rc;
    if (ws == nullptr || ws_bytes < plan.bytes) return GOPS_ERR_WORKSPACE;
    if (in.obs == nullptr || out.v_pi == nullptr) return GOPS_ERR_BAD_ARG;
    if (env_has_ref_table(desc.env.kind) &&
        (!in.state || !in.ref_points || !in.path_num || !in.u_num || !in.ref_time)) return GOPS_ERR_BAD_ARG;
    if (desc.env.kind == GOPS_ENV_VEH3DOF_SURR && desc.env.n_surr > 0 && !in.surr_state) return GOPS_ERR_BAD_ARG;
    RolloutParams& p = plan.p;
    p.in = in;
    p.out = out;
    if (p.open_loop) {
        if (in.head_pre == nullptr) return GOPS_ERR_BAD_ARG;
        hipError_t me = launch_fill_zero(plan.dummy, plan.dummy_floats, s);
        if (me != hipSuccess) return (int)me;
    }
    static unsigned long long* dbg_buf = nullptr;   // debug knob only: GOPS_DBG_TIMING=1
    const bool dbg = env_override().dbg_timing;
    if (dbg && dbg_buf == nullptr) (void)hipMalloc(&dbg_buf, 16 * sizeof(unsigned long long));
    p.dbg = dbg ? dbg_buf : nullptr;
#ifdef GOPS_DUMP
    p.dbg = reinterpret_cast<unsigned long long*>(gops_dump_buffer());
#endif
    // parameter block upload + weight packing + reference table: one launch
    hipError_t ue = launch_prologue(p, plan.dev_params, desc.env.pre_horizon, pdt_of(desc.env), s);
    if (ue != hipSuccess) return (int)ue;
    int ret;
    {
        ProfScope scope(desc.env.kind == GOPS_ENV_NONE ? 3 : 0, s);
        ret = (int)launch_rollout_fwd(p, plan.dev_params, s);
    }
    if (dbg) {
        unsigned long long h[16];
        (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
        if (p.h64)
            fprintf(stderr, "[gops dbg] fwd 64-row half cycles/step: top+sync %llu | convert+xstash+sync %llu | L0 gemm %llu epi %llu sync %llu | "
                    "L1 gemm %llu epi %llu sync %llu | head %llu sync %llu | env %llu\n",
                    h[0] / p.H, h[1] / p.H, h[2] / p.H, h[3] / p.H, h[4] / p.H, h[5] / p.H, h[6] / p.H, h[7] / p.H, h[8] / p.H, h[9] / p.H, h[10] / p.H);
        else if (p.ss)   // (grid-stride launches: the counters add up over the tiles block 0 walked)
            fprintf(stderr, "[gops dbg] fwd streamed-split cycles/step (x tiles of block 0): top+sync %llu | planes of X+sync %llu | L0 gemm %llu | "
                    "epilogues %llu | plane store+sync %llu | L1.. gemm %llu sync %llu | head partials+combine %llu | tanh+wrap %llu sync %llu | "
                    "envstash %llu | env %llu\n", h[0] / p.H, h[1] / p.H, h[14] / p.H, h[8] / p.H, h[12] / p.H, h[11] / p.H, h[9] / p.H, h[2] / p.H,
                    h[7] / p.H, h[3] / p.H, h[4] / p.H, h[5] / p.H);
        else if (p.sp.on)
            fprintf(stderr, "[gops dbg] fwd SPLIT cycles/step: top+sync %llu | xstash+convert+sync %llu | gemm0 %llu | epi0+planes %llu | sync %llu | "
                    "gemm1 %llu | epi1+head fma %llu | head reduce %llu | sync+combine %llu | tanh+wrap %llu | sync %llu | envstash %llu | env %llu || sum %llu\n",
                    h[0] / p.H, h[1] / p.H, h[14] / p.H, h[8] / p.H, h[9] / p.H, h[11] / p.H, h[12] / p.H, h[2] / p.H, h[6] / p.H, h[7] / p.H,
                    h[3] / p.H, h[4] / p.H, h[5] / p.H,
                    (h[0] + h[1] + h[14] + h[8] + h[9] + h[11] + h[12] + h[2] + h[6] + h[7] + h[3] + h[4] + h[5]) / p.H);
        if (p.sp.on) fprintf(stderr, "[gops dbg] fwd SPLIT per launch (cycles of block 0): kernel entry -> first step %llu | behind the last step %llu\n", h[13], h[15]);
        else
        fprintf(stderr, "[gops dbg] fwd cycles/step: top+sync %llu | xstash %llu | hidden-rest %llu | head %llu | envstash %llu | env %llu"
                " || L0 epi %llu sync %llu stash %llu | L1 epi %llu sync %llu stash %llu | gemm(L0+L1) %llu || head: dot %llu tanh+wrap %llu barrier %llu\n",
                h[0] / p.H, h[1] / p.H, h[2] / p.H, h[3] / p.H, h[4] / p.H, h[5] / p.H, h[8] / p.H, h[9] / p.H,
                h[10] / p.H, h[11] / p.H, h[12] / p.H, h[13] / p.H, h[14] / p.H, h[6] / p.H, h[7] / p.H, h[3] / p.H);
    }
    return ret;
}

int run_backward(const GopsRolloutDesc& desc, const GopsRolloutIn& in, const float* grad_v,
                 const GopsMlpGrad& grad, float* g_head_pre, void* ws, size_t ws_bytes, hipStream_t s,
                 const float* ext_delta = nullptr, const GopsRolloutAdjoint* adj = nullptr, bool want_params = true,
                 const GopsUpdateTail* tail = nullptr) {
    if (!desc.need_grad || grad_v == nullptr) return GOPS_ERR_BAD_ARG;
    Plan plan;
    int rc = build_plan(desc, ws, plan);
    if (rc != GOPS_OK) return rc;
    if (ws == nullptr || ws_bytes < plan.bytes) return GOPS_ERR_WORKSPACE;
    RolloutParams& p = plan.p;
    p.in = in;
    p.grad_v = grad_v;
    p.g_head_pre = g_head_pre;
    if ((p.open_loop != 0) != (g_head_pre != nullptr)) return GOPS_ERR_BAD_ARG;
    if (p.open_loop && in.head_pre == nullptr) return GOPS_ERR_BAD_ARG;
    p.ext_delta = ext_delta;   // gops_mlp_backward: the head is a stand-in, its gradient is not formed
    if (ext_delta != nullptr && desc.env.kind != GOPS_ENV_NONE) return GOPS_ERR_BAD_ARG;
    if (!p.open_loop && want_params)
        for (int j = 0; j < p.pol.nl - (ext_delta != nullptr ? 1 : 0); ++j)
            if (grad.weight[j] == nullptr || grad.bias[j] == nullptr) return GOPS_ERR_BAD_ARG;
    if (tail != nullptr) {   // gops_rollout_backward_update: checked before anything is launched
        if (p.open_loop || !want_params || ext_delta != nullptr || adj != nullptr) return GOPS_ERR_BAD_ARG;
        // half a backward (GOPS_VF_BWD_PHASE_A / _B: a data-parallel update all-reduces between gradient and optimizer step) can carry
        // the LOSS MEAN on phase A's reduce launch - it needs no gradient - but never an optimizer / Polyak step
        const unsigned ph = desc.variant_flags & (GOPS_VF_BWD_PHASE_A | GOPS_VF_BWD_PHASE_B);
        if (ph != 0 && (ph != GOPS_VF_BWD_PHASE_A || tail->adam != nullptr || tail->polyak != nullptr)) return GOPS_ERR_BAD_ARG;
        if (tail->mean_x != nullptr && (tail->mean_stats == nullptr || tail->mean_n < 1)) return GOPS_ERR_BAD_ARG;
        if (tail->adam != nullptr) {
            const GopsAdamTensors& T = *tail->adam;
            if (tail->adam_state == nullptr || T.n != 2 * p.pol.nl) return GOPS_ERR_BAD_ARG;
            for (int j = 0; j < p.pol.nl; ++j) {   // every gradient tensor of the policy has its parameter / moments in the table
                const long long nw = (long long)p.pol.dims[j + 1] * p.pol.dims[j], nb = p.pol.dims[j + 1];
                int fw = -1, fb = -1;
                for (int k = 0; k < T.n; ++k) {
                    if (T.grad[k] == grad.weight[j] && T.numel[k] == nw) fw = k;
                    if (T.grad[k] == grad.bias[j] && T.numel[k] == nb) fb = k;
                }
                if (fw < 0 || fb < 0 || !T.param[fw] || !T.exp_avg[fw] || !T.exp_avg_sq[fw] || !T.param[fb] || !T.exp_avg[fb] || !T.exp_avg_sq[fb])
                    return GOPS_ERR_BAD_ARG;
            }
        }
        if (tail->polyak != nullptr) {   // every online tensor of the averaging table is a parameter this call steps
            if (tail->adam == nullptr || tail->polyak->n != tail->adam->n) return GOPS_ERR_BAD_ARG;
            for (int k = 0; k < tail->polyak->n; ++k) {
                bool found = false;
                for (int i = 0; i < tail->adam->n; ++i)
                    found = found || (tail->polyak->grad[k] == tail->adam->param[i] && tail->polyak->numel[k] == tail->adam->numel[i]);
                if (!found || tail->polyak->param[k] == nullptr) return GOPS_ERR_BAD_ARG;
            }
        }
    }
    hipError_t e;
    if (desc.env.repeat_num > 1) p.ext = 1;   // ActionRepeatModel: the general (EXT) instantiations of the sweep
    if (adj != nullptr) {   // gops_rollout_backward_adj / gops_mlp_backward_x: the EXT kernels
        const int k = desc.env.kind;
        if (p.tail || p.f16) return GOPS_ERR_UNSUPPORTED;   // (open loop: gops_rollout_backward_open_loop_adj)
        if (k != GOPS_ENV_NONE && k != GOPS_ENV_LQ && k != GOPS_ENV_IDPENDULUM && k != GOPS_ENV_CARTPOLE && k != GOPS_ENV_PENDULUM &&
            k != GOPS_ENV_MOBILEROBOT)
            return GOPS_ERR_UNSUPPORTED;
        p.ext = 1;
        p.adj_gfo = adj->grad_final_obs;
        p.adj_gobs = adj->grad_obs;
        p.adj_first_only = (adj->first_step_only != 0 && p.H > 1) ? 1 : 0;
        if (p.adj_first_only && want_params) {   // the sweep writes the step-0 deltas only: the rest of the stash is zero
            const size_t S0 = (size_t)((p.B + TB - 1) / TB) * TB * p.H;
            for (int j = 1; j < p.pol.nl; ++j)
                if ((e = launch_fill_zero(p.st.d[j], S0 * p.pol.dims[j], s)) != hipSuccess) return (int)e;
            if ((e = launch_fill_zero(p.st.dy, S0 * 4, s)) != hipSuccess) return (int)e;
        }
    }
    static unsigned long long* dbg_buf = nullptr;   // debug knob only: GOPS_DBG_TIMING=1
    const bool dbg = env_override().dbg_timing;
    if (dbg && dbg_buf == nullptr) (void)hipMalloc(&dbg_buf, 16 * sizeof(unsigned long long));
    p.dbg = dbg ? dbg_buf : nullptr;
    // Split sweep: the output layer's weight gradient is accumulated inside the sweep (one partial per workgroup) - no dw_out
    // pass.  (GELU: the sweep's act' operand is gelu'(z), so it fetches H_2 next to it.)
    // (the streamed-split sweep does the same for the env kinds whose instantiation has the registers: ssb_fuses_out)
    const bool fused_out = (p.sp.on || ssb_fuses_out(p)) && !p.ext && !p.open_loop && want_params &&
                           ext_delta == nullptr && !(p.vflags & GOPS_VF_NO_FUSED_DWOUT);
    const int sweep_grid = std::min((p.B + TB - 1) / TB, p.sp.on ? split_grid_limit() : ssb_grid_limit());
    if (fused_out) {
        p.sp.out_part = plan.dw_part[p.pol.nl - 1];
        p.sp.out_part_b = plan.dw_part_b[p.pol.nl - 1];
    }
    // What this call adds to the forward's device copy of the parameter block travels as a kernel argument (BwdPatch);
    // only the half sweep, which needs max|grad_v| before it starts, still takes the upload launch.
    BwdPatch q;
    memset(&q, 0, sizeof(q));
    q.in = in;
    q.grad_v = grad_v;
    q.g_head_pre = g_head_pre;
    q.ext_delta = ext_delta;
    q.adj_gfo = p.adj_gfo;
    q.adj_gobs = p.adj_gobs;
    q.adj_first_only = p.adj_first_only;
    q.out_part = p.sp.out_part;
    q.out_part_b = p.sp.out_part_b;
    q.dbg = p.dbg;
    const bool fuse_dw0 = h64_fuses_dw0(p) && want_params && ext_delta == nullptr && adj == nullptr &&
                          !(p.vflags & (GOPS_VF_BWD_PHASE_A | GOPS_VF_BWD_PHASE_B));
    if (fuse_dw0) {
        q.w0_part = plan.dw_part[0];
        q.w0_part_b = plan.dw_part_b[0];
    }
    if (tail != nullptr && tail->adam != nullptr) {   // the sweep's thread 0 advances the optimizer state and leaves this step's factors
        q.ad_st = tail->adam_state;
        q.ad_snap = p.gscale + 4;
        q.ad_b1 = tail->beta1; q.ad_b2 = tail->beta2;
    }
    const bool force_upload = (p.vflags & GOPS_VF_BWD_UPLOAD) != 0;   // measurement knob: the pre-patch launch sequence
    // GOPS_VF_BWD_PHASE_A / _B: the call is one half of a backward (see gops_hip.h); only_b skips the sweep
    const unsigned phase = p.vflags & (GOPS_VF_BWD_PHASE_A | GOPS_VF_BWD_PHASE_B);
    const bool only_a = phase == GOPS_VF_BWD_PHASE_A, only_b = phase == GOPS_VF_BWD_PHASE_B;
    if ((only_a || only_b) && (p.open_loop || !want_params || ext_delta != nullptr || adj != nullptr)) return GOPS_ERR_UNSUPPORTED;
    if (!only_b) {
    // max|grad_v| belongs to THIS backward call: the sweep (fp32) / the upload kernel (half) only ever raise gscale[0], so a
    // second backward after the same forward with a much smaller grad_v would inherit the larger scale and push its scaled
    // deltas into half subnormals (advisor finding, round 3).  Invariant: gscale[0] is ZERO when a backward call starts - the
    // forward's prologue zeroes it, and every backward call leaves it zero behind its last reader: fp32 calls that end with the
    // split-K reduce reset it there (ReduceJobs.reset: no extra launch - round 5; it was a 1-block fill in front of every sweep),
    // the others (half precision: the reduce itself reads it; open loop / no parameter gradients: no reduce) with a fill at their end.
    if ((p.f16 || force_upload) && (e = launch_upload_params(p, plan.dev_params, s)) != hipSuccess) return (int)e;
    {
        ProfScope scope(desc.env.kind == GOPS_ENV_NONE ? 4 : 1, s);
        if ((e = launch_rollout_bwd(p, plan.dev_params, q, s)) != hipSuccess) return (int)e;
    }
    }   // !only_b
    if (dbg) {
        unsigned long long h[16];
        (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
        if (p.h64)
            fprintf(stderr, "[gops dbg] bwd 64-row half cycles/step: env adjoint %llu sync %llu | head %llu sync %llu | gemm %llu sync %llu epi %llu sync %llu | "
                    "g_x %llu | end sync %llu\n", h[0] / p.H, h[1] / p.H, h[2] / p.H, h[3] / p.H, h[4] / p.H, h[5] / p.H, h[6] / p.H, h[7] / p.H, h[8] / p.H,
                    h[9] / p.H);
        else if (p.sp.on)
            fprintf(stderr, "[gops dbg] bwd SPLIT cycles/step: top %llu | env points %llu | reduce+sync %llu | env finish+sync %llu | head delta+planes %llu | "
                    "sync %llu | hook %llu | gemm1+epi+planes %llu | sync %llu | gemm0+G %llu | end sync %llu || sum %llu\n",
                    h[0] / p.H, h[10] / p.H, h[11] / p.H, h[1] / p.H, h[3] / p.H, h[4] / p.H, h[5] / p.H, h[6] / p.H, (h[7] + h[8]) / p.H, h[9] / p.H,
                    h[2] / p.H, (h[0] + h[10] + h[11] + h[1] + h[3] + h[4] + h[5] + h[6] + h[7] + h[8] + h[9] + h[2]) / p.H);
        else
        fprintf(stderr, "[gops dbg] bwd cycles/step: touch %llu | env(points %llu, reduce+sync %llu, finish %llu) | head-bwd %llu sync %llu stash %llu | "
                "gemm1+epi %llu sync %llu stash %llu | gemm0+epi %llu | end sync %llu\n",
                h[0] / p.H, h[10] / p.H, h[11] / p.H, h[1] / p.H, h[3] / p.H, h[4] / p.H, h[5] / p.H, h[6] / p.H, h[7] / p.H,
                h[8] / p.H, h[9] / p.H, h[2] / p.H);
    }
    if (p.open_loop || !want_params)   // no parameters behind the rollout / none wanted: no reduce - gscale is reset here
        return (int)launch_fill_zero(p.gscale, 4, s);
    ProfScope scope(desc.env.kind == GOPS_ENV_NONE ? 5 : 2, s);
    const int tile_rows = p.h64 ? 64 : TB;
    const long long S = (long long)((p.B + tile_rows - 1) / tile_rows) * tile_rows * p.H;
    const int L = p.pol.nl - 1;
    ReduceJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    if (p.f16) jobs.unscale = p.gscale;   // the sweep ran on gradients scaled by f16_grad_scale(max|grad_v|)
    else jobs.poison = reinterpret_cast<const unsigned*>(p.gscale) + 3;   // (set by a plane-split forward / sweep whose conversions overflowed)
    // Two-half-plane weight-gradient GEMM (launch_dw_gemm): its delta scale is derived from max|grad_v|, which bounds the
    // deltas only when grad_v is the sweep's one gradient source - a terminal observation adjoint or constraint-sum
    // gradients can be orders of magnitude larger, so those launches keep the exact three-plane product.
    const float* dw_scale = (adj == nullptr && in.grad_constraint == nullptr && in.grad_constraint_prod == nullptr && in.grad_constraint_step == nullptr &&
                             ext_delta == nullptr)
                                ? p.gscale : nullptr;
    for (int j = 0; j < L; ++j) {   // dW_j = D_{j+1}^T * (j == 0 ? X : H_j)
        if ((only_a && j == 0) || (only_b && j != 0)) continue;   // (two-phase backward: layer 0 is phase B)
        const int N = p.pol.dims[j + 1], Kp = p.f16 ? p.pol.kp32[j] : p.pol.kp[j], K = p.pol.dims[j];
        const DwPlan d = plan_dw(N, Kp, S, p.f16 != 0, p.vflags, p.dw_wgs);
        const float* X = (j == 0) ? p.st.x : p.st.h[j];
        if (j == 0 && fuse_dw0) {   // formed inside the 64-row sweep: one slab per workgroup
            reduce_jobs_add(jobs, plan.dw_part[0], h64_sweep_grid(p), N, K, 8, grad.weight[0]);
            reduce_jobs_add(jobs, plan.dw_part_b[0], h64_sweep_grid(p), 1, N, N, grad.bias[0]);
            continue;
        }
        if (p.f16) {
            if ((e = launch_dw_gemm_f16(p.st.d[j + 1], N, X, Kp, S, d.splits, d.chunks_per_split, plan.dw_part[j],
                                        plan.dw_part_b[j], s)) != hipSuccess) return (int)e;
        } else
        if ((e = launch_dw_gemm(p.st.d[j + 1], N, X, Kp, S, d.splits, d.chunks_per_split, plan.dw_part[j],
                                plan.dw_part_b[j], d.big, s, dw_scale, p.vflags)) != hipSuccess) return (int)e;
        reduce_jobs_add(jobs, plan.dw_part[j], d.splits, N, K, Kp, grad.weight[j]);
        reduce_jobs_add(jobs, plan.dw_part_b[j], d.splits, 1, N, N, grad.bias[j]);
    }
    if (ext_delta == nullptr && !only_b) {
        const int K = p.pol.dims[L], A = p.pol.dims[p.pol.nl];
        long long splits = fused_out ? sweep_grid : DW_OUT_SPLITS;
        if (splits > S) splits = S;
        if (!fused_out)
        if ((e = launch_dw_out(p.st.dy, p.st.h[L], p.f16 != 0, K, A, S, (int)splits, plan.dw_part[L], plan.dw_part_b[L], s)) != hipSuccess) return (int)e;
        reduce_jobs_add(jobs, plan.dw_part[L], (int)splits, A, K, K, grad.weight[L]);
        reduce_jobs_add(jobs, plan.dw_part_b[L], (int)splits, 1, A, A, grad.bias[L]);
    }
    // (two-phase backward: phase B's GEMM still reads the scale phase A's sweep left - the reset belongs to the LAST reduce of the call pair)
    if (!p.f16 && !only_a && jobs.n > 0) jobs.reset = p.gscale;
    if (tail != nullptr) {   // the update's tail rides on the reduce: Adam per gradient element, the loss mean in one more block
        if (tail->adam != nullptr) {
            const GopsAdamTensors& T = *tail->adam;
            for (int i = 0; i < jobs.n; ++i)
                for (int k = 0; k < T.n; ++k)
                    if (T.grad[k] == jobs.out[i]) { jobs.ad_p[i] = T.param[k]; jobs.ad_m[i] = T.exp_avg[k]; jobs.ad_v[i] = T.exp_avg_sq[k]; }
            jobs.ad_snap = p.gscale + 4;
            jobs.ad_skipped = &tail->adam_state->skipped_nonfinite;
            jobs.ad_b1 = tail->beta1; jobs.ad_b2 = tail->beta2; jobs.ad_eps = (float)tail->eps;
            if (tail->polyak != nullptr) {
                for (int i = 0; i < jobs.n; ++i)
                    for (int k = 0; k < tail->polyak->n; ++k)
                        if (jobs.ad_p[i] != nullptr && tail->polyak->grad[k] == jobs.ad_p[i]) jobs.pk_t[i] = tail->polyak->param[k];
                jobs.pk_omt = (float)(1.0 - tail->polyak_tau);
                jobs.pk_tau = (float)tail->polyak_tau;
            }
        }
        if (tail->mean_x != nullptr) {
            jobs.mean_x = tail->mean_x; jobs.mean_n = tail->mean_n; jobs.mean_sc = (float)tail->mean_scale; jobs.mean_stats = tail->mean_stats;
        }
    }
    if ((e = launch_reduce(jobs, s)) != hipSuccess) return (int)e;
    if ((p.f16 || jobs.n == 0) && !only_a && (e = launch_fill_zero(p.gscale, 4, s)) != hipSuccess) return (int)e;
    return GOPS_OK;
}

GopsRolloutDesc value_desc(const GopsMlp& value, int batch) {
    GopsRolloutDesc d;
    memset(&d, 0, sizeof(d));
    d.batch = batch;
    d.horizon = 1;
    d.need_grad = 1;
    d.gamma = 1.0;
    d.env.kind = GOPS_ENV_NONE;
    d.env.obs_dim = value.sizes[0];
    d.env.act_dim = 1;
    d.policy = value;
    d.dtype = value.dtype;
    d.variant_flags = value.variant_flags;   // (ABI v10: plain MLP batches carry their variant flags in GopsMlp)
    return d;
}

// ---- gops_mlp_*: hidden stack on the GOPS_ENV_NONE tiles + an output layer of any width ------------------------
struct MlpPlan {
    GopsRolloutDesc d;        // hidden stack with a 1-wide stand-in head (zero weights in the workspace)
    size_t inner = 0;         // bytes of the rollout workspace at the front
    float* zeros = nullptr;   // [K + 1] stand-in head weight + bias
    float* v = nullptr;       // [B] stand-in output
    float* gv = nullptr;      // [B] zeros: grad of the stand-in output
    float* gh = nullptr;      // [S][K] adjoint of the last hidden activation
    float* gyp = nullptr;     // [S][Wp] zero-padded copy of grad_y (delta operand of the output layer's dW GEMM)
    float* part = nullptr;    // split-K slabs of the output layer [splits][Wp][K]
    float* part_b = nullptr;  // [splits][Wp]
    int K = 0, W = 0, Wp = 0;
    long long S = 0;
    size_t bytes = 0;
};

int plan_mlp(const GopsMlp& mlp, int batch, void* ws, MlpPlan& m) {
    if (mlp.n_layers < 2 || mlp.n_layers > GOPS_MAX_LAYERS || batch < 1) return GOPS_ERR_BAD_ARG;
    if (mlp.dtype != GOPS_DTYPE_F32) return GOPS_ERR_UNSUPPORTED;
    m.W = mlp.sizes[mlp.n_layers];
    m.K = mlp.sizes[mlp.n_layers - 1];
    if (m.W < 1 || m.W > 4096 || (m.K & 15)) return GOPS_ERR_UNSUPPORTED;
    m.Wp = pad16(m.W);
    m.S = (long long)((batch + TB - 1) / TB) * TB;
    GopsMlp hidden = mlp;
    hidden.sizes[mlp.n_layers] = 1;
    m.d = value_desc(hidden, batch);
    Plan inner;
    // the stand-in head pointers only have to be non-null for the size query
    m.d.policy.weight[mlp.n_layers - 1] = reinterpret_cast<const float*>(&m);
    m.d.policy.bias[mlp.n_layers - 1] = reinterpret_cast<const float*>(&m);
    int rc = build_plan(m.d, nullptr, inner);
    if (rc != GOPS_OK) return rc;
    m.inner = (inner.bytes + kAlign - 1) / kAlign * kAlign;
    Carver c(ws ? static_cast<char*>(ws) + m.inner : nullptr);
    m.zeros = c.take((size_t)m.K + 16);
    m.v = c.take((size_t)batch);
    m.gv = c.take((size_t)batch);
    m.gh = c.take((size_t)m.S * m.K);
    m.gyp = c.take((size_t)m.S * m.Wp);
    const unsigned vf = mlp.variant_flags | env_override().flags;
    const DwPlan d = plan_dw(m.Wp, m.K, m.S, false, vf);
    m.part = c.take((size_t)d.splits * m.Wp * m.K);
    m.part_b = c.take((size_t)d.splits * m.Wp);
    m.bytes = m.inner + c.off + kAlign;
    m.d.policy.weight[mlp.n_layers - 1] = m.zeros;
    m.d.policy.bias[mlp.n_layers - 1] = m.zeros ? m.zeros + m.K : nullptr;
    return GOPS_OK;
}

}  // namespace

extern "C" {

#ifdef GOPS_DUMP
int gops_dbg_dump_read(void* host, size_t bytes) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpy(host, gops_dump_buffer(), bytes < GOPS_DUMP_BYTES ? bytes : GOPS_DUMP_BYTES, hipMemcpyDeviceToHost);
}
#endif

size_t gops_mlp_workspace_bytes(const GopsMlp* mlp, int32_t batch) {
    if (!mlp) return 0;
    MlpPlan m;
    return plan_mlp(*mlp, batch, nullptr, m) == GOPS_OK ? m.bytes : 0;
}

int gops_mlp_forward(const GopsMlp* mlp, int32_t batch, const float* x, float* y, void* workspace, size_t workspace_bytes,
                     void* stream) {
    if (!mlp || !x || !y) return GOPS_ERR_BAD_ARG;
    MlpPlan m;
    int rc = plan_mlp(*mlp, batch, workspace, m);
    if (rc != GOPS_OK) return rc;
    if (workspace == nullptr || workspace_bytes < m.bytes) return GOPS_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = launch_fill_zero(m.zeros, (size_t)m.K + 16, s);
    if (e != hipSuccess) return (int)e;
    GopsRolloutIn in;
    memset(&in, 0, sizeof(in));
    in.obs = x;
    GopsRolloutOut out;
    memset(&out, 0, sizeof(out));
    out.v_pi = m.v;
    if ((rc = run_forward(m.d, in, out, workspace, m.inner, s)) != GOPS_OK) return rc;   // stashes the hidden activations
    Plan inner;
    build_plan(m.d, workspace, inner);
    const int L = mlp->n_layers - 1;
    return (int)launch_linear_out_fwd(inner.p.st.h[L], m.K, mlp->weight[L], mlp->bias[L], m.W, batch, y, s);
}

static int mlp_backward_impl(const GopsMlp* mlp, int32_t batch, const float* x, const float* grad_y, const GopsMlpGrad* grad,
                             float* grad_x, void* workspace, size_t workspace_bytes, void* stream) {
    if (!mlp || !x || !grad_y || (!grad && !grad_x)) return GOPS_ERR_BAD_ARG;
    MlpPlan m;
    int rc = plan_mlp(*mlp, batch, workspace, m);
    if (rc != GOPS_OK) return rc;
    if (workspace == nullptr || workspace_bytes < m.bytes) return GOPS_ERR_WORKSPACE;
    const int L = mlp->n_layers - 1;
    if (grad && (!grad->weight[L] || !grad->bias[L])) return GOPS_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if ((e = launch_fill_zero(m.gv, (size_t)batch, s)) != hipSuccess) return (int)e;
    // output layer: g_h = g_y Wo (+ padded copy of g_y), dWo = g_y^T h and db = column sums through the regular dW GEMM
    if ((e = launch_linear_out_bwd(grad_y, m.W, m.Wp, mlp->weight[L], m.K, batch, m.S, m.gh, m.gyp, s)) != hipSuccess) return (int)e;
    Plan inner;
    build_plan(m.d, workspace, inner);
    if (grad) {
        const unsigned vf = mlp->variant_flags | env_override().flags;
        const DwPlan d = plan_dw(m.Wp, m.K, m.S, false, vf);
        if ((e = launch_dw_gemm(m.gyp, m.Wp, inner.p.st.h[L], m.K, m.S, d.splits, d.chunks_per_split, m.part, m.part_b, d.big, s, nullptr, vf)) != hipSuccess)
            return (int)e;
        ReduceJobs jobs;
        memset(&jobs, 0, sizeof(jobs));
        reduce_jobs_add(jobs, m.part, d.splits, m.W, m.K, m.K, grad->weight[L], m.Wp);
        reduce_jobs_add(jobs, m.part_b, d.splits, 1, m.W, m.Wp, grad->bias[L]);
        if ((e = launch_reduce(jobs, s)) != hipSuccess) return (int)e;
    }
    // hidden stack: the sweep starts from g_h instead of a head
    GopsRolloutIn in;
    memset(&in, 0, sizeof(in));
    in.obs = x;
    GopsMlpGrad none;
    memset(&none, 0, sizeof(none));
    GopsRolloutAdjoint adj;
    memset(&adj, 0, sizeof(adj));
    adj.grad_obs = grad_x;
    return run_backward(m.d, in, m.gv, grad ? *grad : none, nullptr, workspace, m.inner, s, m.gh, grad_x ? &adj : nullptr,
                        grad != nullptr);
}

int gops_mlp_backward(const GopsMlp* mlp, int32_t batch, const float* x, const float* grad_y, const GopsMlpGrad* grad,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad) return GOPS_ERR_BAD_ARG;
    return mlp_backward_impl(mlp, batch, x, grad_y, grad, nullptr, workspace, workspace_bytes, stream);
}

int gops_mlp_backward_x(const GopsMlp* mlp, int32_t batch, const float* x, const float* grad_y, const GopsMlpGrad* grad,
                        float* grad_x, void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad_x) return GOPS_ERR_BAD_ARG;
    return mlp_backward_impl(mlp, batch, x, grad_y, grad, grad_x, workspace, workspace_bytes, stream);
}

int gops_hip_version(void) { return GOPS_HIP_ABI_VERSION; }

size_t gops_rollout_workspace_bytes(const GopsRolloutDesc* desc) {
    if (desc == nullptr) return 0;
    Plan plan;
    if (build_plan(*desc, nullptr, plan) != GOPS_OK) return 0;
    return plan.bytes;
}

int gops_rollout_variant(const GopsRolloutDesc* desc) {
    if (desc == nullptr) return GOPS_ERR_BAD_ARG;
    Plan plan;
    const int rc = build_plan(*desc, nullptr, plan);
    if (rc != GOPS_OK) return rc;
    if (plan.p.sp.on) return GOPS_VARIANT_SPLIT;
    if (plan.p.ss) return GOPS_VARIANT_STREAMED_SPLIT_FWD;
    if (plan.p.h64) return GOPS_VARIANT_HALF_TILE64;
    int sk[2];
    rollout_variant(plan.p, sk, false);
    return sk[1] > 0 ? GOPS_VARIANT_STATIONARY_F32 : 0;
}

int gops_rollout_forward(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const GopsRolloutOut* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !in || !out) return GOPS_ERR_BAD_ARG;
    return run_forward(*desc, *in, *out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int gops_rollout_backward(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                          const GopsMlpGrad* policy_grad, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (!desc || !in || !policy_grad) return GOPS_ERR_BAD_ARG;
    return run_backward(*desc, *in, grad_v, *policy_grad, nullptr, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream));
}

int gops_rollout_backward_update(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                                 const GopsMlpGrad* policy_grad, const GopsUpdateTail* tail, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!desc || !in || !policy_grad || !tail) return GOPS_ERR_BAD_ARG;
    return run_backward(*desc, *in, grad_v, *policy_grad, nullptr, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream), nullptr, nullptr, true, tail);
}

int gops_rollout_backward_adj(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                              const GopsMlpGrad* policy_grad, const GopsRolloutAdjoint* adj, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!desc || !in || !adj) return GOPS_ERR_BAD_ARG;
    GopsMlpGrad none;
    memset(&none, 0, sizeof(none));
    return run_backward(*desc, *in, grad_v, policy_grad ? *policy_grad : none, nullptr, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream), nullptr, adj, policy_grad != nullptr);
}

int gops_rollout_backward_open_loop(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                                    float* grad_head_pre, void* workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !in || !grad_head_pre || !desc->open_loop) return GOPS_ERR_BAD_ARG;
    GopsMlpGrad none;
    memset(&none, 0, sizeof(none));
    return run_backward(*desc, *in, grad_v, none, grad_head_pre, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream));
}

int gops_rollout_backward_open_loop_adj(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                                        float* grad_head_pre, const GopsRolloutAdjoint* adj, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (!desc || !in || !grad_head_pre || !adj || !desc->open_loop) return GOPS_ERR_BAD_ARG;
    GopsMlpGrad none;
    memset(&none, 0, sizeof(none));
    GopsRolloutAdjoint a = *adj;
    a.first_step_only = 0;
    return run_backward(*desc, *in, grad_v, none, grad_head_pre, workspace, workspace_bytes, static_cast<hipStream_t>(stream),
                        nullptr, &a, false);
}

int gops_env_step(const GopsEnv* env, int32_t batch, const GopsStepIO* io, void* stream) {
    if (!env || !io || batch < 1) return GOPS_ERR_BAD_ARG;
    if (env->kind < GOPS_ENV_LQ || env->kind > GOPS_ENV_MOBILEROBOT) return GOPS_ERR_BAD_ARG;
    if (env->kind == GOPS_ENV_MOBILEROBOT &&
        (env->obs_dim != MOB_OBS || env->act_dim != 2 || env->n_constraint != 1 || env->scale_obs || !io->constraint)) return GOPS_ERR_BAD_ARG;
    if (env->data_env && (env->kind == GOPS_ENV_PENDULUM || (env->kind == GOPS_ENV_VEH2DOF && env->cstr_err)))
        return GOPS_ERR_UNSUPPORTED;   // these data envs are not restated
    if (!io->obs || !io->action || !io->next_obs || !io->reward || !io->next_done) return GOPS_ERR_BAD_ARG;
    if (env->kind == GOPS_ENV_VEH3DOF_SURR &&
        (env->data_env || env->n_surr < (env->cstr_err ? 0 : 1) || env->n_surr > (env->cstr_err ? 0 : GOPS_MAX_SURR) ||
         (env->cstr_err ? env->n_constraint != 2 : (env->n_constraint != 1 && env->n_constraint != 3)) ||
         (env->n_surr > 0 && (!io->surr_state || !io->next_surr_state)) || !io->constraint)) return GOPS_ERR_BAD_ARG;
    if (env_has_ref_table(env->kind) &&
        (!io->state || !io->ref_points || !io->path_num || !io->u_num || !io->ref_time ||
         !io->next_state || !io->next_ref_points || !io->next_ref_time)) return GOPS_ERR_BAD_ARG;
    if (env->kind == GOPS_ENV_LQ && env->obs_dim > GOPS_MAX_LQ_STATE) return GOPS_ERR_UNSUPPORTED;
    if (env->scale_obs && env->kind != GOPS_ENV_LQ && env->kind != GOPS_ENV_IDPENDULUM && env->kind < GOPS_ENV_CARTPOLE) return GOPS_ERR_UNSUPPORTED;
    if (env->repeat_num < 0 || env->repeat_num > GOPS_MAX_REPEAT) return GOPS_ERR_BAD_ARG;
    if (env->repeat_num > 1 && (env->data_env || (env->kind != GOPS_ENV_LQ && env->kind != GOPS_ENV_IDPENDULUM &&
                                                  env->kind != GOPS_ENV_CARTPOLE && env->kind != GOPS_ENV_PENDULUM)))
        return GOPS_ERR_UNSUPPORTED;
    GopsEnv e = *env;
    fill_ref_defaults(e);
    return (int)launch_env_step(e, batch, *io, pdt_of(e), static_cast<hipStream_t>(stream));
}

int gops_env_constraint(const GopsEnv* env, int32_t batch, const GopsStepIO* io, void* stream) {
    if (!env || !io || batch < 1 || !io->constraint) return GOPS_ERR_BAD_ARG;
    const bool err = env->cstr_err != 0 && (env->kind == GOPS_ENV_VEH3DOF_SURR || env->kind == GOPS_ENV_VEH2DOF);
    const bool surr = env->kind == GOPS_ENV_VEH3DOF_SURR && !env->cstr_err;
    if (!err && !surr) return GOPS_ERR_UNSUPPORTED;   // the model defines no get_constraint
    if (err && (!io->obs || env->n_constraint != (env->kind == GOPS_ENV_VEH2DOF ? 1 : 2))) return GOPS_ERR_BAD_ARG;
    if (surr && (!io->state || !io->surr_state || env->n_surr < 1 || env->n_surr > GOPS_MAX_SURR ||
                 (env->n_constraint != 1 && env->n_constraint != 3))) return GOPS_ERR_BAD_ARG;
    return (int)launch_env_constraint(*env, batch, *io, static_cast<hipStream_t>(stream));
}

size_t gops_value_workspace_bytes(const GopsMlp* value, int32_t batch) {
    if (!value) return 0;
    const GopsRolloutDesc d = value_desc(*value, batch);
    return gops_rollout_workspace_bytes(&d);
}

int gops_value_forward(const GopsMlp* value, int32_t batch, const float* obs, float* v, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!value || !obs || !v) return GOPS_ERR_BAD_ARG;
    const GopsRolloutDesc d = value_desc(*value, batch);
    GopsRolloutIn in;
    memset(&in, 0, sizeof(in));
    in.obs = obs;
    GopsRolloutOut out;
    memset(&out, 0, sizeof(out));
    out.v_pi = v;
    return run_forward(d, in, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int gops_value_backward(const GopsMlp* value, int32_t batch, const float* obs, const float* grad_v,
                        const GopsMlpGrad* grad, void* workspace, size_t workspace_bytes, void* stream) {
    if (!value || !obs || !grad_v || !grad) return GOPS_ERR_BAD_ARG;
    const GopsRolloutDesc d = value_desc(*value, batch);
    GopsRolloutIn in;
    memset(&in, 0, sizeof(in));
    in.obs = obs;
    return run_backward(d, in, grad_v, *grad, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int gops_value_backward_update(const GopsMlp* value, int32_t batch, const float* obs, const float* grad_v,
                               const GopsMlpGrad* grad, const GopsUpdateTail* tail, void* workspace, size_t workspace_bytes, void* stream) {
    if (!value || !obs || !grad_v || !grad || !tail) return GOPS_ERR_BAD_ARG;
    const GopsRolloutDesc d = value_desc(*value, batch);
    GopsRolloutIn in;
    memset(&in, 0, sizeof(in));
    in.obs = obs;
    return run_backward(d, in, grad_v, *grad, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream), nullptr, nullptr, true, tail);
}

int gops_adam_step(const GopsAdamTensors* tensors, GopsAdamState* state_dev, double beta1, double beta2,
                   double eps, void* stream) {
    if (!tensors || !state_dev || tensors->n < 1 || tensors->n > GOPS_ADAM_MAX_TENSORS) return GOPS_ERR_BAD_ARG;
    for (int i = 0; i < tensors->n; ++i)
        if (!tensors->param[i] || !tensors->grad[i] || !tensors->exp_avg[i] || !tensors->exp_avg_sq[i] ||
            tensors->numel[i] < 1) return GOPS_ERR_BAD_ARG;
    return (int)launch_adam(*tensors, state_dev, beta1, beta2, (float)eps, static_cast<hipStream_t>(stream));
}

int gops_polyak_update(const GopsAdamTensors* tensors, double tau, void* stream) {
    if (!tensors || tensors->n < 1 || tensors->n > GOPS_ADAM_MAX_TENSORS) return GOPS_ERR_BAD_ARG;
    for (int i = 0; i < tensors->n; ++i)
        if (!tensors->param[i] || !tensors->grad[i] || tensors->numel[i] < 1) return GOPS_ERR_BAD_ARG;
    return (int)launch_polyak(*tensors, (float)(1.0 - tau), (float)tau, static_cast<hipStream_t>(stream));
}

int gops_value_loss(const float* v, const float* target, int32_t n, float* grad, float* stats, void* stream) {
    if (!v || !target || !stats || n < 1) return GOPS_ERR_BAD_ARG;
    return (int)launch_batch_loss(v, target, n, (float)(2.0 / n), 1.f, grad, stats, static_cast<hipStream_t>(stream));
}

int gops_mean_loss(const float* x, int32_t n, double scale, float* stats, void* stream) {
    if (!x || !stats || n < 1) return GOPS_ERR_BAD_ARG;
    return (int)launch_batch_loss(x, nullptr, n, 0.f, (float)scale, nullptr, stats, static_cast<hipStream_t>(stream));
}

void gops_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
}

void gops_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto& v : g_prof.ev) {
        for (auto& pr : v) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        v.clear();
    }
}

int gops_profile_read(int32_t kernel_id, double* avg_ms, int64_t* launches) {
    if (kernel_id < 0 || kernel_id > 5 || !avg_ms || !launches) return GOPS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double tot = 0.0;
    int64_t n = 0;
    for (auto& pr : g_prof.ev[kernel_id]) {
        if (hipEventSynchronize(pr.second) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            tot += ms;
            ++n;
        }
    }
    *avg_ms = n ? tot / n : 0.0;
    *launches = n;
    return GOPS_OK;
}

}  // extern "C"
