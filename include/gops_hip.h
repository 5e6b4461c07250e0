/*
 * gops_hip.h - C ABI of the MI355X-native GOPS ADP hot path (libgops_hip.so).
 *
 * Drop-in boundary for the reference's model-based ADP gradient loop.  Each entry point names
 * the reference interface it replaces (paths relative to the GOPS tree):
 *
 *   gops_rollout_forward   <- the H-step loop  `a = policy(o, t); o,r,d,info = envmodel.forward(...)`
 *                             of FHADP._compute_loss_policy      (gops/algorithm/fhadp.py:113-125)
 *                             and INFADP.__compute_loss_v/_policy (gops/algorithm/infadp.py:159-213),
 *                             i.e. MLP policy eval (gops/apprfunc/mlp.py:73-77,103-111), the wrapper
 *                             chain (gops/create_pkg/create_env_model.py:104-126) and the env models
 *                             pyth_lq / pyth_idpendulum / pyth_veh3dofconti
 *                             (gops/env/env_ocp/resources/lq_base.py:343-354,
 *                              gops/env/env_ocp/env_model/pyth_idpendulum_model.py:199-216,
 *                              gops/env/env_ocp/env_model/pyth_veh3dofconti_model.py:91-145).
 *   gops_rollout_backward  <- `loss.backward()` of the same functions (fhadp.py:108, infadp.py:143,151):
 *                             fills per-parameter gradients in torch nn.Linear layout.
 *   gops_rollout_backward_open_loop <- `loss.backward()` of FHADP2 (gops/algorithm/fhadp2.py:89) down to
 *                             the action sequence emitted by the single policy evaluation.
 *   gops_env_step          <- one wrapped `env_model.forward(obs, action, done, info)`
 *                             (gops/env/env_ocp/env_model/pyth_base_model.py:59-67), kept for
 *                             per-step consumers (gops/sys_simulator/opt_controller.py:240-300).
 *   gops_value_forward/backward <- `v = self.networks.v(o)` and its backward in INFADP PEV
 *                             (infadp.py:167,185; StateValue gops/apprfunc/mlp.py:327-329).
 *
 * Conventions: every buffer is caller-allocated DEVICE memory (fp32 unless noted); the library
 * allocates no device memory on the data path and keeps no per-call state, so calls on different
 * workspaces are re-entrant.  Process-wide state is limited to: the opt-in timing hook
 * (gops_profile_*: HIP events, mutex-guarded), the cached CU count of the device, a one-time
 * hipFuncSetAttribute per kernel, and - debug builds only - one counter buffer.  All work is enqueued on `stream`
 * (a hipStream_t passed as void*); no call synchronises.  Return value: 0 on success, a
 * negative GOPS_ERR_* code or a positive hipError_t otherwise; nothing throws across the ABI.
 */
#ifndef GOPS_HIP_H
#define GOPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define GOPS_HIP_ABI_VERSION 13

#define GOPS_MAX_LAYERS 5   /* Linear layers per MLP (<= 4 hidden + output) */
#define GOPS_MAX_ACT 4      /* action dimensions */
#define GOPS_MAX_LQ_STATE 6
#define GOPS_MAX_HORIZON 256
#define GOPS_TILE 16        /* trajectories per workgroup tile (MFMA M) */

enum { GOPS_OK = 0, GOPS_ERR_BAD_ARG = -1, GOPS_ERR_UNSUPPORTED = -2, GOPS_ERR_WORKSPACE = -3 };

/* env kinds: the three env models named by BASELINE.json + NONE (plain MLP batch evaluation) */
enum { GOPS_ENV_NONE = 0, GOPS_ENV_LQ = 1, GOPS_ENV_IDPENDULUM = 2, GOPS_ENV_VEH3DOFCONTI = 3,
       /* veh3dofconti + surrounding vehicles + constraint outputs: pyth_veh3dofconti_surrcstr_model.py:42-148 and
          pyth_veh3dofconti_detour_model.py:40-181 (the models behind FHADPExterior / Interior / Lagrangian) */
       GOPS_ENV_VEH3DOF_SURR = 4,
       /* gym-style models of the INFADP / MAC example scripts (obs == state, no info):
          gops/env/env_gym/env_model/gym_cartpoleconti_model.py:24-129 and gym_pendulum_model.py:26-115 */
       GOPS_ENV_CARTPOLE = 5, GOPS_ENV_PENDULUM = 6,
       /* pyth_veh2dofconti_model.py:24-174: lateral 2-DOF vehicle at constant speed tracking the reference path: state
          (y, phi, v, omega) [B,4], action steer, obs = (y - y_ref0, phi - phi_ref0, v, omega, y - y_ref_1 .. y - y_ref_P),
          info["ref_points"] [B, P+1, 2] = (y, phi) */
       GOPS_ENV_VEH2DOF = 7,
       /* pyth_mobilerobot_model.py:24-213 (the model of example_train/spil/spil_mlp_mobilerobot_{offserial,async}.py):
          obs == state [B,13] = ego (x, y, theta, v, w), tracking errors (e_y, e_theta, e_v), one obstacle robot
          (x, y, theta, v, w); action (v_cmd, w_cmd); dt = 0.2.  The ego follows its rate-limited, saturated commands, the
          obstacle its own (v, w) plus the noise draws handed in through GopsRolloutIn.noise / GopsStepIO.noise;
          info["constraint"] [B,1] = 0.89 - distance(obstacle', ego') of the NEW state (n_constraint = 1, unmasked);
          done = x' < -2 | |y'| > 4 | constraint > 0.15.  fp32 only; ClipObservation over all 13 columns. */
       GOPS_ENV_MOBILEROBOT = 8 };
#define GOPS_MAX_SURR 4        /* surrounding vehicles */
#define GOPS_MAX_CONSTRAINT 3  /* constraint outputs per step */
#define GOPS_MAX_REPEAT 8      /* ActionRepeatModel: sub-steps per env step */
#define GOPS_MAX_CLIP_OBS 16   /* observation columns ClipObservationModel can act on */

/* hidden activations: gops/utils/common_utils.py:26-55 */
enum { GOPS_ACT_LINEAR = 0, GOPS_ACT_RELU = 1, GOPS_ACT_ELU = 2, GOPS_ACT_GELU = 3,
       GOPS_ACT_SELU = 4, GOPS_ACT_SIGMOID = 5, GOPS_ACT_TANH = 6 };

/* Arithmetic of the MLP contractions (BASELINE.json configs[4]: "fp16 MFMA MLP path").
 *   GOPS_DTYPE_F32: fp32 results at the 1e-4 parity bar (default).  NOT bit-level fp32 arithmetic where the plane-split
 *                   kernels run (gops_rollout_variant bits 0 / 2; every 256-wide net): weights, activations and deltas are
 *                   each carried as two IEEE-half planes x s = hi + lo / 2^11 (22 significant bits, fp32 has 24; s a power of
 *                   two - per 16 output features for weights, 2^-4 for forward activations, from max|delta| in the sweep),
 *                   a w = hi hi + (lo hi + hi lo) / 2^11 on three v_mfma_f32_16x16x32_f16 with fp32 accumulation; the large
 *                   weight-gradient GEMMs likewise (deltas scaled by the sweep's largest |delta_y|, saturated blocks redone exactly).
 *                   Forward range |activation| < 1.05e6: beyond it the launch's results are NaN (never silently wrong).
 *                   Measured distance to the reference on trained 256-wide networks 2e-6 .. 3e-5, at the level of the exact
 *                   fp32 kernels.  GOPS_VF_STREAMED_FP32 | GOPS_VF_DW_F32 in the descriptor's
 *                   variant_flags select v_mfma_f32_16x16x4_f32 (an fmaf chain) throughout.
 *   GOPS_DTYPE_F16: weights, hidden activations and deltas rounded to IEEE half, products accumulated
 *                   in fp32 by v_mfma_f32_16x16x32_f16; the activation stash is half (2 bytes/element).
 *                   Env model, wrapper chain, rewards, returns, observation / state adjoints, policy
 *                   head and every result stay fp32.  The backward pass runs on gradients scaled by a
 *                   power of two chosen from max|grad_v| on the device (deltas stay inside half's
 *                   range) and un-scales the parameter gradients; tolerance: DESIGN.md section 2. */
enum { GOPS_DTYPE_F32 = 0, GOPS_DTYPE_F16 = 1 };

/* An MLP in torch nn.Linear layout: layer j has weight [sizes[j+1]][sizes[j]] row-major and
 * bias [sizes[j+1]].  Hidden widths must be multiples of 16 (64 with GOPS_DTYPE_F16); the input width is arbitrary.
 * Weights and biases are always fp32 in memory; the F16 path rounds them when it packs them. */
typedef struct GopsMlp {
    int32_t n_layers;                      /* number of Linear layers, 2..GOPS_MAX_LAYERS */
    int32_t sizes[GOPS_MAX_LAYERS + 1];    /* in, hidden..., out */
    int32_t hidden_act;                    /* GOPS_ACT_* */
    int32_t dtype;                         /* GOPS_DTYPE_*: read by gops_value_forward / _backward only
                                              (a rollout takes GopsRolloutDesc.dtype for all its nets) */
    uint32_t variant_flags;                /* ABI v10: GOPS_VF_* for gops_value_* / gops_mlp_* calls (a rollout takes
                                              GopsRolloutDesc.variant_flags); 0 = the library's choice.  Occupies what was
                                              alignment padding in v9: offsets and size are unchanged */
    const float* weight[GOPS_MAX_LAYERS];  /* device pointers */
    const float* bias[GOPS_MAX_LAYERS];
} GopsMlp;

/* Gradients of one MLP, same shapes/layout as GopsMlp weight/bias (device, overwritten). */
typedef struct GopsMlpGrad {
    float* weight[GOPS_MAX_LAYERS];
    float* bias[GOPS_MAX_LAYERS];
} GopsMlpGrad;

/* The wrapped env model: base model constants + the wrapper chain built by create_env_model. */
typedef struct GopsEnv {
    int32_t kind;                 /* GOPS_ENV_* */
    int32_t obs_dim, act_dim;
    int32_t pre_horizon;          /* veh3dofconti: number of preview points P (obs_dim = 6+4P) */
    /* ScaleActionModel / ClipActionModel (scale_action.py:75-83, clip_action.py:34-36) */
    float min_action[GOPS_MAX_ACT], max_action[GOPS_MAX_ACT];
    float act_low[GOPS_MAX_ACT], act_high[GOPS_MAX_ACT];       /* base-model action bounds */
    /* tanh squash of the policy head (mlp.py:73-77): act_high_lim / act_low_lim buffers */
    float policy_low[GOPS_MAX_ACT], policy_high[GOPS_MAX_ACT];
    /* ClipObservationModel (clip_observation.py:38-40); +-inf = inactive */
    int32_t clip_obs;             /* 0: all bounds infinite (idpendulum, veh3dofconti) */
    float obs_low[GOPS_MAX_CLIP_OBS], obs_high[GOPS_MAX_CLIP_OBS];
    /* ShapingRewardModel (shaping_reward.py:84-88), applied outside MaskAtDone */
    int32_t shaping;
    float reward_scale, reward_shift;
    /* pyth_lq constants (lq_base.py:39-57): x' = inv_IA (x + dt B u), r = rs*(rsh - (Qx^2+Ru^2)) */
    float lq_inv_IA[GOPS_MAX_LQ_STATE * GOPS_MAX_LQ_STATE];    /* row-major n x n */
    float lq_B[GOPS_MAX_LQ_STATE * GOPS_MAX_ACT];              /* row-major n x m */
    float lq_Q[GOPS_MAX_LQ_STATE], lq_R[GOPS_MAX_ACT];
    float lq_dt, lq_reward_scale, lq_reward_shift;
    /* GOPS_ENV_VEH3DOF_SURR: obs_dim = 6 + 4 P + 4 n_surr; each surrounding vehicle follows its own kinematic
     * bicycle (x, y, phi, u, delta; SurrVehicleModel :28-39) independently of the policy; the observation appends
     * (x, y, phi, u)_surr - (x, y, phi, u)_ego per vehicle; constraint[0] = 2 r - min distance between the two ego
     * circles and the two circles of every surrounding vehicle (bicircle model, :98-148); n_constraint = 3 adds the road
     * boundary violations of the detour model (:143-151); the stage reward is
     * -(w[0] dx^2 + w[1] dy^2 + w[2] dphi^2 + w[3] du^2 + w[4] omega^2 + w[5] steer^2 + w[6] a_x^2 + w[7] v^2).
     * surr_penalty = 1 selects pyth_veh3dofconti_surrcstr_penalty_model.py:74-246 (one surrounding vehicle): the stage
     * reward also carries the collision penalty -15 (tanh(max(8 - 16 dis, 0) - 4) + 1), dis = min circle distance - 2 r,
     * evaluated on the CURRENT state; the appended observation is the NEXT surrounding-vehicle state in the ego frame of
     * the CURRENT state (x, y, phi rotated, u relative); the model never reports done; info["constraint"] (and the
     * constraint sums) are those of the CURRENT pose and carry no gradient (the model computes them on detached copies). */
    /* 1: no MaskAtDoneModel in the chain (mask_at_done.py:33-40): the model keeps stepping after its done test fired,
     * rewards are not zeroed (the raw model OptController drives, opt_controller.py:261-265; create_env_model(mask_at_done =
     * False), create_env_model.py:104-105).  The done flags handed in are ignored; final_done (and the mask of a tail
     * value) is the base model's done test on the LAST state.  Not for GOPS_ENV_VEH3DOF_SURR (final_done stays 0 there). */
    int32_t no_mask_at_done;
    int32_t n_surr, n_constraint, surr_penalty;
    float veh_length, veh_width, road_upper, road_lower;
    float reward_w[8];
    /* gops_env_step only (rollouts reject it): 1 = the step of the DATA environment the reference's samplers drive
     * (gops/env/env_ocp/pyth_veh3dofconti.py:195-271, resources/lq_base.py:209-231, pyth_idpendulum.py:71-87) instead
     * of the env MODEL's: same dynamics and stage reward, but the data env's termination tests (veh3dofconti: world-frame
     * |x - x_ref| > 5, |y - y_ref| > 2, |dphi| > pi; lq: next state outside the state bounds), a -100 terminal penalty
     * (veh3dofconti, lq), no observation clipping and no MaskAtDone (an episode that is done gets reset by the caller:
     * the `done` input is ignored).  GOPS_ENV_MOBILEROBOT (round 6; pyth_mobilerobot.py:108-152): the model's step with both new
     * headings clipped to +-pi, no terminal penalty. */
    int32_t data_env;
    /* ScaleObservationModel (gops/env/wrapper/scale_observation.py:74-119; create_env_model.py:115-118 puts it between
     * ShapingReward and ClipObservation): the observations the policy and the caller see are (obs + obs_shift) * obs_scale;
     * the model steps obs / obs_scale - obs_shift.  GOPS_ENV_LQ / _IDPENDULUM / _CARTPOLE / _PENDULUM only (obs_dim <= 8).  As in the
     * reference, ClipObservationModel then clips the SCALED observation with the model's own (unscaled) bounds. */
    int32_t scale_obs;
    float obs_scale[8], obs_shift[8];
    /* GOPS_ENV_VEH3DOF_SURR with cstr_err = 1: pyth_veh3dofconti_errcstr_model.py:22-55 - no surrounding vehicles
     * (n_surr = 0), n_constraint = 2: info["constraint"] = (|obs[1]| - err_tol[0], |obs[3]| - err_tol[1]) of the CURRENT
     * observation (lateral and speed tracking errors). */
    int32_t cstr_err;
    float err_tol[2];
    /* Reference-trajectory parameters of the veh3dofconti / veh2dofconti families (MultiRefTrajModel(path_para, u_para),
     * gops/env/env_ocp/resources/ref_traj_model.py:26-52; defaults ref_traj_data.py:18-37).  ref_custom = 0: the library
     * uses the default set.  ref_custom = 1: ref_c holds the constants, each folded in double on the host exactly where
     * the reference folds Python scalars and then rounded to fp32:
     *   [0] -A/omega  [1] omega  [2] phi  [3] b  [4] A/omega*cos(phi)  [5] A      sine speed profile
     *   [6] u                                                                   constant speed profile
     *   [7] A  [8] omega  [9] phi                                                sine path
     *   [10..13] t1..t4  [14] y1  [15] y2  [16] (y2-y1)/(t2-t1)  [17] (y1-y2)/(t4-t3)   double-lane path
     *   [18] T  [19] 2A/T  [20] -2A/T  [21] T/2                                  triangle path
     *   [22] r                                                                  circle path */
    int32_t ref_custom;
    float ref_c[24];
    /* ActionRepeatModel around MaskAtDoneModel (gops/create_pkg/create_env_model.py:104-107, gops/env/wrapper/
     * action_repeat.py:54-87): repeat_num >= 2 applies the masked base step repeat_num times to the advancing observation
     * with the INITIAL done flags; the rewards are summed (repeat_last_reward = 1: only the last one is returned), done is
     * the last sub-step's.  0 / 1 = no wrapper.  Supported for the models whose observation is the state (GOPS_ENV_LQ,
     * _IDPENDULUM, _CARTPOLE, _PENDULUM; the reference wrapper does not advance `info`), fp32, repeat_num <= GOPS_MAX_REPEAT. */
    int32_t repeat_num;
    int32_t repeat_last_reward;
} GopsEnv;

typedef struct GopsRolloutDesc {
    int32_t batch;            /* B trajectories */
    int32_t horizon;          /* H = pre_horizon (FHADP) or forward_step (INFADP) */
    int32_t finite_horizon;   /* 1: FiniteHorizonPolicy, time index t+1 appended to the input */
    int32_t need_grad;        /* 1: keep the activation stash for gops_rollout_backward */
    int32_t tail_value;       /* 1: v += (~done_H) gamma^H V(obs_H)   (infadp.py:182-184, 210) */
    int32_t open_loop;        /* 1: no policy evaluation inside the rollout - the pre-tanh head outputs of
                                 every step come from GopsRolloutIn.head_pre (FHADP2: one MLP evaluation
                                 emits all H actions, gops/algorithm/fhadp2.py:100-121); `policy` is ignored.
                                 2: as 1, but head_pre holds the MODEL ACTIONS themselves (no tanh squash, no
                                 ScaleAction / ClipAction): the shooting rollout of an optimal-control solver over
                                 the raw model, gops/sys_simulator/opt_controller.py:240-300; the backward returns
                                 d(loss)/d(action) */
    int32_t dtype;            /* GOPS_DTYPE_F32 (default) or GOPS_DTYPE_F16: arithmetic of the MLP contractions
                                 (hidden widths must then be multiples of 64) */
    int32_t tail_unmasked;    /* with tail_value: 1 = v += gamma^H V(obs_H) for finished trajectories too
                                 (SPIL's evaluation target, gops/algorithm/spil.py:208) */
    uint32_t variant_flags;   /* ABI v10: GOPS_VF_* bits - which kernel variants this rollout may NOT / must take.  0 = the
                                 library's choice (gops_rollout_variant reports it).  Part of the description: two callers
                                 in one process can choose differently, and forward / backward of one rollout must pass
                                 the same value */
    int32_t l2_warmup;        /* ABI v10: backward sweep's L2 warm-up of the next step's stash rows: 0 = library default,
                                 1 = off, 2 = all at the step top, 3 = spread over the step (tuning) */
    int32_t dw_workgroups;    /* ABI v10: target workgroup count of a weight-gradient GEMM, 0 = default (512) (tuning) */
    int32_t reserved0;
    double gamma;             /* discount; gamma^t is formed in double then rounded (fhadp.py:120) */
    GopsEnv env;
    GopsMlp policy;           /* out width = act_dim */
    GopsMlp value;            /* used iff tail_value (out width 1): INFADP's v_target */
} GopsRolloutDesc;

/* Batch inputs (the `data` dict of the algorithms). */
typedef struct GopsRolloutIn {
    const float* obs;         /* [B, obs_dim] */
    const float* done;        /* [B] 0/1 */
    const float* state;       /* veh3dofconti info["state"]      [B,6]      else NULL */
    const float* ref_points;  /* veh3dofconti info["ref_points"] [B,P+1,4]  else NULL */
    const float* path_num;    /* [B] */
    const float* u_num;       /* [B] */
    const float* ref_time;    /* [B] */
    const float* head_pre;    /* open_loop only: [B, H, act_dim] policy-head outputs BEFORE the tanh squash */
    const float* surr_state;  /* GOPS_ENV_VEH3DOF_SURR info["surr_state"] [B, n_surr, 5] (x, y, phi, u, delta) else NULL */
    const float* grad_constraint; /* gops_rollout_backward, GOPS_ENV_VEH3DOF_SURR: d(loss)/d(constraint_sums) [3, B]
                                 (rows as in GopsRolloutOut.constraint_sums); NULL = zeros */
    const float* grad_constraint_prod; /* gops_rollout_backward, GOPS_ENV_VEH3DOF_SURR: [n_constraint, B]
                                 d(loss)/d(P_k) * P_k for the products P_k of GopsRolloutOut.constraint_prods; NULL = zeros */
    const float* ref_appended; /* ABI v8, models with reference trajectories, or NULL: [B, H, 4] (x, y, phi, u) - the point the
                                 model APPENDS to info["ref_points"] at rollout step s (ref_traj_model.py:26-148 evaluated at
                                 ref_time + (s + 1) dt + P dt), supplied by the caller instead of evaluated by the library.
                                 Bit-parity mode: the reference's heading is a 1 ms finite difference in fp32 whose last-ulp
                                 behaviour depends on the host's libm; a caller that fills this tensor with the reference's own
                                 MultiRefTrajModel values gets observations identical to the reference's. */
    const float* noise;       /* ABI v9, GOPS_ENV_MOBILEROBOT, or NULL (= zeros): [H, B, 2] the obstacle robot's N(0, 0.03) / N(0, 0.02)
                                 draws of every rollout step, exactly what np.random.normal returns inside Robot.f_xu(.., "obs")
                                 (pyth_mobilerobot_model.py:143-167; the model adds 0.5 x draw to the obstacle's v / w).  Must stay
                                 valid until the backward call (d x' / d theta of the obstacle depends on it). */
    const float* grad_constraint_step; /* ABI v9, backward calls of the models with constraint outputs, or NULL (= zeros): [H, B, n_constraint]
                                 d(loss)/d(c_tk) for the per-step constraint values of GopsRolloutOut.constraints - a caller that needs
                                 the Jacobian of the constraint path (OptController's inequality constraints, gops/sys_simulator/
                                 opt_controller.py:178-206) replicates the trajectory once per row and seeds one unit entry each */
} GopsRolloutIn;

typedef struct GopsRolloutOut {
    float* v_pi;              /* [B] sum_t gamma^t r_t (+ tail); required */
    float* rewards;           /* [H,B] per-step (masked, shaped) rewards, or NULL */
    float* final_obs;         /* [B, obs_dim] or NULL */
    float* final_done;        /* [B] or NULL */
    float* final_state;       /* veh3dofconti [B,6] or NULL */
    /* Models with constraint outputs (GOPS_ENV_VEH3DOF_SURR, _VEH2DOF with cstr_err, _MOBILEROBOT), or NULL: [4, B] discounted sums over the rollout of the UNMASKED info["constraint"] c_t
     * (the algorithms read it regardless of `done`):
     *   row 0  sum_t gamma^t sum_k max(c_tk, 0)^2            (fhadp_exterior.py:64, fhadp_interior.py:66)
     *   row 1  sum_t gamma^t sum_k max(c_tk, 0)              (fhadp_lagrangian.py:66)
     *   row 2  sum_t gamma^t sum_k log(-min(c_tk, 0) + 1e-8) (fhadp_interior.py:65)
     *   row 3  1 if every c_tk < 0 (feasible trajectory, fhadp_interior.py:71) else 0 */
    float* constraint_sums;
    /* Same models, or NULL: [2 n_constraint, B] products over the rollout (gops/algorithm/spil.py:189-251):
     *   rows 0 .. n_constraint-1        P_k = prod_t Phi(c_tk),  Phi(y) = 1.07 / (1 + 0.0315 exp(clamp(y / 0.07, -10, 5)))
     *   rows n_constraint .. 2 n_c - 1  prod_t [c_tk <= 0]       (trajectory safe w.r.t. constraint k: 0 / 1) */
    float* constraint_prods;
    /* ABI v9, same models, or NULL: [H, B, n_constraint] the UNMASKED info["constraint"] the model returns at every step
     * (errcstr models: of the observation the step STARTS from; surrcstr / detour / mobilerobot: of the NEW state) */
    float* constraints;
} GopsRolloutOut;

int gops_hip_version(void);

/* Bytes of scratch `workspace` needed by forward(+backward) for this descriptor. */
size_t gops_rollout_workspace_bytes(const GopsRolloutDesc* desc);

int gops_rollout_forward(const GopsRolloutDesc* desc, const GopsRolloutIn* in,
                         const GopsRolloutOut* out, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Back-propagate d(loss)/d(v_pi[b]) = grad_v[b] through the stashed rollout into the policy
 * parameters (value/target parameters receive no gradient, as in the reference). */
int gops_rollout_backward(const GopsRolloutDesc* desc, const GopsRolloutIn* in,
                          const float* grad_v, const GopsMlpGrad* policy_grad,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Open-loop counterpart of gops_rollout_backward: d(loss)/d(head_pre) [B, H, act_dim] of a forward
 * run with desc->open_loop = 1 (the caller back-propagates it through its single MLP evaluation -
 * FiniteHorizonFullPolicy.forward_all_policy, gops/apprfunc/mlp.py:140-145). */
int gops_rollout_backward_open_loop(const GopsRolloutDesc* desc, const GopsRolloutIn* in,
                                    const float* grad_v, float* grad_head_pre,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* Adjoints around a closed-loop rollout (ABI v5): what a caller needs to put a rollout in the middle of a larger
 * differentiable expression.  MPG's model return (gops/algorithm/mpg.py:340-353) is
 *     sum_t gamma^t r_t + gamma^H q1_target(o_H, policy(o_H))
 * where only step 0 acts through `policy`; later steps act through the frozen copy `policy4rollout` (same weights,
 * requires_grad False: gradients still flow through its INPUT).  The terminal term is evaluated by the caller
 * (gops_mlp_forward / gops_mlp_backward_x) and comes back as `grad_final_obs`.
 * Supported for the env kinds whose observation is the model state (GOPS_ENV_NONE, _LQ, _IDPENDULUM, _CARTPOLE,
 * _PENDULUM), fp32, tail_value = 0, closed loop. */
typedef struct GopsRolloutAdjoint {
    const float* grad_final_obs;  /* [B, obs_dim] d(loss)/d(final_obs) (GopsRolloutOut.final_obs); NULL = zeros */
    float* grad_obs;              /* out [B, obs_dim]: d(loss)/d(obs) of the initial observation; NULL = not wanted */
    int32_t first_step_only;      /* 1: parameter gradients only through the action of step 0 (mpg.py:343-349) */
    int32_t reserved;
} GopsRolloutAdjoint;
/* gops_rollout_backward with the adjoints above; policy_grad may be NULL (no parameter gradients wanted). */
int gops_rollout_backward_adj(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                              const GopsMlpGrad* policy_grad, const GopsRolloutAdjoint* adj,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Open-loop rollout with adjoint I/O (ABI v8): gops_rollout_backward_open_loop seeded with d(loss)/d(final_obs) and
 * returning d(loss)/d(obs) of the initial observation next to d(loss)/d(head_pre) - one collocation interval of
 * OptController (gops/sys_simulator/opt_controller.py:104-109, 196-215: the decision variables are the actions AND the
 * states at the control points; cost and transition-constraint Jacobians need both adjoints).  Same env kinds as
 * gops_rollout_backward_adj; adj->first_step_only is ignored. */
int gops_rollout_backward_open_loop_adj(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                                        float* grad_head_pre, const GopsRolloutAdjoint* adj,
                                        void* workspace, size_t workspace_bytes, void* stream);

/* One wrapped env-model step.  `action` is the raw (pre-wrapper) action [B, act_dim].  For
 * veh3dofconti the info tensors are updated into the next_* outputs (may alias the inputs
 * except ref_points). */
typedef struct GopsStepIO {
    const float* obs; const float* action; const float* done;
    const float* state; const float* ref_points; const float* path_num; const float* u_num;
    const float* ref_time;
    float* next_obs; float* reward; float* next_done;
    float* next_state; float* next_ref_points; float* next_ref_time;
    /* GOPS_ENV_VEH3DOF_SURR: info["surr_state"] [B, n_surr, 5] in / out, info["constraint"] [B, n_constraint] out */
    const float* surr_state; float* next_surr_state; float* constraint;
    const float* ref_appended;   /* ABI v8, or NULL: [B, 4] (veh2dofconti: (., y, phi, .)) the reference point this step appends,
                                    from the caller (see GopsRolloutIn.ref_appended) */
    const float* noise;          /* ABI v9, GOPS_ENV_MOBILEROBOT, or NULL (= zeros): [B, 2] this step's obstacle draws (GopsRolloutIn.noise) */
} GopsStepIO;
int gops_env_step(const GopsEnv* env, int32_t batch, const GopsStepIO* io, void* stream);

/* model.get_constraint(obs, info) (ABI v9; gops/env/env_ocp/env_model/pyth_base_model.py:69-75 - the hook OptController evaluates on
 * every state of its prediction, opt_controller.py:178-198): io->constraint [B, n_constraint] of the given io->obs (errcstr models:
 * pyth_veh3dofconti_errcstr_model.py:49-56, pyth_veh2dofconti_errcstr_model.py:47-51) or of the given io->state / io->surr_state
 * (pyth_veh3dofconti_surrcstr_model.py:98-148, _detour_model.py:102-151, _surrcstr_penalty_model.py:183-234).  Nothing else of
 * `io` is read or written.  GOPS_ERR_UNSUPPORTED for models that define no get_constraint. */
int gops_env_constraint(const GopsEnv* env, int32_t batch, const GopsStepIO* io, void* stream);

/* StateValue batch evaluation v = V(obs) with stash, and its backward into V's parameters. */
size_t gops_value_workspace_bytes(const GopsMlp* value, int32_t batch);
int gops_value_forward(const GopsMlp* value, int32_t batch, const float* obs, float* v,
                       void* workspace, size_t workspace_bytes, void* stream);
int gops_value_backward(const GopsMlp* value, int32_t batch, const float* obs, const float* grad_v,
                        const GopsMlpGrad* grad, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Batch evaluation of an MLP whose output layer has ANY width, and its backward into the parameters:
 * FiniteHorizonFullPolicy.pi (gops/apprfunc/mlp.py:114-145: obs -> hidden ... -> act_dim * pre_horizon, the single
 * evaluation of FHADP2, gops/algorithm/fhadp2.py:100-104) and `pre.backward(grad)` (:89).  y, grad_y: [batch, sizes[n_layers]]
 * row-major.  The hidden stack runs on the rollout kernels (GOPS_ENV_NONE tiles), the output layer on its own kernels;
 * mlp->dtype must be GOPS_DTYPE_F32. */
size_t gops_mlp_workspace_bytes(const GopsMlp* mlp, int32_t batch);
int gops_mlp_forward(const GopsMlp* mlp, int32_t batch, const float* x, float* y, void* workspace,
                     size_t workspace_bytes, void* stream);
int gops_mlp_backward(const GopsMlp* mlp, int32_t batch, const float* x, const float* grad_y,
                      const GopsMlpGrad* grad, void* workspace, size_t workspace_bytes, void* stream);
/* gops_mlp_backward that also returns d(loss)/d(x) [batch, sizes[0]] (ABI v5); `grad` may be NULL when only the input
 * adjoint is wanted: `q1(o, policy(o))` differentiated with respect to the action with frozen q parameters
 * (gops/algorithm/mpg.py:186-191,338) and the policy evaluated at the rollout's last observation (:352-354). */
int gops_mlp_backward_x(const GopsMlp* mlp, int32_t batch, const float* x, const float* grad_y,
                        const GopsMlpGrad* grad, float* grad_x, void* workspace, size_t workspace_bytes,
                        void* stream);

/* One Adam step (torch.optim.Adam defaults semantics: no weight decay, no amsgrad) over up to
 * GOPS_ADAM_MAX_TENSORS parameter tensors in ONE launch - replaces `self.networks.policy_optimizer.step()`
 * (gops/algorithm/fhadp.py:89, gops/algorithm/infadp.py:124).  The learning rate and the step count
 * live in DEVICE memory (`GopsAdamState`, caller-owned; initialise step = updates done so far,
 * beta1_pow = beta1^step, beta2_pow = beta2^step, ticket = 0): the kernel uses t = step + 1 for the
 * bias corrections and its last block stores the advanced state, so the call can be captured in a
 * HIP graph and replayed.  exp_avg / exp_avg_sq are updated in place.  Every gradient element is
 * multiplied by `grad_scale` as it is read (1/N after a SUM all-reduce over N data-parallel replicas:
 * the averaging costs no extra pass over the gradient buffer; set it to 1 otherwise). */
#define GOPS_ADAM_MAX_TENSORS 16
typedef struct GopsAdamTensors {
    int32_t n;
    int32_t reserved;
    int64_t numel[GOPS_ADAM_MAX_TENSORS];
    float* param[GOPS_ADAM_MAX_TENSORS];
    const float* grad[GOPS_ADAM_MAX_TENSORS];
    float* exp_avg[GOPS_ADAM_MAX_TENSORS];
    float* exp_avg_sq[GOPS_ADAM_MAX_TENSORS];
} GopsAdamTensors;
typedef struct GopsAdamState {   /* 48 bytes of device memory */
    double lr;
    int64_t step;
    double beta1_pow, beta2_pow;
    uint32_t ticket;
    uint32_t skipped_nonfinite;   /* ABI v13 (was `reserved`): gradient elements that were NOT finite and therefore took no step (parameter,
                                   * moments and - in a fused tail - Polyak target untouched), cumulative; the caller reads / clears it */
    double grad_scale;
} GopsAdamState;
int gops_adam_step(const GopsAdamTensors* tensors, GopsAdamState* state_dev, double beta1, double beta2,
                   double eps, void* stream);

/* ABI v12.  A single-process update - compute_gradient, then optimizer.step() (gops/algorithm/fhadp.py:87-90, infadp.py:101-104) - in
 * ONE call: gops_rollout_backward with what follows it folded into its last kernel, the split-K reduce that forms the final
 * gradients: the Adam step of gops_adam_step on every gradient element as it is formed (`adam`), and the loss mean of
 * gops_mean_loss in one more block of the same launch (`mean_x`; more than 8192 values: its own launch, still inside this call).
 * Element for element the arithmetic of the three separate calls (tests: bit-equal weights, moments and loss scalars); two
 * launches and their gaps less per update.  The gradients are written to policy_grad as well.  With GOPS_VF_BWD_PHASE_A only a tail
 * WITHOUT `adam` / `polyak` is accepted (round 6: the loss mean rides on phase A's reduce launch; a data-parallel update all-reduces
 * between gradient and optimizer step), never with GOPS_VF_BWD_PHASE_B, and not for open-loop rollouts. */
typedef struct GopsUpdateTail {
    const GopsAdamTensors* adam;   /* NULL: no optimizer step.  grad[i] must be one of policy_grad's tensors with numel[i] its element
                                    * count, and EVERY tensor of policy_grad must appear (a parameter without its step is a bug) */
    GopsAdamState* adam_state;     /* device memory, as for gops_adam_step */
    double beta1, beta2, eps;
    const float* mean_x;           /* NULL: no loss mean; else mean_n device values (the forward's v_pi) */
    int32_t mean_n;
    int32_t reserved;
    double mean_scale;             /* mean_stats[0] = mean_scale * mean(x), mean_stats[1] = mean(x) */
    float* mean_stats;             /* GOPS_LOSS_STATS_FLOATS floats, as for gops_mean_loss */
    const GopsAdamTensors* polyak; /* NULL: no target-network averaging; else the table of gops_polyak_update (param[i] = target tensor,
                                    * grad[i] = online tensor) for the network `adam` steps: every online tensor must be one of adam's
                                    * param[] - the target element is averaged with the NEW parameter value right where Adam forms it
                                    * (infadp.py:124-133 after :104), same two roundings as gops_polyak_update.  Needs `adam` */
    double polyak_tau;
} GopsUpdateTail;
int gops_rollout_backward_update(const GopsRolloutDesc* desc, const GopsRolloutIn* in, const float* grad_v,
                                 const GopsMlpGrad* policy_grad, const GopsUpdateTail* tail,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* The same for a value / MLP batch: gops_value_backward with the tail (INFADP's policy-evaluation update: Adam step of the value
 * net + Polyak step of its target in the launch that forms the gradients; tail->mean_x is usually NULL there - gops_value_loss
 * has to run BEFORE the backward, it produces grad_v). */
int gops_value_backward_update(const GopsMlp* value, int32_t batch, const float* obs, const float* grad_v,
                               const GopsMlpGrad* grad, const GopsUpdateTail* tail,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ABI v11.  Polyak averaging of a target network, every tensor in one launch - replaces the two passes of
 * gops/algorithm/infadp.py:124-133 (`p_targ.mul_(1 - tau); p_targ.add_(tau * p)`), same roundings per element.
 * Table: param[i] = target tensor i (updated in place), grad[i] = online tensor i, numel[i]; the moment slots are ignored. */
int gops_polyak_update(const GopsAdamTensors* tensors, double tau, void* stream);

/* ABI v11.  The scalars (and the loss gradient) the algorithms form from a batch of values, one launch, no host sync:
 *   gops_value_loss:  stats[0] = mean((v - target)^2), stats[1] = mean(v), grad[i] = (2 / n) (v[i] - target[i]) (grad may be NULL)
 *                     - `loss_v = ((v - backup) ** 2).mean()` and `torch.mean(v)` of gops/algorithm/infadp.py:172-173 and
 *                     d(loss_v) / d(v);
 *   gops_mean_loss:   stats[0] = scale * mean(x), stats[1] = mean(x) - `-v_pi.mean()` (infadp.py:213, fhadp.py:123) with scale = -1.
 * `stats` is 8-byte aligned device memory of GOPS_LOSS_STATS_FLOATS floats, ALL ZERO before the first call: behind the two
 * results it holds the blocks' partial sums and a ticket counter that every call leaves zero again; sums are formed in double
 * in a fixed order (run-to-run reproducible). */
#define GOPS_LOSS_STATS_FLOATS 260
int gops_value_loss(const float* v, const float* target, int32_t n, float* grad, float* stats, void* stream);
int gops_mean_loss(const float* x, int32_t n, double scale, float* stats, void* stream);

/* Which kernels a rollout description runs on this device (ABI v8; for benchmarks / profiles, no launch):
 * bit 0 (GOPS_VARIANT_SPLIT): the register-stationary kernels with plane-split contractions - hidden-layer weights,
 *        activations and deltas as two half planes each (22 bits), 3 f16 MFMAs (16x16x32) per 32-deep block, fp32
 *        accumulation and results (GOPS_DTYPE_F32 above);
 * bit 1 (GOPS_VARIANT_STATIONARY_F32): the register-stationary kernels on exact fp32 MFMAs;
 * bit 2 (GOPS_VARIANT_STREAMED_SPLIT_FWD, ABI v9): the FORWARD rollout on the streamed plane-split kernel (any number of
 *        256-wide hidden layers, weight planes streamed from L2, tail value net included - except that the tail value net of a
 *        relu / selu launch that keeps a gradient is evaluated with exact fp32 products); the backward sweep of such a
 *        launch runs on the streamed plane-split sweep where its LDS image fits twice per CU, else on the streamed fp32-MFMA kernel;
 * bit 3 (GOPS_VARIANT_HALF_TILE64, ABI v10): GOPS_DTYPE_F16 on the 64-trajectory-tile kernels (rollout_h64.hip);
 * none: the streamed kernels (exact fp32 MFMAs, or half-precision MFMAs for GOPS_DTYPE_F16).
 * Negative: a GOPS_ERR_* code for a description the library rejects. */
/* GopsRolloutDesc.variant_flags / GopsMlp.variant_flags (ABI v10; replaces the process-environment knobs of v9, which
 * survive only as a debug override read ONCE when the library is loaded - see INTEGRATION.md). */
#define GOPS_VF_NO_STATIONARY_SPLIT 0x1u     /* not the register-stationary plane-split kernels            (v9: GOPS_SPLIT=0) */
#define GOPS_VF_NO_STREAMED_SPLIT_FWD 0x2u   /* not the streamed plane-split forward                        (GOPS_SS=0) */
#define GOPS_VF_NO_STREAMED_SPLIT_BWD 0x4u   /* not the streamed plane-split sweep                          (GOPS_SSB=0) */
#define GOPS_VF_NO_STREAMED_SPLIT_VALUE 0x8u /* value / MLP batches (GOPS_ENV_NONE) on the fp32-MFMA kernels (GOPS_SS_VALUE=0) */
#define GOPS_VF_STREAMED_FP32 0x10u          /* the plain streamed exact-fp32 kernels: no stationary weights, no planes (GOPS_SK=0,0) */
#define GOPS_VF_STREAM_LAYER0 0x20u          /* stationary kernels keep layer 1 only                         (GOPS_SK=0,16) */
#define GOPS_VF_STATIONARY_ANY_BATCH 0x40u   /* stationary fp32 kernels also with more tiles than CUs        (GOPS_SK set) */
#define GOPS_VF_NO_SPLIT_STREAM0 0x80u       /* no plane-split kernel for policies with 129 .. 256 inputs    (GOPS_SPLIT_STREAM0=0) */
#define GOPS_VF_SPLIT_TAIL_MULTI 0x100u      /* stationary plane-split kernels also for tail + more tiles than CUs (GOPS_SPLIT_TAIL_MULTI) */
#define GOPS_VF_NO_HALF_TILE64 0x200u        /* GOPS_DTYPE_F16: the 16-trajectory-tile kernels instead of the 64-row ones (A/B) */
#define GOPS_VF_NO_NARROW_LDS 0x400u         /* narrow policies on the streamed fp32 kernels: hidden-layer weights from L2 every step, not LDS-resident (GOPS_NARROW=0; A/B) */
#define GOPS_VF_NO_NARROW_N64 0x800u         /* obs-64-64-act policies on the generic narrow kernels, not the ones written out for that shape (GOPS_N64=0; A/B, identical results) */
#define GOPS_VF_DW_EXACT 0x10000u            /* weight-gradient GEMM: exact three-plane bf16 split           (GOPS_DW_EXACT) */
#define GOPS_VF_DW_F32 0x20000u              /*   fp32-MFMA GEMM                                              (GOPS_DW_F32) */
#define GOPS_VF_DW_NO_GUARD 0x40000u         /*   test knob: no exact redo of saturated blocks                (GOPS_DW_NOGUARD) */
#define GOPS_VF_DW_NO_SKINNY 0x80000u        /*   no 16-input-layer kernel                                    (GOPS_DW_SKINNY=0) */
#define GOPS_VF_DW_NO_SPEC 0x100000u         /*   no wave-specialised kernel                                  (GOPS_DW_SPEC=0) */
#define GOPS_VF_DW_DIRECT 0x200000u          /*   register-direct kernel for the large layers too             (GOPS_DW_DIRECT) */
#define GOPS_VF_NO_FUSED_DWOUT 0x400000u     /*   output layer's gradient in its own pass                     (GOPS_NO_FUSED_DWOUT) */
#define GOPS_VF_NO_FUSED_DW0 0x4000000u      /*   GOPS_DTYPE_F16, 64-row kernels: the first layer's gradient by its GEMM, not inside the sweep (A/B) */
#define GOPS_VF_BWD_UPLOAD 0x800000u         /*   measurement: parameter upload launch in front of the sweep  (GOPS_BWD_UPLOAD) */
/* A backward call in two halves, so that a data-parallel caller can put the all-reduce of the gradients that are ready
 * first on a side stream while the rest is still being formed (gops_amd/trainer/grad_sync.py):
 *   PHASE_A: sweep, output layer's gradient and every hidden layer's EXCEPT the first one's (reduced and final in the caller's
 *            buffers when the call's work completes); PHASE_B (same description, same workspace, after PHASE_A): the first
 *            hidden layer's weight gradient + bias from the stash PHASE_A left.  Neither / both bits: the whole backward. */
#define GOPS_VF_BWD_PHASE_A 0x1000000u
#define GOPS_VF_BWD_PHASE_B 0x2000000u

#define GOPS_VARIANT_SPLIT 1
#define GOPS_VARIANT_STATIONARY_F32 2
#define GOPS_VARIANT_STREAMED_SPLIT_FWD 4
#define GOPS_VARIANT_HALF_TILE64 8   /* ABI v10: GOPS_DTYPE_F16 on the 64-trajectory-tile kernels (GOPS_ENV_LQ / _NONE, 256-wide nets) */
int gops_rollout_variant(const GopsRolloutDesc* desc);

/* Timing hook for bench.py: average duration in ms of the named internal kernel over the
 * launches recorded since the last reset (HIP events on the launch stream).  kernel ids:
 * 0 = forward rollout, 1 = backward sweep, 2 = weight-gradient GEMMs (+ reduce) of gops_rollout_*;
 * 3, 4, 5 = the same three of gops_value_forward / _backward (plain MLP batches, INFADP's V(o)). */
void gops_profile_enable(int32_t on);
void gops_profile_reset(void);
int gops_profile_read(int32_t kernel_id, double* avg_ms, int64_t* launches);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GOPS_HIP_H */
