// Per-trajectory env-model arithmetic (device), forward and hand-derived adjoints.
//   pyth_lq            gops/env/env_ocp/resources/lq_base.py:89-141,343-354
//   pyth_idpendulum    gops/env/env_ocp/env_model/pyth_idpendulum_model.py:31-172,199-216
//   pyth_veh3dofconti  gops/env/env_ocp/env_model/pyth_veh3dofconti_model.py:25-61,147-203
#pragma once
#include "common.h"

// ================================ pyth_lq =====================================================
// x' = inv_IA (x + dt B u);  r = rs * (rsh - (sum Q x^2 + sum R u^2)) on the CURRENT x.
// `e` is a PADDED description (common.h: lq_pad_env, applied by the host code to its own copy): inv_IA / B with the fixed row
// strides GOPS_MAX_LQ_STATE / GOPS_MAX_ACT, everything beyond (n, m) zero, and the callers' x / u / adjoints are zero there
// too - so every loop has a compile-time trip count AND compile-time addresses: the constants arrive as a few wide scalar
// loads instead of one s_load per run-time index (measured in the 64-row half kernels: the env phase of one wave was
// 5.3 k of the forward's 27 k cycles per step and 11 k of the sweep's 34 k with the n x n / n x m layout of the ABI).
// N, M: compile-time bounds of the state / action loops (>= the env's n, m; the arrays keep their maximal sizes) for callers
// that know the dimensions (the 64-row half kernels special-case n = 4, m = 2).
template <int N = GOPS_MAX_LQ_STATE, int M = GOPS_MAX_ACT>
__device__ __forceinline__ void lq_forward(const GopsEnv& e, const float* x, const float* u,
                                           float* xn, float& r) {
    float tmp[GOPS_MAX_LQ_STATE];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float bu = 0.f;
#pragma unroll
        for (int j = 0; j < M; ++j) bu += e.lq_B[i * GOPS_MAX_ACT + j] * u[j];
        tmp[i] = bu * e.lq_dt + x[i];
    }
    float rs = 0.f, ra = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) acc += e.lq_inv_IA[i * GOPS_MAX_LQ_STATE + k] * tmp[k];
        xn[i] = acc;
        rs += x[i] * x[i] * e.lq_Q[i];
    }
#pragma unroll
    for (int j = 0; j < M; ++j) ra += u[j] * u[j] * e.lq_R[j];
    r = e.lq_reward_scale * (e.lq_reward_shift - 1.0f * (rs + ra));
}

// adjoints: gxn (adjoint of x'), gr (adjoint of r) -> gx (accumulated), gu (overwritten)
template <int N = GOPS_MAX_LQ_STATE, int M = GOPS_MAX_ACT>
__device__ __forceinline__ void lq_backward(const GopsEnv& e, const float* x, const float* u,
                                            const float* gxn, float gr, float* gx, float* gu) {
    float gt[GOPS_MAX_LQ_STATE];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) acc += e.lq_inv_IA[i * GOPS_MAX_LQ_STATE + k] * gxn[i];
        gt[k] = acc;
        gx[k] += acc + gr * e.lq_reward_scale * (-2.f * e.lq_Q[k] * x[k]);
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) acc += e.lq_B[i * GOPS_MAX_ACT + j] * gt[i];
        gu[j] = acc * e.lq_dt + gr * e.lq_reward_scale * (-2.f * e.lq_R[j] * u[j]);
    }
}

// ================================ gym_cartpoleconti ===========================================
// gops/env/env_gym/env_model/gym_cartpoleconti_model.py:101-129: Euler step (dt = 0.02) of the cart-pole with
// force = 10 a; done on the NEXT state (|x| > 2.4 or |theta| > 12 deg); reward = 1 - done (no gradient).
// Python-double constants reach the fp32 tensors rounded once, like torch does.
struct CartConst { float g, M, mp, L, pml, fmag, dt, xth, thth; };
__device__ __forceinline__ CartConst cart_const() {
    CartConst c;
    c.g = 9.8f; c.M = (float)(0.1 + 1.0); c.mp = 0.1f; c.L = 0.5f; c.pml = (float)(0.1 * 0.5); c.fmag = 10.f; c.dt = 0.02f;
    c.xth = 2.4f; c.thth = (float)(12 * 2 * 3.14159265358979323846 / 360);
    return c;
}
__device__ __forceinline__ void cart_forward(const CartConst& C, const float* x, float a, float* xn, float& r, bool& done) {
    float sth, cth;
    sincosf(x[2], &sth, &cth);
    const float force = C.fmag * a;
    const float temp = (force + C.pml * x[3] * x[3] * sth) / C.M;
    const float thacc = (C.g * sth - cth * temp) / (C.L * (4.0f / 3.0f - C.mp * cth * cth / C.M));
    const float xacc = temp - C.pml * thacc * cth / C.M;
    xn[0] = x[0] + C.dt * x[1];
    xn[1] = x[1] + C.dt * xacc;
    xn[2] = x[2] + C.dt * x[3];
    xn[3] = x[3] + C.dt * thacc;
    done = (xn[0] < -C.xth) || (xn[0] > C.xth) || (xn[2] < -C.thth) || (xn[2] > C.thth);
    r = done ? 0.f : 1.f;
}
// adjoint of the state part: gxn (adjoint of x') -> gx (accumulated), ga (overwritten); the reward has no gradient
__device__ __forceinline__ void cart_backward(const CartConst& C, const float* x, float a, const float* gxn, float* gx, float& ga) {
    float s, c;
    sincosf(x[2], &s, &c);
    const float thd = x[3];
    const float temp = (C.fmag * a + C.pml * thd * thd * s) / C.M;
    const float D = C.L * (4.0f / 3.0f - C.mp * c * c / C.M), N = C.g * s - c * temp, thacc = N / D;
    const float dtemp_dth = C.pml * thd * thd * c / C.M, dtemp_dthd = 2.f * C.pml * thd * s / C.M, dtemp_dF = 1.f / C.M;
    const float dN_dth = C.g * c + s * temp - c * dtemp_dth, dD_dth = C.L * (2.f * C.mp * c * s / C.M);
    const float dA_dth = (dN_dth * D - N * dD_dth) / (D * D);        // thetaacc
    const float dA_dthd = -c * dtemp_dthd / D, dA_dF = -c * dtemp_dF / D;
    const float k = C.pml / C.M;
    const float dX_dth = dtemp_dth - k * (dA_dth * c - thacc * s);   // xacc
    const float dX_dthd = dtemp_dthd - k * c * dA_dthd, dX_dF = dtemp_dF - k * c * dA_dF;
    gx[0] += gxn[0];
    gx[1] += gxn[1] + C.dt * gxn[0];
    gx[2] += gxn[2] + C.dt * (gxn[1] * dX_dth + gxn[3] * dA_dth);
    gx[3] += gxn[3] + C.dt * gxn[2] + C.dt * (gxn[1] * dX_dthd + gxn[3] * dA_dthd);
    ga = C.fmag * C.dt * (gxn[1] * dX_dF + gxn[3] * dA_dF);
}

// ================================ gym_pendulum ================================================
// gops/env/env_gym/env_model/gym_pendulum_model.py:72-115: obs = (cos th, sin th, thdot); th = arccs(sin, cos) with the
// 0.9999 guard; thdot' = clamp(thdot + (-15 sin(th + pi) + 3 a) dt, +-8) (the UNCLAMPED value advances th), dt = 0.05;
// reward = -(angle_normalize(th)^2 + 0.1 thdot^2 + 0.001 a^2) on the CURRENT state; never done.
struct PendStep { float th, nw_raw, nth; };
__device__ __forceinline__ float pend_th(float costh, float sinth) {
    const float pi = 3.14159265358979323846f;
    const float t = acosf(0.9999f * costh);
    return (sinth > 0.f) ? t : (2.f * pi - t);
}
__device__ __forceinline__ void pend_forward(const float* x, float a, float* xn, float& r, PendStep& w) {
    const float pi = 3.14159265358979323846f;
    w.th = pend_th(x[0], x[1]);
    w.nw_raw = x[2] + (-15.f * sinf(w.th + pi) + 3.f * a) * 0.05f;
    w.nth = w.th + w.nw_raw * 0.05f;
    sincosf(w.nth, &xn[1], &xn[0]);
    xn[2] = fminf(fmaxf(w.nw_raw, -8.f), 8.f);
    const float an = angle_normalize(w.th);
    r = -(an * an + 0.1f * (x[2] * x[2]) + 0.001f * (a * a));
}
// adjoints: gxn (adjoint of x'), gr (adjoint of r) -> gx (accumulated), ga (overwritten)
__device__ __forceinline__ void pend_backward(const float* x, float a, const float* gxn, float gr, float* gx, float& ga) {
    const float pi = 3.14159265358979323846f;
    float xn[3], r;
    PendStep w;
    pend_forward(x, a, xn, r, w);
    const float g_nth = -xn[1] * gxn[0] + xn[0] * gxn[1];
    const float g_raw = g_nth * 0.05f + ((w.nw_raw >= -8.f && w.nw_raw <= 8.f) ? gxn[2] : 0.f);
    const float g_th = g_nth + g_raw * (-15.f * cosf(w.th + pi)) * 0.05f + gr * (-2.f * angle_normalize(w.th));
    const float u = 0.9999f * x[0];
    const float dth_dc = ((x[1] > 0.f) ? -1.f : 1.f) * 0.9999f / sqrtf(1.f - u * u);
    gx[0] += g_th * dth_dc;
    gx[2] += g_raw + gr * (-0.2f * x[2]);
    ga = g_raw * (3.f * 0.05f) + gr * (-0.002f * a);
}

// ================================ pyth_idpendulum =============================================
#ifndef GOPS_IDP_FAST
#define GOPS_IDP_FAST 1
#endif
struct IdpConst {   // products of the Python-double constants, rounded once like torch does
    float a, b, e, f, h, k, gb, ge, l1, l2;
};
__device__ __forceinline__ IdpConst idp_const() {
    const double m = 9.42477796, m1 = 4.1033127, m2 = 4.1033127, l1 = 0.6, l2 = 0.6, g = 9.81;
    IdpConst c;
    c.a = (float)(m + m1 + m2);
    c.b = (float)(l1 * (0.5 * m1 + m2));
    c.e = (float)(0.5 * m2 * l2);
    c.f = (float)(l1 * l1 * (0.3333 * m1 + m2));
    c.h = (float)(0.5 * l1 * l2 * m2);
    c.k = (float)(0.3333 * l2 * l2 * m2);
    c.gb = (float)(g * (0.5 * m1 + m2) * l1);
    c.ge = (float)(g * 0.5 * l2 * m2);
    c.l1 = (float)l1;
    c.l2 = (float)l2;
    return c;
}

struct IdpSub {     // intermediates of one sub-step that the adjoint re-uses
    float s1, c1, s2, c2, s12, c12;
    float inv[6];   // symmetric M^-1: 00 01 02 11 12 22
    float qdd[3];
};

// One explicit-Euler sub-step.  FIRST: the sin / cos of theta1, theta2 are evaluated here (sincosf); otherwise
// w.s1 .. w.c2 hold them on entry.  On exit w_next (may alias w) holds the intermediates of THIS sub-step and, in
// its s1 .. c2, the sin / cos of the NEXT state's angles: theta' = theta + tau * theta_dot is a rotation by
// d = tau * theta_dot (|d| < 0.1 for any state the pendulum reaches), so sin / cos advance with a 5th / 4th order
// series of d (error < 2e-11) instead of two more sincosf per sub-step - 2 instead of 15 libm calls per env step;
// sin / cos(theta1 - theta2) follow from the angle-difference identities.  All within fp32 round-off (1e-7) of the
// direct evaluation, far inside the reference's own 1e-5 step tolerance.
__device__ __forceinline__ void idp_rotate(float& sn_, float& cs_, float d) {
    const float d2 = d * d;
    const float sd = d * (1.f + d2 * (-1.f / 6.f + d2 * (1.f / 120.f)));
    const float cd = 1.f + d2 * (-0.5f + d2 * (1.f / 24.f));
    const float s0 = sn_, c0 = cs_;
    sn_ = s0 * cd + c0 * sd;
    cs_ = c0 * cd - s0 * sd;
}

template <bool FIRST = true>
__device__ __forceinline__ void idp_substep(const IdpConst& C, const float* s, float u, float tau,
                                            float* sn, IdpSub& w) {
    const float th1d = s[4], th2d = s[5];
    if (FIRST) {
        sincosf(s[1], &w.s1, &w.c1);
        sincosf(s[2], &w.s2, &w.c2);
    }
    w.s12 = w.s1 * w.c2 - w.c1 * w.s2;
    w.c12 = w.c1 * w.c2 + w.s1 * w.s2;
    const float m00 = C.a, m01 = C.b * w.c1, m02 = C.e * w.c2, m11 = C.f, m12 = C.h * w.c12, m22 = C.k;
    const float f0 = C.b * (th1d * th1d) * w.s1 + C.e * (th2d * th2d) * w.s2 + u;
    const float f1 = -C.h * (th2d * th2d) * w.s12 + C.gb * w.s1;
    const float f2 = C.h * (th1d * th1d) * w.s12 + C.ge * w.s2;
    const float c00 = m11 * m22 - m12 * m12;
    const float c01 = m02 * m12 - m01 * m22;
    const float c02 = m01 * m12 - m02 * m11;
    const float c11 = m00 * m22 - m02 * m02;
    const float c12 = m01 * m02 - m00 * m12;
    const float c22 = m00 * m11 - m01 * m01;
    const float det = m00 * c00 + m01 * c01 + m02 * c02;
#if GOPS_IDP_FAST
    // hardware reciprocal + one Newton step (<= 1 ulp of the IEEE quotient; det is 0.1 .. 2, never scaled): 3 instead of 10
    // instructions of the sub-step's dependent chain
    float rdet = __builtin_amdgcn_rcpf(det);
    rdet = rdet * (2.f - det * rdet);
#else
    const float rdet = 1.f / det;
#endif
    w.inv[0] = c00 * rdet; w.inv[1] = c01 * rdet; w.inv[2] = c02 * rdet;
    w.inv[3] = c11 * rdet; w.inv[4] = c12 * rdet; w.inv[5] = c22 * rdet;
    w.qdd[0] = w.inv[0] * f0 + w.inv[1] * f1 + w.inv[2] * f2;
    w.qdd[1] = w.inv[1] * f0 + w.inv[3] * f1 + w.inv[4] * f2;
    w.qdd[2] = w.inv[2] * f0 + w.inv[4] * f1 + w.inv[5] * f2;
    sn[0] = s[0] + tau * s[3];
    sn[1] = s[1] + tau * s[4];
    sn[2] = s[2] + tau * s[5];
    sn[3] = s[3] + tau * w.qdd[0];
    sn[4] = s[4] + tau * w.qdd[1];
    sn[5] = s[5] + tau * w.qdd[2];
}
// sin / cos of the angles after the sub-step whose intermediates are in `w` (input state s): for the next sub-step
__device__ __forceinline__ void idp_advance_trig(const float* s, float tau, const IdpSub& w, IdpSub& nxt) {
    float s1 = w.s1, c1 = w.c1, s2 = w.s2, c2 = w.c2;
    idp_rotate(s1, c1, tau * s[4]);
    idp_rotate(s2, c2, tau * s[5]);
    nxt.s1 = s1; nxt.c1 = c1; nxt.s2 = s2; nxt.c2 = c2;
}

// adjoint of one sub-step: g (adjoint of sn, overwritten with adjoint of s), gu accumulated
__device__ __forceinline__ void idp_substep_bwd(const IdpConst& C, const float* s, float tau,
                                                const IdpSub& w, float* g, float& gu) {
    const float th1d = s[4], th2d = s[5];
    const float q0 = tau * g[3], q1 = tau * g[4], q2 = tau * g[5];       // adjoint of qdd
    const float fb0 = w.inv[0] * q0 + w.inv[1] * q1 + w.inv[2] * q2;    // adjoint of f = M^-1 q
    const float fb1 = w.inv[1] * q0 + w.inv[3] * q1 + w.inv[4] * q2;
    const float fb2 = w.inv[2] * q0 + w.inv[4] * q1 + w.inv[5] * q2;
    // Mbar_ij = -fb_i qdd_j ; only the cos-dependent entries matter
    const float cb1 = -C.b * (fb0 * w.qdd[1] + fb1 * w.qdd[0]);
    const float cb2 = -C.e * (fb0 * w.qdd[2] + fb2 * w.qdd[0]);
    const float cb12 = -C.h * (fb1 * w.qdd[2] + fb2 * w.qdd[1]);
    gu += fb0;
    float th1db = fb0 * C.b * 2.f * th1d * w.s1 + fb2 * C.h * 2.f * th1d * w.s12;
    float th2db = fb0 * C.e * 2.f * th2d * w.s2 - fb1 * C.h * 2.f * th2d * w.s12;
    const float sb1 = fb0 * C.b * th1d * th1d + fb1 * C.gb;
    const float sb2 = fb0 * C.e * th2d * th2d + fb2 * C.ge;
    const float sb12 = -fb1 * C.h * th2d * th2d + fb2 * C.h * th1d * th1d;
    const float th1b = -w.s1 * cb1 + w.c1 * sb1 - w.s12 * cb12 + w.c12 * sb12;
    const float th2b = -w.s2 * cb2 + w.c2 * sb2 + w.s12 * cb12 - w.c12 * sb12;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    g[1] = g1 + th1b;
    g[2] = g2 + th2b;
    g[3] = g[3] + tau * g0;
    g[4] = g[4] + tau * g1 + th1db;
    g[5] = g[5] + tau * g2 + th2db;
}

__device__ __forceinline__ float idp_reward(const float* s, float a) {
    const float dist = 0.f * (s[0] * s[0]) + 5.f * (s[1] * s[1]) + 10.f * (s[2] * s[2]);
    const float vel = 0.5f * (s[3] * s[3]) + 0.5f * (s[4] * s[4]) + 1.f * (s[5] * s[5]);
    return 10.f - dist - vel - 1.f * (a * a);
}

__device__ __forceinline__ bool idp_done(const IdpConst& C, const float* s) {
    const float tip_y = C.l1 * cosf(s[1]) + C.l2 * cosf(s[2]);
    return (tip_y <= 1.0f) || (fabsf(s[0]) >= 15.f);
}
// the same test with the cosines of the new angles at hand (idp_advance_trig of the last sub-step: within 3e-7 of cosf, the
// reference's own torch.cos is within 1 ulp of it): two rotations instead of two libm cosf on the step's dependent chain
__device__ __forceinline__ bool idp_done_trig(const IdpConst& C, const float* s, float c1, float c2) {
    const float tip_y = C.l1 * c1 + C.l2 * c2;
    return (tip_y <= 1.0f) || (fabsf(s[0]) >= 15.f);
}

// ================================ pyth_mobilerobot ============================================
// gops/env/env_ocp/env_model/pyth_mobilerobot_model.py:24-213.  State = observation, 13 columns:
//   [0..4] ego (x, y, theta, v, w)   [5..7] tracking errors (e_y, e_theta, e_v) of the NEW ego state   [8..12] obstacle (x, y, theta, v, w)
// Robot.f_xu (:129-181), T = 0.2:  dv = clamp(v_cmd - v, +-1.8 T), dw = clamp(w_cmd - w, +-0.8 T),
//   vc = clamp(v + dv, +-0.4) + 0.5 n_v,  wc = clamp(w + dw, +-pi/2) + 0.5 n_w,  (x, y, theta)' = (x + T cos(theta) vc, y + T sin(theta) vc, theta + T wc),
//   (v, w)' = (vc, wc).  The ego is driven by the action with n = 0; the obstacle by ITS OWN (v, w) as commands (dv = dw = 0 with
//   unit derivative through `v + dv`) and the caller's draws n ~ N(0, 0.03), N(0, 0.02).
// The reference path is y = 0 sin(x / 3), phi = arctan(0 cos(x / 3)) (:199-213): tracking errors (y', theta', v' - 0.3).
// constraint = (0.37 + 0.37 + 0.15) - |obstacle' - ego'| (:84-95); reward = -1.4 e_y^2 - e_theta^2 - 16 e_v^2 - 0.2 a_0^2 - 0.5 a_1^2 (:98-104);
// done = x' < -2 | |y'| > 4 | constraint > 0.15 (:116-121).  Clamp derivatives as torch.clamp: 1 on the closed interval.
struct MobConst { float T, dv_max, dw_max, v_max, w_max, safe_dis, margin; };
__device__ __forceinline__ MobConst mob_const() {
    MobConst c;
    c.T = 0.2f;
    c.dv_max = (float)(1.8 * 0.2);
    c.dw_max = (float)(0.8 * 0.2);
    c.v_max = 0.4f;
    c.w_max = (float)(3.14159265358979323846 / 2);
    c.safe_dis = (float)(0.74 / 2 + 0.74 / 2 + 0.15);
    c.margin = 0.15f;
    return c;
}

struct MobStep {   // intermediates shared by forward and adjoint
    float sth, cth, vc, wc;        // ego: sin / cos of the heading, saturated commands
    float soth, coth, ovc, owc;    // obstacle
    float dx, dy, dist;            // obstacle' - ego'
    bool m_dv, m_vc, m_dw, m_wc, m_ov, m_ow;   // clamp pass-through masks
};

__device__ __forceinline__ bool mob_inside(float v, float lim) { return v >= -lim && v <= lim; }

// clip_heading: the DATA env's Robot.f_xu (env_ocp/pyth_mobilerobot.py:271-311) clips both new headings to +-pi (the model
// does not); everything behind it - tracking error, constraint, reward, termination test - is the model's arithmetic
template <bool CLIP_HEADING = false>
__device__ __forceinline__ void mob_forward(const MobConst& C, const float* x, float a0, float a1, float nv, float nw,
                                            float* xn, float& r, float& c, bool& done, MobStep& w) {
    const float dvr = a0 - x[3], dwr = a1 - x[4];
    const float dv = clampf(dvr, -C.dv_max, C.dv_max), dw = clampf(dwr, -C.dw_max, C.dw_max);
    const float vs = x[3] + dv, ws = x[4] + dw;
    w.m_dv = mob_inside(dvr, C.dv_max); w.m_dw = mob_inside(dwr, C.dw_max);
    w.m_vc = mob_inside(vs, C.v_max);   w.m_wc = mob_inside(ws, C.w_max);
    w.vc = clampf(vs, -C.v_max, C.v_max);
    w.wc = clampf(ws, -C.w_max, C.w_max);
    sincosf(x[2], &w.sth, &w.cth);
    xn[0] = x[0] + (C.T * w.cth) * w.vc;
    xn[1] = x[1] + (C.T * w.sth) * w.vc;
    xn[2] = x[2] + C.T * w.wc;
    if (CLIP_HEADING) xn[2] = clampf(xn[2], -3.14159265358979323846f, 3.14159265358979323846f);
    xn[3] = w.vc;
    xn[4] = w.wc;
    xn[5] = xn[1];
    xn[6] = xn[2];
    xn[7] = xn[3] - 0.3f;
    w.m_ov = mob_inside(x[11], C.v_max); w.m_ow = mob_inside(x[12], C.w_max);
    w.ovc = clampf(x[11], -C.v_max, C.v_max) + nv * 0.5f;
    w.owc = clampf(x[12], -C.w_max, C.w_max) + nw * 0.5f;
    sincosf(x[10], &w.soth, &w.coth);
    xn[8] = x[8] + (C.T * w.coth) * w.ovc;
    xn[9] = x[9] + (C.T * w.soth) * w.ovc;
    xn[10] = x[10] + C.T * w.owc;
    if (CLIP_HEADING) xn[10] = clampf(xn[10], -3.14159265358979323846f, 3.14159265358979323846f);
    xn[11] = w.ovc;
    xn[12] = w.owc;
    w.dx = xn[8] - xn[0];
    w.dy = xn[9] - xn[1];
    w.dist = sqrtf(w.dx * w.dx + w.dy * w.dy);
    c = C.safe_dis - w.dist;
    const float r_track = -1.4f * (xn[5] * xn[5]) - 1.f * (xn[6] * xn[6]) - 16.f * (xn[7] * xn[7]);
    const float r_act = -0.2f * (a0 * a0) - 0.5f * (a1 * a1);
    r = r_track + r_act;
    done = (xn[0] < -2.f) || (fabsf(xn[1]) > 4.f) || (c > C.margin);
}

// gx += (d xn / d x)^T gxn + g_r d r / d x + g_c d c / d x;  ga = (d . / d a)^T (...)   (gx carries what the caller put there)
__device__ __forceinline__ void mob_backward(const MobConst& C, const float* x, float a0, float a1, float nv, float nw,
                                             const float* gxn_in, float g_r, float g_c, float* gx, float* ga) {
    float xn[MOB_OBS], r, c;
    bool done;
    MobStep w;
    mob_forward(C, x, a0, a1, nv, nw, xn, r, c, done, w);
    float g[MOB_OBS];
#pragma unroll
    for (int i = 0; i < MOB_OBS; ++i) g[i] = gxn_in[i];
    // constraint: c = safe_dis - sqrt(dx^2 + dy^2)
    const float inv = w.dist > 0.f ? 1.f / w.dist : 0.f;
    const float ux = w.dx * inv, uy = w.dy * inv;
    g[0] += g_c * ux;  g[1] += g_c * uy;
    g[8] -= g_c * ux;  g[9] -= g_c * uy;
    // reward on the new tracking errors, tracking errors on the new ego state
    g[5] += g_r * (-2.8f * xn[5]);
    g[6] += g_r * (-2.f * xn[6]);
    g[7] += g_r * (-32.f * xn[7]);
    g[1] += g[5];  g[2] += g[6];  g[3] += g[7];
    // ego
    const float gvc = g[3] + C.T * (w.cth * g[0] + w.sth * g[1]);
    const float gwc = g[4] + C.T * g[2];
    gx[0] += g[0];
    gx[1] += g[1];
    gx[2] += g[2] + C.T * w.vc * (-w.sth * g[0] + w.cth * g[1]);
    const float gvs = w.m_vc ? gvc : 0.f, gws = w.m_wc ? gwc : 0.f;
    const float gdv = w.m_dv ? gvs : 0.f, gdw = w.m_dw ? gws : 0.f;
    gx[3] += gvs - gdv;
    gx[4] += gws - gdw;
    ga[0] = gdv + g_r * (-0.4f * a0);
    ga[1] = gdw + g_r * (-1.f * a1);
    // obstacle
    const float govc = g[11] + C.T * (w.coth * g[8] + w.soth * g[9]);
    const float gowc = g[12] + C.T * g[10];
    gx[8] += g[8];
    gx[9] += g[9];
    gx[10] += g[10] + C.T * w.ovc * (-w.soth * g[8] + w.coth * g[9]);
    gx[11] += w.m_ov ? govc : 0.f;
    gx[12] += w.m_ow ? gowc : 0.f;
}

// ================================ pyth_veh2dofconti ===========================================
// gops/env/env_ocp/env_model/pyth_veh2dofconti_model.py:24-174 (vehicle parameters pyth_veh2dofconti.py:24-34, u = 5):
// linear 2-DOF lateral dynamics stepped in the ego frame (y = phi = 0), then placed back:
//   v' = ((m v) u + c1 w - (c2 steer) u - c3 w) / den_v,  w' = ((Iz w) u + c1 v - (c4 steer) u) / den_w,
//   y' = y + (u sin phi) dt + (dt v) cos phi,  phi' = angle_normalize(phi + dt w)
// with the Python-double coefficient products rounded once (as torch does when a scalar meets an fp32 tensor).
struct Veh2Const { float m, Iz, u, dt, c1, c2, c3, c4, den_v, den_w; };
__device__ __forceinline__ Veh2Const veh2_const() {
    const double k_f = -128915.5, k_r = -85943.6, l_f = 1.06, l_r = 1.85, m = 1412.0, I_z = 1536.7, u = 5.0, dt = 0.1;
    Veh2Const c;
    c.m = (float)m; c.Iz = (float)I_z; c.u = (float)u; c.dt = (float)dt;
    c.c1 = (float)(dt * (l_f * k_f - l_r * k_r));
    c.c2 = (float)(dt * k_f);
    c.c3 = (float)(dt * m * (u * u));
    c.c4 = (float)(dt * l_f * k_f);
    c.den_v = (float)(m * u - dt * (k_f + k_r));
    c.den_w = (float)(I_z * u - dt * (l_f * l_f * k_f + l_r * l_r * k_r));
    return c;
}
__device__ __forceinline__ void veh2_f_xu(const Veh2Const& C, const float* s, float steer, float sphi, float cphi, float* sn) {
    const float v = s[2], w = s[3];
    sn[2] = (C.m * v * C.u + C.c1 * w - C.c2 * steer * C.u - C.c3 * w) / C.den_v;
    sn[3] = (C.Iz * w * C.u + C.c1 * v - C.c4 * steer * C.u) / C.den_w;
    sn[0] = s[0] + C.u * sphi * C.dt + (C.dt * v) * cphi;
    sn[1] = angle_normalize(s[1] + C.dt * w);
}
// adjoint: ln (adjoint of the next state) -> l (adjoint of the state, overwritten), g_steer
__device__ __forceinline__ void veh2_f_xu_bwd(const Veh2Const& C, const float* s, float sphi, float cphi, const float* ln,
                                              float* l, float& g_steer) {
    const float v = s[2];
    l[0] = ln[0];
    l[1] = ln[0] * (C.u * C.dt * cphi - C.dt * v * sphi) + ln[1];
    l[2] = ln[0] * (C.dt * cphi) + ln[2] * (C.m * C.u / C.den_v) + ln[3] * (C.c1 / C.den_w);
    l[3] = ln[1] * C.dt + ln[2] * ((C.c1 - C.c3) / C.den_v) + ln[3] * (C.Iz * C.u / C.den_w);
    g_steer = ln[2] * (-C.c2 * C.u / C.den_v) + ln[3] * (-C.c4 * C.u / C.den_w);
}
__device__ __forceinline__ float veh2_reward(const float* o, float steer) {
    return -(0.04f * (o[0] * o[0]) + 0.02f * (o[1] * o[1]) + 0.01f * (o[2] * o[2]) + 0.01f * (o[3] * o[3]) + 0.01f * (steer * steer));
}

// ================================ pyth_veh3dofconti ===========================================
struct VehConst {
    float m, Iz, dt, c_lk, dt_kf, dt_m, den_v, dt_lfkf, den_w;
};
__device__ __forceinline__ VehConst veh_const() {
    const double k_f = -128915.5, k_r = -85943.6, l_f = 1.06, l_r = 1.85, m = 1412.0, I_z = 1536.7,
                 dt = 0.1;
    VehConst c;
    c.m = (float)m; c.Iz = (float)I_z; c.dt = (float)dt;
    c.c_lk = (float)(dt * (l_f * k_f - l_r * k_r));
    c.dt_kf = (float)(dt * k_f);
    c.dt_m = (float)(dt * m);
    c.den_v = (float)(dt * (k_f + k_r));
    c.dt_lfkf = (float)(dt * l_f * k_f);
    c.den_w = (float)(dt * (l_f * l_f * k_f + l_r * l_r * k_r));
    return c;
}

struct VehStep {   // intermediates shared by forward and adjoint
    float sphi, cphi, Dv, Dw, Nv, Nw;
};

__device__ __forceinline__ void veh_f_xu(const VehConst& C, const float* s, float steer, float ax,
                                         float* sn, VehStep& w) {
    // w.sphi / w.cphi = sin / cos of s[2], supplied by the caller (carried across rollout steps)
    const float x = s[0], y = s[1], phi = s[2], u = s[3], v = s[4], om = s[5];
    sn[0] = x + C.dt * (u * w.cphi - v * w.sphi);
    sn[1] = y + C.dt * (u * w.sphi + v * w.cphi);
    sn[2] = angle_normalize(phi + C.dt * om);
    sn[3] = u + C.dt * ax;
    w.Nv = C.m * v * u + C.c_lk * om - C.dt_kf * steer * u - C.dt_m * (u * u) * om;
    w.Dv = C.m * u - C.den_v;
    sn[4] = w.Nv / w.Dv;
    w.Nw = C.Iz * om * u + C.c_lk * v - C.dt_lfkf * steer * u;
    w.Dw = C.Iz * u - C.den_w;
    sn[5] = w.Nw / w.Dw;
}

// adjoint: lam (adjoint of sn) -> ls (adjoint of s), g_steer, g_ax
__device__ __forceinline__ void veh_f_xu_bwd(const VehConst& C, const float* s, float steer,
                                             const VehStep& w, const float* lam, float* ls,
                                             float& g_steer, float& g_ax) {
    const float u = s[3], v = s[4], om = s[5];
    const float lx = lam[0], ly = lam[1], lp = lam[2], lu = lam[3], lv = lam[4], lw = lam[5];
    // (adjoint only - no decision hangs on these, the reference's own autograd rounds differently anyway: the hardware reciprocal,
    // 1 ulp, instead of two IEEE divisions = twenty dependent instructions of the sweep's one-wave env phase)
    const float ivD = __builtin_amdgcn_rcpf(w.Dv), iwD = __builtin_amdgcn_rcpf(w.Dw);
    const float dv_du = (C.m * v - C.dt_kf * steer - 2.f * C.dt_m * u * om) * ivD - w.Nv * C.m * ivD * ivD;
    const float dw_du = (C.Iz * om - C.dt_lfkf * steer) * iwD - w.Nw * C.Iz * iwD * iwD;
    ls[0] = lx;
    ls[1] = ly;
    ls[2] = lp + lx * C.dt * (-u * w.sphi - v * w.cphi) + ly * C.dt * (u * w.cphi - v * w.sphi);
    ls[3] = lu + lx * C.dt * w.cphi + ly * C.dt * w.sphi + lv * dv_du + lw * dw_du;
    ls[4] = -lx * C.dt * w.sphi + ly * C.dt * w.cphi + lv * (C.m * u * ivD) + lw * (C.c_lk * iwD);
    ls[5] = lp * C.dt + lv * ((C.c_lk - C.dt_m * u * u) * ivD) + lw * (C.Iz * u * iwD);
    g_steer = lv * (-C.dt_kf * u * ivD) + lw * (-C.dt_lfkf * u * iwD);
    g_ax = lu * C.dt;
}

__device__ __forceinline__ float veh_reward(const float* o, float steer, float ax) {
    return -(0.04f * (o[0] * o[0]) + 0.04f * (o[1] * o[1]) + 0.02f * (o[2] * o[2]) +
             0.02f * (o[3] * o[3]) + 0.01f * (o[5] * o[5]) + 0.01f * (steer * steer) + 0.01f * (ax * ax));
}

// ================================ veh3dofconti + surrounding vehicles ==========================
// pyth_veh3dofconti_surrcstr_model.py:28-39 (SurrVehicleModel), :98-148 (get_constraint);
// pyth_veh3dofconti_detour_model.py:115-151 (road-boundary terms), :153-173 (reward weights).
__device__ __forceinline__ f32x4 surr_next(const f32x4& p, float delta) {   // (x, y, phi, u); delta, u constant
    float s, c;
    sincosf(p[2], &s, &c);
    f32x4 n;
    n[0] = p[0] + p[3] * c * 0.1f;
    n[1] = p[1] + p[3] * s * 0.1f;
    n[2] = angle_normalize(p[2] + p[3] * tanf(delta) / 3.0f * 0.1f);
    n[3] = p[3];
    return n;
}

struct SurrCstr {   // constraint values and their derivatives w.r.t. the ego pose (x, y, phi)
    float c[GOPS_MAX_CONSTRAINT];
    float dx[GOPS_MAX_CONSTRAINT], dy[GOPS_MAX_CONSTRAINT], dphi[GOPS_MAX_CONSTRAINT];
};

// `surr`: the n_surr (x, y, phi, u) points of the SAME time as the ego pose (x, y, phi).
template <bool WANT_GRAD>
__device__ __forceinline__ void surr_constraint(const GopsEnv& e, float x, float y, float sphi, float cphi,
                                                const f32x4* surr, SurrCstr& o) {
    const float d = (e.veh_length - e.veh_width) / 2.f;
    const float r = 0.70710678118654752440f * e.veh_width;
    const float ex[2] = {x + d * cphi, x - d * cphi}, ey[2] = {y + d * sphi, y - d * sphi};
    float best = 3.402823466e+38f, bvx = 0.f, bvy = 0.f, bsign = 1.f;
    for (int k = 0; k < e.n_surr; ++k) {
        float ss, sc;
        sincosf(surr[k][2], &ss, &sc);
        const float sx[2] = {surr[k][0] + d * sc, surr[k][0] - d * sc}, sy[2] = {surr[k][1] + d * ss, surr[k][1] - d * ss};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float vx = ex[i] - sx[j], vy = ey[i] - sy[j];
                const float dist = sqrtf(vx * vx + vy * vy);
                if (dist < best) { best = dist; bvx = vx; bvy = vy; bsign = i == 0 ? 1.f : -1.f; }
            }
    }
    o.c[0] = 2.f * r - best;
    if (WANT_GRAD) {
        const float inv = best > 0.f ? 1.f / best : 0.f;
        o.dx[0] = -bvx * inv;
        o.dy[0] = -bvy * inv;
        o.dphi[0] = -(bvx * (-bsign * d * sphi) + bvy * (bsign * d * cphi)) * inv;
    }
    if (e.n_constraint >= 3) {   // detour: road boundaries on the circles' y extent
        const int iu = ey[0] >= ey[1] ? 0 : 1;            // max over the two circles of (y_i + r - upper)
        o.c[1] = ey[iu] + r - e.road_upper;
        const int il = ey[0] <= ey[1] ? 0 : 1;            // max over the two circles of (lower - (y_i - r))
        o.c[2] = e.road_lower - (ey[il] - r);
        if (WANT_GRAD) {
            o.dx[1] = 0.f; o.dy[1] = 1.f; o.dphi[1] = (iu == 0 ? 1.f : -1.f) * d * cphi;
            o.dx[2] = 0.f; o.dy[2] = -1.f; o.dphi[2] = -(il == 0 ? 1.f : -1.f) * d * cphi;
        }
    }
}

__device__ __forceinline__ float veh_reward_w(const float* w, const float* o, float steer, float ax) {
    return -(w[0] * (o[0] * o[0]) + w[1] * (o[1] * o[1]) + w[2] * (o[2] * o[2]) + w[3] * (o[3] * o[3]) +
             w[4] * (o[5] * o[5]) + w[5] * (steer * steer) + w[6] * (ax * ax) + w[7] * (o[4] * o[4]));
}

// SPIL's constraint-to-cost map (gops/algorithm/spil.py:224-232): Phi(y) = (1 + tau m1) / (1 + m2 tau exp(clamp(y / tau, -10, 5)))
// with m1 = 1, m2 = m1 / (1 + m1) * 0.9 = 0.45, tau = 0.07; dlog = Phi'(y) / Phi(y).
__device__ __forceinline__ float spil_phi(float y, float& dlog) {
    const float z = y / 0.07f, zc = fminf(fmaxf(z, -10.f), 5.f);
    const float be = (0.45f * 0.07f) * expf(zc);
    dlog = (z > -10.f && z < 5.f) ? -(be / (1.f + be)) / 0.07f : 0.f;
    return (1.f + 0.07f * 1.f) / (1.f + be);
}

// Collision penalty of pyth_veh3dofconti_surrcstr_penalty_model.py:145-160 as a function of the constraint value
// c = 2 r - min distance (dis = -c):  pen = 15 (tanh(max(8 + 16 c, 0) - 4) + 1), and d pen / d c.
__device__ __forceinline__ float surr_penalty(float c, float& dpen_dc) {
    const float z = fmaxf(8.f + 16.f * c, 0.f);          // 8 - 8 dis / 0.5
    const float th = tanhf(z - 4.f);
    dpen_dc = (z > 0.f) ? 15.f * (1.f - th * th) * 16.f : 0.f;
    return 15.f * (th + 1.f);
}
