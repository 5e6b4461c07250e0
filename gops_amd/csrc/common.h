// Shared device-side definitions for the gfx950 rollout kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gops_hip.h"

#define TB GOPS_TILE      // trajectories per workgroup tile = MFMA M
#define NTHREADS 256      // 4 wavefronts of 64
#define DW_SC_HOST 32     // samples per staged chunk of the dW GEMM (== DW_SC in aux_kernels.hip)
#define DW_OUT_SPLITS 1024  // sample splits of the output-layer weight gradient
#define ENV_STASH 16      // floats of per-(t,b) env stash: [0..3] abar, [4] done_t, [5..10] state_t

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- device-side descriptors (passed by value as kernel arguments) -------------------------
struct MlpDev {
    int nl;                       // Linear layers
    int dims[GOPS_MAX_LAYERS + 1];
    int kp[GOPS_MAX_LAYERS];      // input width of layer j padded to a multiple of 16
    int act;
    const float* w[GOPS_MAX_LAYERS];    // torch layout [out][in]
    const float* b[GOPS_MAX_LAYERS];
    const f32x4* wp[GOPS_MAX_LAYERS];   // forward MFMA-fragment packing of layer j (hidden layers)
    const f32x4* wpt[GOPS_MAX_LAYERS];  // backward (transposed operand) packing of layer j
};

struct StashDev {
    float* x;                         // [S][kp0]  policy input rows (obs_t | t+1 | 0-pad)
    float* h[GOPS_MAX_LAYERS];        // h[j] (j=1..L): [S][dims[j]] hidden activations
    float* z[GOPS_MAX_LAYERS];        // pre-activations, only for GELU
    float* d[GOPS_MAX_LAYERS];        // d[j] (j=1..L): [S][dims[j]] adjoint of z_j
    float* dy;                        // [S][4] adjoint of the head pre-activation
    float* env;                       // [S][ENV_STASH]
    float* tail_h[GOPS_MAX_LAYERS];   // [B][dims[j]] hidden activations of the tail value net
    float* tail_z[GOPS_MAX_LAYERS];
    float* tail_done;                 // [B] done flag after the last step
};

struct RolloutParams {
    int B, H, fh, need_grad, tail;
    int ldx, ldh;                     // LDS leading dims: input tile / hidden tiles (floats)
    GopsEnv env;
    MlpDev pol, val;
    StashDev st;
    GopsRolloutIn in;
    GopsRolloutOut out;
    const float* grad_v;              // backward only
    const float* ref_table;           // veh: [B][P+1+H][4]
    float gpow[GOPS_MAX_HORIZON + 1]; // gamma^t rounded from double
};

// ---- activations ---------------------------------------------------------------------------
#define SELU_SCALE 1.0507009873554804934193349852946f
#define SELU_ALPHA 1.6732632423543772848170429916717f

__device__ __forceinline__ float act_fwd(int kind, float z) {
    switch (kind) {
        case GOPS_ACT_RELU: return fmaxf(z, 0.f);
        case GOPS_ACT_ELU: return z > 0.f ? z : expm1f(z);
        case GOPS_ACT_GELU: return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
        case GOPS_ACT_SELU: return SELU_SCALE * (z > 0.f ? z : SELU_ALPHA * expm1f(z));
        case GOPS_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
        case GOPS_ACT_TANH: return tanhf(z);
        default: return z;
    }
}

// derivative act'(z); `h` is the stashed activation act(z), `z` only valid for GELU
__device__ __forceinline__ float act_bwd(int kind, float h, float z) {
    switch (kind) {
        case GOPS_ACT_RELU: return h > 0.f ? 1.f : 0.f;
        case GOPS_ACT_ELU: return h > 0.f ? 1.f : h + 1.f;
        case GOPS_ACT_GELU: {
            const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
            return cdf + z * pdf;
        }
        case GOPS_ACT_SELU: return h > 0.f ? SELU_SCALE : h + SELU_SCALE * SELU_ALPHA;
        case GOPS_ACT_SIGMOID: return h * (1.f - h);
        case GOPS_ACT_TANH: return 1.f - h * h;
        default: return 1.f;
    }
}

// ---- wrapper chain on the action (ScaleActionModel -> ClipActionModel) -----------------------
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ float wrap_action(const GopsEnv& e, int i, float abar) {
    const float a1 = clampf(abar, e.min_action[i], e.max_action[i]);
    const float a2 = e.act_low[i] + (e.act_high[i] - e.act_low[i]) *
                                        ((a1 - e.min_action[i]) / (e.max_action[i] - e.min_action[i]));
    const float a3 = clampf(a2, e.act_low[i], e.act_high[i]);
    return clampf(a3, e.act_low[i], e.act_high[i]);
}

// adjoint of wrap_action w.r.t. abar given the adjoint of the wrapped action
__device__ __forceinline__ float wrap_action_bwd(const GopsEnv& e, int i, float abar, float g) {
    const float a1 = clampf(abar, e.min_action[i], e.max_action[i]);
    const float a2 = e.act_low[i] + (e.act_high[i] - e.act_low[i]) *
                                        ((a1 - e.min_action[i]) / (e.max_action[i] - e.min_action[i]));
    if (!(a2 >= e.act_low[i] && a2 <= e.act_high[i])) g = 0.f;   // both clamps see the same range
    g = g * (e.act_high[i] - e.act_low[i]) / (e.max_action[i] - e.min_action[i]);
    if (!(abar >= e.min_action[i] && abar <= e.max_action[i])) g = 0.f;
    return g;
}

// ((x + pi) mod 2pi) - pi with Python/torch remainder semantics (gops/utils/math_utils.py:8-11)
__device__ __forceinline__ float angle_normalize(float x) {
    const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
    float r = fmodf(x + pi, two_pi);
    if (r < 0.f) r += two_pi;
    return r - pi;
}

// ---- MFMA tile GEMM: acc[j] (16 x 16 tile nt0+j) += A[16 x 16*kchunks] * Wp ------------------
// A lives in LDS row-major with leading dimension lda (floats, lda % 4 == 0); lane l supplies
// A[m = l&15][16c + 4*(l>>4) + i] to the i-th v_mfma_f32_16x16x4_f32 of chunk c, and the packed
// operand holds the matching B values so that one dwordx4 load per lane feeds four MFMAs.
template <int NT>
__device__ __forceinline__ void mfma_gemm(const float* __restrict__ A, int lda, int kchunks,
                                          const f32x4* __restrict__ Wp, int nt0, int lane,
                                          f32x4 (&acc)[NT]) {
    const float* arow = A + (lane & 15) * lda + 4 * (lane >> 4);
    const f32x4* wbase = Wp + (size_t)nt0 * kchunks * 64 + lane;
    f32x4 bcur[NT], bnxt[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bcur[j] = wbase[(size_t)j * kchunks * 64];
    for (int c = 0; c < kchunks; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * c);
        const int cn = (c + 1 < kchunks) ? c + 1 : c;
#pragma unroll
        for (int j = 0; j < NT; ++j) bnxt[j] = wbase[((size_t)j * kchunks + cn) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bcur[j][i], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) bcur[j] = bnxt[j];
    }
}

// One dense layer on the tile: out tiles are dealt to the 4 waves in contiguous groups, each wave
// walks its group 4 / 2 / 1 MFMA n-tiles at a time and hands every finished 16x16 accumulator
// (rows 4*(lane>>4)+r, column 16*ntile + (lane&15)) to `epi(acc, ntile)`.
template <class Epi>
__device__ __forceinline__ void gemm_layer(const float* A, int lda, int kch, int nt_tot,
                                           const f32x4* Wp, int tid, Epi&& epi) {
    const int lane = tid & 63, wave = tid >> 6;
    const int per = (nt_tot + 3) >> 2;
    int nt = wave * per;
    const int nt_end = min(nt_tot, nt + per);
    while (nt < nt_end) {
        const int left = nt_end - nt;
        if (left >= 4) {
            f32x4 acc[4] = {};
            mfma_gemm<4>(A, lda, kch, Wp, nt, lane, acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) epi(acc[q], nt + q);
            nt += 4;
        } else if (left >= 2) {
            f32x4 acc[2] = {};
            mfma_gemm<2>(A, lda, kch, Wp, nt, lane, acc);
#pragma unroll
            for (int q = 0; q < 2; ++q) epi(acc[q], nt + q);
            nt += 2;
        } else {
            f32x4 acc[1] = {};
            mfma_gemm<1>(A, lda, kch, Wp, nt, lane, acc);
            epi(acc[0], nt);
            nt += 1;
        }
    }
}

// Copy a [TB][ncols] LDS tile (leading dim ld) to global rows g[(row0+m)*ncols ...], coalesced.
__device__ __forceinline__ void stash_tile(const float* lds, int ld, int ncols, float* g, size_t row0,
                                           int nrows_valid, int tid) {
    const int vec_per_row = ncols >> 2;   // ncols % 4 == 0
    for (int idx = tid; idx < TB * vec_per_row; idx += NTHREADS) {
        const int m = idx / vec_per_row, c4 = idx - m * vec_per_row;
        if (m < nrows_valid) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lds + m * ld + 4 * c4);
            *reinterpret_cast<f32x4*>(g + (row0 + m) * ncols + 4 * c4) = v;
        }
    }
}
