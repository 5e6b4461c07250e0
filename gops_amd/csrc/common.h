// Shared device-side definitions for the gfx950 rollout kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gops_hip.h"

// NOTE (round 4): this library is compiled with the `packed-fp32-ops` subtarget feature OFF (csrc/Makefile: NOPK) - a gfx950
// hazard between dependent v_pk_*_f32 instructions that hipcc's SLP vectoriser forms out of plain scalar code made the streamed
// plane-split kernels run-to-run non-deterministic (DESIGN_LOG.md, round 4, reproducer tools/microbench/pk_hazard.hip).
// tests/test_host_cpu.py disassembles the built library and fails on any v_pk_{mul,add,fma}_f32.

#define TB GOPS_TILE      // trajectories per workgroup tile = MFMA M
#define NTHREADS 256      // 4 wavefronts of 64
#define DW_SC_HOST 32     // samples per staged chunk of the dW GEMM (== DW_SC in aux_kernels.hip)
#define DW_OUT_SPLITS 1024  // sample splits of the output-layer weight gradient
#define MOB_OBS 13        // pyth_mobilerobot: observation = state columns
// LDS "reference points" (x 4 TB floats) of the idpendulum sweeps: one [TB][5][24] parking, or - plane-split stationary sweep -
// the two parity halves of the staged [TB][IDP_PARK] parking the forward wrote
#define IDP_POINTS(split) ((split) ? 64 : 32)
#define IDP_PARK 128      // floats of the idpendulum sub-step parking per (t, b): [k][24] (state, sin / cos, M^-1, qdd), [120..125] final state
#define ENV_STASH 16      // floats of per-(t,b) env stash: [0..3] abar, [4] done_t, [5..10] state_t, [12..15] veh3dof: sin, cos of the heading before / after the step

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));   // one A / B fragment of v_mfma_f32_16x16x32_f16
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));    // one A / B fragment of v_mfma_f32_16x16x32_bf16
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Pointers that reach the kernels through the parameter block are generic to the compiler, and
// generic accesses become FLAT instructions, which count on lgkmcnt as well as vmcnt: every LDS wait
// would then also wait for the pending stash stores.  gptr() re-types them as global (addrspace 1).
#define GLOBAL_AS __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ GLOBAL_AS T* gptr(T* p) { return (GLOBAL_AS T*)p; }
template <class T> __device__ __forceinline__ const GLOBAL_AS T* gptr(const T* p) { return (const GLOBAL_AS T*)p; }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const GLOBAL_AS float* p) { return *(const GLOBAL_AS f32x4*)p; }

// models that read a reference-trajectory table [B][P + 1 + H][4] = (x, y, phi, u) built by the prologue
__host__ __device__ inline bool env_has_ref_table(int kind) {
    return kind == GOPS_ENV_VEH3DOFCONTI || kind == GOPS_ENV_VEH3DOF_SURR || kind == GOPS_ENV_VEH2DOF;
}

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
    // GOPS_DTYPE_F16: half-precision fragment packings (rollout_f16.h) and the input width padded to 32
    int kp32[GOPS_MAX_LAYERS];
    const f16x8* wph[GOPS_MAX_LAYERS];
    const f16x8* wpth[GOPS_MAX_LAYERS];
};

// Plane-split contractions of the register-stationary kernels (policy: obs -> 256 -> 256 -> act, fp32 results): every fp32
// operand of a hidden-layer contraction is carried as two 16-bit planes (2 + 2 bytes, the register footprint of the fp32 value)
// and the products run on 16-cycle v_mfma_f32_16x16x32_* with fp32 accumulation instead of 8 x 32-cycle v_mfma_f32_16x16x4_f32
// per 32-deep block.  WHICH planes: the GOPS_SPLIT_F16X2 block below (default: two half planes per operand, 3 MFMAs; the
// round-3 form - w = bf16(w) + f16 residual, a = three exact bf16 planes + one half plane, 4 MFMAs - stays behind the switch).
// The member types name the round-3 planes; the F16X2 packing stores half bit patterns in both (gemm_split bit-casts).
struct SplitDev {
    int on;                     // 1: the stationary kernels run the plane-split contractions (else fp32 MFMA)
    int kc[2];                  // 32-wide k-chunks of hidden layer j's input
    const bf16x8* w1[2];        // forward B operand [n-tile][chunk][lane]: bf16(W_j[16 nt + (lane & 15)][slot feature])
    const f16x8* r[2];          //   residual plane, scaled per n-tile
    const float* inv[2];        //   [n-tile] 1 / s_r
    const bf16x8* w1t[2];       // backward (delta_{j+1} W_j): n-tiles over the layer's INPUT features, chunks over its outputs
    const f16x8* rt[2];
    const float* invt[2];
    float* out_part;            // backward: per-workgroup partials of the output layer's weight gradient [grid][A][K] (null: the
    float* out_part_b;          //   separate dw_out pass forms it), and of its bias gradient [grid][A]
};
// Plane-split operands of ALL hidden layers of one net, forward orientation, for the streamed-split forward kernels (SS):
// every layer's planes stream from L2 (StreamQ), layer 0 padded to kc[0] in {1, 2, 4, 8} chunks of 32 inputs.
struct SplitNetDev {
    int kc[GOPS_MAX_LAYERS - 1];
    const bf16x8* w1[GOPS_MAX_LAYERS - 1];
    const f16x8* r[GOPS_MAX_LAYERS - 1];
    const float* inv[GOPS_MAX_LAYERS - 1];
};
__host__ __device__ inline int ss_kc0(int kp32) { const int c = kp32 >> 5; return c <= 1 ? 1 : (c <= 2 ? 2 : (c <= 4 ? 4 : 8)); }
#ifndef GOPS_PIN_MODE
#define GOPS_PIN_MODE 2   // layer-1 bf16 planes of the split kernels pinned to AGPRs (StatQ PIN; modes 1 / 2 / 3 measured within 1 %, r03)
#endif
// ---- which plane split the contractions run on ----------------------------------------------------------------------------
// GOPS_SPLIT_F16X2 = 1 (round 5, default): BOTH operands as two half planes.
//     w s_w = wh + wl / 2^11   with wh = f16(w s_w) (round to nearest), wl = f16((w s_w - wh) 2^11), s_w the power of two that brings
//                           the n-tile's largest |w| into [2^13, 2^14) (packing kernel): 22 significant bits - the systematic part
//                           of the error (the same perturbed network for every sample: it does not average out over the batch)
//                           is 2^-22 |w| against 2^-20 |w| of the bf16 + f16 pair;
//     a s = ah + al / 2^11  with ah = f16(a s), al = f16((a s - ah) 2^11): 2^-22 |a|, independent from sample to sample;
//     a w = [ah wh] + [al wh + ah wl] / 2^11   - THREE v_mfma_f32_16x16x32_f16 per 32-deep block and n-tile (two accumulators),
//                           the dropped al wl term is 2^-22 |a w|; products of two halfs are exact in the fp32 accumulator.
//   s: a power of two that keeps the planes inside the half range: SPLIT_FWD_SA in the forward (below), per tile and step from
//   max|delta_y| in the sweep (as before).  Nothing is clamped: a value beyond the range converts to inf, every conversion
//   records it (split2h's `ovf`) and the kernels poison the tile's results with NaN at tile end - an overflow is LOUD.
//   Against the bf16x3 + f16 form: 3 instead of 4 matrix instructions, 2 instead of 4 plane images in LDS (half the plane
//   stores and A-fragment reads), ~5 instead of ~8 VALU instructions per element split.
// GOPS_SPLIT_F16X2 = 0: the round-3 form (three exact bf16 planes of the activation, bf16 + scaled f16 planes of the weight).
#ifndef GOPS_SPLIT_F16X2
#define GOPS_SPLIT_F16X2 1
#endif
#if GOPS_SPLIT_F16X2
// forward scale of the activation planes: 2^-4 puts the top of the half range at |a| = 1.05e6 and keeps 22 bits down to
// |a| = 1e-3 (below that the two planes still resolve 2^-32 ABSOLUTE: hi and lo both run into half subnormals, lo's at 2^-35 / s).
// Beyond the range the conversion yields inf and the rollout returns non-finite values - loudly, like the reference's own overflow,
// only earlier; the algorithm classes' PrecisionGuard treats a non-finite distance as exceeded and moves to the exact-fp32 rollout kernels.
#define SPLIT_FWD_SA 0.0625f
#define SPLIT_LO_SCALE 2048.0f   // 2^11: the residual planes of both operands
#else
#define SPLIT_FWD_SA 0.015625f  // forward: af = f16(a / 64): the correction term saturates only beyond |a| = 4.2e6; below |a| = 4e-3 af is a
                                // subnormal half (absolute error 4e-6 * 2^-8 |w| per term: under the fp32 rounding of a unit-sized term)
#endif
// forward pre-activation z = a W + b from the two accumulators of a plane-split contraction (activation planes scaled by SPLIT_FWD_SA)
__device__ __forceinline__ float split_preact(float acc, float accr, float inv, float bias) {
#if GOPS_SPLIT_F16X2
    return fmaf(fmaf(accr, 1.f / SPLIT_LO_SCALE, acc), inv * (1.f / SPLIT_FWD_SA), bias);   // inv = 1 / (the n-tile's weight scale)
#else
    return fmaf(accr, inv * (1.f / SPLIT_FWD_SA), acc) + bias;
#endif
}
// result of a plane-split contraction from its two accumulators: `inv` = 1 / (the n-tile's weight scale; bf16 + f16 form: of its
// residual plane), `inv_s` = 1 / (scale the caller put on the activation planes)
__device__ __forceinline__ float split_combine(float acc, float accr, float inv, float inv_s) {
#if GOPS_SPLIT_F16X2
    return fmaf(accr, 1.f / SPLIT_LO_SCALE, acc) * (inv * inv_s);   // both planes of both operands carry their scale
#else
    return fmaf(accr, inv * inv_s, acc);      // only the half plane of the activation is scaled
#endif
}
// position e of a hidden tile's plane row (the order plane_store writes, = the contraction order of the next GEMM) -> feature
__host__ __device__ inline int split_perm(int e) { return 64 * (e >> 6) + 16 * (e & 3) + ((e & 63) >> 2); }
// LDS image of a [TB][K] activation tile: 4 planes (a1, a2, a3 bf16; af f16), each [TB] rows of rowb bytes (16 bytes of pad)
__host__ __device__ inline int split_rowb(int K) { return 2 * K + 16; }
__host__ __device__ inline int split_tile_floats(int K) { return TB * split_rowb(K); }   // 4 planes x TB x rowb bytes

// fp32 stash tensors are FEATURE-MAJOR inside every 16-sample tile: element (sample tile q, feature n, row m) of a
// tensor with N features sits at (q * N + n) * 16 + m (FM layout).  A lane of the rollout kernels' MFMA result layout
// (feature n = lane & 15, rows 4 (lane >> 4) .. +3) then stores / loads ONE 16-byte vector, a wave's n-tile is 1 KiB
// of contiguous memory, and the weight-gradient GEMM - whose contraction index is the sample - reads its MFMA
// operand fragments (consecutive samples of one feature) straight from global memory with no transpose.
struct StashDev {
    float* x;                         // FM [S/16][kp0][16]  policy input rows (obs_t | t+1 | 0-pad)
    float* h[GOPS_MAX_LAYERS];        // h[j] (j=1..L): FM [S/16][dims[j]][16] hidden activations
    float* z[GOPS_MAX_LAYERS];        // only for GELU: gelu'(z_j) (formed by the forward epilogue together with gelu(z_j))
    float* d[GOPS_MAX_LAYERS];        // d[j] (j=1..L): FM [S/16][dims[j]][16] adjoint of z_j
    float* dy;                        // [S][4] adjoint of the head pre-activation
    float* env;                       // [S][ENV_STASH]
    float* tail_h[GOPS_MAX_LAYERS];   // FM [ceil(B/16)][dims[j]][16] hidden activations of the tail value net
    float* tail_z[GOPS_MAX_LAYERS];
    float* tail_done;                 // [B] done flag after the last step
    // GOPS_DTYPE_F16: x / h / z / d / tail_h / tail_z are ROW-major [S][width] _Float16 (x rows are kp32[0] wide, z
    // holds act'(z) instead of z), and the first 8 observation columns are kept in fp32 for the env adjoints:
    float* xf;                        // [S][8]
    // pyth_idpendulum on the plane-split stationary kernels: the forward parks the intermediates of the five Euler sub-steps
    // (IDP_PARK floats per row: 5 x 24 + the state after the step) so that the sweep does not recompute them
    float* idp;                       // [S][IDP_PARK] or null
};

struct RolloutParams {
    int B, H, fh, need_grad, tail;
    int tail_unmasked;                // SPIL: the terminal value is added for finished trajectories too
    int ldx, ldh;                     // LDS leading dims: input tile / hidden tiles (floats)
    int touch_mode;                   // backward L2 warm-up: 0 off, 1 all at the step top, 2 spread over the step
    GopsEnv env;
    MlpDev pol, val;
    StashDev st;
    GopsRolloutIn in;
    GopsRolloutOut out;
    const float* grad_v;              // backward only
    int open_loop;                    // 1: head outputs come from in.head_pre, the MLP phases are skipped
    float* g_head_pre;                // open-loop backward: d(loss)/d(head_pre) [B][H][A]
    const float* ext_delta;           // gops_mlp_backward (GOPS_ENV_NONE): adjoint of the LAST hidden activation [S][dims[L]],
                                      // taken instead of the head's (delta_y W_o) product; the head then has no gradient
    const float* adj_gfo;             // EXT backward kernels (gops_rollout_backward_adj): d(loss)/d(final_obs) [B][obs_dim] or null
    float* adj_gobs;                  //   out: d(loss)/d(obs_0) [B][obs_dim] or null
    int adj_first_only;               //   1: the delta stash is written at step 0 only (zeroed beforehand)
    int ext;                          //   1: launch the EXT instantiation
    const float* ref_table;           // veh: [B][P+1+H][4]
    const f32x4* surr_table;          // GOPS_ENV_VEH3DOF_SURR: [B][H+1][n_surr] (x, y, phi, u) of every surrounding vehicle after t steps
    unsigned long long* dbg;          // debug: per-phase cycle counters of block 0 (GOPS_DBG_TIMING)
    int f16;                          // 1: GOPS_DTYPE_F16
    float* gscale;                    // f16 backward: gscale[0] = max|grad_v| of the launch (upload_params_kernel)
    SplitDev sp;                      // plane-split contractions of the stationary fp32 kernels
    int ss;                           // 1: streamed-split FORWARD kernel (all hidden layers 256 wide, planes streamed from L2)
    int tail_fp32;                    //   ... with the TAIL value net on exact fp32 products (relu / selu nets that keep a gradient: rollout_fwd.hip)
    SplitNetDev ssp, ssv;             //   planes of the policy / the tail value net
    SplitNetDev sspt, ssvt;           //   streamed-split SWEEP (ssb): transposed planes (n-tiles over a layer's inputs, 8 chunks over its outputs)
    int ssb;
    unsigned vflags;                  // GOPS_VF_* of the description (| the debug override of the process environment, read once at load)
    int dw_wgs;                       // target workgroup count of a weight-gradient GEMM
    int h64;                          // 1: GOPS_DTYPE_F16 launch on the 64-trajectory-tile kernels (rollout_h64.hip): stash rows in 64-row tiles
    int narrow;                       // 1 / 2: plain streamed fp32 kernels with the packed hidden-layer weights of the POLICY resident in LDS (narrow nets;
                                      //    2: obs -> 64 -> 64 -> act, the N64 instantiations:
                                      //    all of them <= NARROW_MAX_FLOATS; forward: wp[0 .. L-1], sweep: wpt[L-1 .. 0], narrow_floats each)
    int narrow_floats, narrow_off_fwd, narrow_off_bwd;   //    size of the image, its offset (floats from the start of dynamic LDS) in either kernel
    float gpow[GOPS_MAX_HORIZON + 1]; // gamma^t rounded from double
};

// What a backward launch knows that the forward's device copy of RolloutParams does not: handed to the sweep by value
// (pointers and two ints: scalar registers straight from the kernel-argument segment), so that no upload launch has to
// sit between the forward pass and the sweep.  The fp32 sweeps also fold max|grad_v| into gscale[0] themselves (one
// atomicMax per tile) - the weight-gradient GEMMs read it after the sweep; the half sweeps need it BEFORE they start and
// keep the upload kernel.
struct BwdPatch {
    GopsRolloutIn in;                 // the backward call's inputs (grad_constraint*, noise, head_pre, ...)
    const float* grad_v;
    float* g_head_pre;
    const float* ext_delta;
    const float* adj_gfo;
    float* adj_gobs;
    int adj_first_only;
    int pad_;
    float* out_part;
    float* out_part_b;
    unsigned long long* dbg;
    // gops_rollout_backward_update: the Adam step that rides on this call's reduce launch.  ONE thread of the sweep advances the
    // device-resident optimizer state and leaves the step's scalar factors in `ad_snap` (adam_snapshot below) - the ~1500 blocks of
    // the reduce then read three floats instead of each taking a ticket on the state (1500 atomics on one address: 15 us, measured)
    GopsAdamState* ad_st;
    float* ad_snap;
    double ad_b1, ad_b2;
    // 64-row half sweep (rollout_h64.hip), policy input of <= 8 columns (pyth_lq): the first layer's weight / bias gradient is
    // formed INSIDE the sweep, one slab [256][8] / [256] per workgroup - its delta tile never goes to the stash and the layer's
    // GEMM launch disappears (null: the GEMM path)
    float* w0_part;
    float* w0_part_b;
};
// adam_kernel's scalar factors of THIS step (formed in double like torch's host code, rounded to fp32) -> snap[0 .. 2] = step_size,
// sqrt(1 - beta2^t), grad_scale; the state moves on to the next step
__device__ __forceinline__ void adam_snapshot(GopsAdamState* st, float* snap, double beta1, double beta2) {
    const double b1p = st->beta1_pow * beta1, b2p = st->beta2_pow * beta2;   // beta^t, t = step + 1
    snap[0] = (float)(st->lr / (1.0 - b1p));
    snap[1] = (float)sqrt(1.0 - b2p);
    snap[2] = (float)st->grad_scale;
    st->step += 1;
    st->beta1_pow = b1p;
    st->beta2_pow = b2p;
}

// A gradient element that is not finite takes NO optimizer step (ABI v13): parameter, both moments and - in the fused tail - the
// Polyak target of that element stay as they are, and GopsAdamState::skipped_nonfinite counts it.  The plane-split kernels
// answer a half-range overflow with NaN gradients on purpose (a loud failure instead of a wrong number); without this gate one
// such update would write NaN into every weight and target before the host - which reads losses lazily - could react.
__device__ __forceinline__ bool grad_is_finite(float g) { return (__float_as_uint(g) & 0x7f800000u) != 0x7f800000u; }

// upload_params_kernel / prologue_kernel take the block BY VALUE: it has to fit the 4 KiB kernel-argument segment
static_assert(sizeof(RolloutParams) <= 4000, "RolloutParams outgrew the kernel-argument segment (move gpow[] out)");


// per-phase cycle accounting of block 0 / thread 0 (debug builds of the timing knob only)
#ifdef GOPS_DBG_BUILD   // make -C gops_amd/csrc DBG=1 : in-kernel phase timing (costs ~32 VGPRs)
struct DbgClock {
    bool on;
    long long acc[16], last;
    __device__ __forceinline__ void init(bool enable) {
        on = enable;
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        last = enable ? clock64() : 0;
    }
    __device__ __forceinline__ void tick(int i) {
        if (on) {
            const long long now = clock64();
            acc[i] += now - last;
            last = now;
        }
    }
    __device__ __forceinline__ void dump(unsigned long long* out) const {
        if (on)
            for (int i = 0; i < 16; ++i) gptr(out)[i] = (unsigned long long)acc[i];
    }
};
#else
struct DbgClock {
    __device__ __forceinline__ void init(bool) {}
    __device__ __forceinline__ void tick(int) {}
    __device__ __forceinline__ void dump(unsigned long long*) const {}
};
#endif
#define DBG_TICK(i) dbg.tick(i);

// Batched split-K reductions: job i sums `splits` partial slabs [rows][ld] into out[rows][cols].
struct ReduceJobs {
    int n;
    int block0[2 * GOPS_MAX_LAYERS + 1];   // first block of each job; block0[n] = total
    const float* part[2 * GOPS_MAX_LAYERS];
    float* out[2 * GOPS_MAX_LAYERS];
    int splits[2 * GOPS_MAX_LAYERS], rows[2 * GOPS_MAX_LAYERS], cols[2 * GOPS_MAX_LAYERS], ld[2 * GOPS_MAX_LAYERS];
    int slab_rows[2 * GOPS_MAX_LAYERS];    // rows of one split's slab (>= rows: the slab of a padded output layer has more)
    const float* unscale;                  // f16: device pointer to max|grad_v| (RolloutParams::gscale), else null
    const unsigned* poison;                // fp32 launches: RolloutParams::gscale + 3, the overflow mark of this call's forward / sweep (null: none)
    float* reset;                          // fp32 launches: RolloutParams::gscale, zeroed here for the NEXT backward call (nothing in this
                                           // kernel reads it, and every consumer of this call - sweep, weight-gradient GEMMs - is done)
    // gops_rollout_backward_update (ABI v12): the Adam step on every gradient element as this kernel forms it, and the loss mean
    // in one extra block (block0[n])
    float* ad_p[2 * GOPS_MAX_LAYERS];      // parameter / first / second moment tensor behind out[j] (null: no step for job j)
    float* ad_m[2 * GOPS_MAX_LAYERS];
    float* ad_v[2 * GOPS_MAX_LAYERS];
    const float* ad_snap;                  // this step's scalar factors (adam_snapshot, written by the sweep kernel of the same call); null: no optimizer step
    unsigned* ad_skipped;                  // GopsAdamState::skipped_nonfinite of the stepped optimizer (ABI v13)
    double ad_b1, ad_b2;
    float ad_eps;
    int mean_n;
    const float* mean_x;                   // null: no loss mean
    float* mean_stats;
    float mean_sc;
    float* pk_t[2 * GOPS_MAX_LAYERS];      // Polyak target tensor behind out[j] (null: none): averaged with the new parameter value
    float pk_omt, pk_tau;                  // (1 - tau, tau as gops_polyak_update rounds them)
};

// GOPS_DTYPE_F16 backward: the power of two s that brings max|grad_v| = m into [1, 2).  The whole sweep runs
// on s * grad_v (half deltas then sit mid-range: 2^15 of headroom above - conversions saturate - and normals
// down to 2^-14 of the largest) and the reduce kernel multiplies the parameter gradients by 1/s (exact).
__device__ __forceinline__ float f16_grad_scale(float m) {
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
    int ex;
    (void)frexpf(m, &ex);          // m = f * 2^ex, f in [0.5, 1)
    int sh = 1 - ex;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    return ldexpf(1.f, sh);
}

// Host side: the ABI's row-major n x n / n x m matrices -> fixed row strides, zero padded (in place, on the library's own copy)
inline void lq_pad_env(GopsEnv& e) {
    if (e.kind != GOPS_ENV_LQ) return;
    const int n = e.obs_dim, m = e.act_dim;
    float ia[GOPS_MAX_LQ_STATE * GOPS_MAX_LQ_STATE] = {}, b[GOPS_MAX_LQ_STATE * GOPS_MAX_ACT] = {};
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < n; ++k) ia[i * GOPS_MAX_LQ_STATE + k] = e.lq_inv_IA[i * n + k];
        for (int j = 0; j < m; ++j) b[i * GOPS_MAX_ACT + j] = e.lq_B[i * m + j];
    }
    for (int i = 0; i < GOPS_MAX_LQ_STATE * GOPS_MAX_LQ_STATE; ++i) e.lq_inv_IA[i] = ia[i];
    for (int i = 0; i < GOPS_MAX_LQ_STATE * GOPS_MAX_ACT; ++i) e.lq_B[i] = b[i];
    for (int i = n; i < GOPS_MAX_LQ_STATE; ++i) e.lq_Q[i] = 0.f;
    for (int j = m; j < GOPS_MAX_ACT; ++j) e.lq_R[j] = 0.f;
}

// LDS copy of the env description (kernels whose env phases read ~100 of its scalars per step: read through the parameter
// pointer they do not fit the SGPR file - hundreds of lane spills per step; from LDS they are broadcast reads with immediate
// offsets).  Used by the 64-row half kernels and the pyth_lq instantiations of the streamed plane-split kernels.
#define ENV_LDS_FLOATS ((int)((sizeof(GopsEnv) + 15) / 16) * 4)
__host__ __device__ constexpr bool env_in_lds(int env_kind, bool streamed_split) { return streamed_split && env_kind == GOPS_ENV_LQ; }

// A uniform value of the parameter block, pinned to scalar registers for the duration of a kernel.  The rollout kernels read
// their description through a `const RolloutParams&` in global memory; under register pressure hipcc does not keep such values -
// it RELOADS them where they are used (s_load_dword + s_waitcnt lgkmcnt(0), which also drains the LDS queue): 12 - 22 scalar
// round trips per rollout step in the round-4 plane-split kernels.  Passing the value through an empty asm makes it opaque: it
// can no longer be rematerialised from memory and stays in an SGPR (or a VGPR lane) for the whole step loop.
template <bool ON = true, class T>
__device__ __forceinline__ T keep_s(T v) {
    if constexpr (ON) asm volatile("" : "+s"(v));
    return v;
}
// ---- activations ---------------------------------------------------------------------------
#define SELU_SCALE 1.0507009873554804934193349852946f
#define SELU_ALPHA 1.6732632423543772848170429916717f

// gelu(z) and gelu'(z) from ONE exponential: Phi(-|z|) = erfc(|z| / sqrt 2) / 2 by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 on erfc: fp32 round-off class, far inside the 1e-4 parity bar) and
// phi(z) = exp(-z^2/2) / sqrt(2 pi) with the same exponential.  ~17 branch-free VALU ops for both values (libm's
// erff + expf are ~100 with divergent branches, and were the largest VALU item of the GELU workloads).  The forward
// epilogue stashes gelu'(z) in the Z tensor, so the backward sweep evaluates no transcendental at all.
__device__ __forceinline__ void gelu_pair(float z, float& h, float& dh) {
    const float x = fabsf(z) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
    const float e = __expf(-0.5f * z * z);
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float q = (0.5f * t) * poly * e;       // Phi(-|z|)
    const float cdf = z < 0.f ? q : 1.f - q;
    h = z * cdf;
    dh = fmaf(z * 0.39894228040143267794f, e, cdf);
}

// The same pair for the HALF-precision kernels (results are rounded to 2^-11 anyway): Phi(-|z|) = exp(-z^2/2) P7(|z|) with P7 the
// degree-7 least-squares fit of the Mills ratio Phi(-x) / exp(-x^2/2) on [0, 6] under the weight exp(-x^2/2) (|error| <= 1.6e-5
// on Phi, 1.2e-5 on gelu, evaluated in fp32 Horner form; numpy: chebvander + lstsq, converted to monomials).  No reciprocal:
// one v_exp_f32 + 17 full-rate instructions, ~21 issue slots against ~26 - the forward of the 64-row half kernels is bound
// by exactly this arithmetic.
__device__ __forceinline__ void gelu_pair_h(float z, float& h, float& dh) {
    const float x = fminf(fabsf(z), 6.f);
    const float e = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170368f);   // exp(-z^2 / 2)
    float p = fmaf(x, -1.45366924e-04f, 2.07320246e-03f);
    p = fmaf(p, x, -1.29176200e-02f);
    p = fmaf(p, x, 4.79239046e-02f);
    p = fmaf(p, x, -1.23755032e-01f);
    p = fmaf(p, x, 2.46921233e-01f);
    p = fmaf(p, x, -3.98505880e-01f);
    p = fmaf(p, x, 4.99984820e-01f);
    const float q = p * e;                       // Phi(-|z|)
    const float cdf = z < 0.f ? q : 1.f - q;
    h = z * cdf;
    dh = fmaf(z * 0.39894228040143267794f, e, cdf);
}

// ELU / SELU negative branch: exp(min(z,0)) - 1 on the hardware exponential (v_exp_f32, ~1 ulp of a
// value <= 1, i.e. absolute error <= 1.2e-7 - fp32 round-off class, far inside the 1e-4 parity
// bar) and written as max(z,0) + (e - 1) so that no lane diverges: for z > 0, e == 1 exactly.
template <int ACT>
__device__ __forceinline__ float act_fwd_t(float z) {
    if (ACT == GOPS_ACT_RELU) return fmaxf(z, 0.f);
    if (ACT == GOPS_ACT_ELU) return fmaxf(z, 0.f) + (__expf(fminf(z, 0.f)) - 1.f);
    if (ACT == GOPS_ACT_GELU) { float h, dh; gelu_pair(z, h, dh); return h; }
    if (ACT == GOPS_ACT_SELU) return SELU_SCALE * (fmaxf(z, 0.f) + SELU_ALPHA * (__expf(fminf(z, 0.f)) - 1.f));
    if (ACT == GOPS_ACT_SIGMOID) return 1.f / (1.f + expf(-z));
    if (ACT == GOPS_ACT_TANH) return tanhf(z);
    return z;
}

// tanh on the hardware exponential: (1 - t) / (1 + t) with t = exp(-2|x|) away from zero (relative error ~5e-7 worst
// case, just above the switch point), the odd series up to x^7 below |x| = 1/8 (truncation 2e-10); branch-free.  libm's
// tanhf is ~200 instructions with divergent branches and sat on the critical path of every step (head -> action).
__device__ __forceinline__ float fast_tanh(float x) {
#ifdef GOPS_EXACT_TANH   // A/B knob (make variant VFLAGS=-DGOPS_EXACT_TANH): libm's tanhf in the plane-split kernels too
    return tanhf(x);
#endif
    const float ax = fabsf(x), x2 = x * x;
    const float t = __expf(-2.f * ax);
    const float big = (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
    float poly = fmaf(x2, -17.f / 315.f, 2.f / 15.f);
    poly = fmaf(poly, x2, -1.f / 3.f);
    poly = fmaf(poly, x2, 1.f);
    return ax < 0.125f ? x * poly : copysignf(big, x);
}

// derivative act'(z); `h` is the stashed activation act(z); for GELU `z` is the stashed DERIVATIVE (see gelu_pair)
template <int ACT>
__device__ __forceinline__ float act_bwd_t(float h, float z) {
    if (ACT == GOPS_ACT_RELU) return h > 0.f ? 1.f : 0.f;
    if (ACT == GOPS_ACT_ELU) return h > 0.f ? 1.f : h + 1.f;
    if (ACT == GOPS_ACT_GELU) return z;
    if (ACT == GOPS_ACT_SELU) return h > 0.f ? SELU_SCALE : h + SELU_SCALE * SELU_ALPHA;
    if (ACT == GOPS_ACT_SIGMOID) return h * (1.f - h);
    if (ACT == GOPS_ACT_TANH) return 1.f - h * h;
    return 1.f;
}

// Run `body.template operator()<ACT>()` for the runtime activation id: the switch is taken once per
// layer, the per-element loops inside are branch-free.
template <class F>
__device__ __forceinline__ void act_dispatch(int kind, F&& body) {
    switch (kind) {
        case GOPS_ACT_RELU: body.template operator()<GOPS_ACT_RELU>(); break;
        case GOPS_ACT_ELU: body.template operator()<GOPS_ACT_ELU>(); break;
        case GOPS_ACT_GELU: body.template operator()<GOPS_ACT_GELU>(); break;
        case GOPS_ACT_SELU: body.template operator()<GOPS_ACT_SELU>(); break;
        case GOPS_ACT_SIGMOID: body.template operator()<GOPS_ACT_SIGMOID>(); break;
        case GOPS_ACT_TANH: body.template operator()<GOPS_ACT_TANH>(); break;
        default: body.template operator()<GOPS_ACT_LINEAR>(); break;
    }
}

// cond ? a : b on two elements of per-thread arrays: written plainly, hipcc turns it into a select of the two ADDRESSES and
// a load - which keeps both arrays in scratch memory (measured: the pyth_lq env phase of the plane-split kernels at 6.7 k
// cycles per step instead of 0.7 k).  The empty asm pins both values to registers first.
__device__ __forceinline__ float sel_reg(bool c, float a, float b) {
    asm volatile("" : "+v"(a), "+v"(b));
    return c ? a : b;
}

// ---- wrapper chain on the action (ScaleActionModel -> ClipActionModel) -----------------------
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// Per-action constants staged in LDS ([GOPS_MAX_ACT][8]: sc, of, min_action, max_action, act_low,
// act_high) so that lane `a` of a 16-lane group can squash + wrap action `a` with a lane-dependent index
// (the parameter block is only addressable with wave-uniform indices): all actions of a trajectory are
// then processed in parallel lanes instead of one after the other.
struct ActC {
    float sc, of, min_action, max_action, act_low, act_high;   // abar = sc * tanh(y) + of, then the wrapper chain
    // 1 / (max_action - min_action) when that range is a power of two - the pipeline's [-1, 1]: 2 - else 0: dividing by a power of
    // two IS multiplying by its reciprocal, bit for bit, and an IEEE division is ten dependent instructions on the step's
    // critical path (the sweep's env phase carried four of them per step for two actions)
    float inv_range = 0.f;
};
__device__ __forceinline__ float exact_inverse_or_zero(float range) {
    const unsigned b = __float_as_uint(range);
    const bool pow2 = (b & 0x007fffffu) == 0u && range >= 9.313225746154785e-10f && range <= 1073741824.f;   // 2^-30 .. 2^30: no subnormal results
    return pow2 ? 1.f / range : 0.f;
}
__device__ __forceinline__ float div_range(const ActC& e, float x) {   // x / (max_action - min_action)
    if (e.inv_range != 0.f) return x * e.inv_range;
    return x / (e.max_action - e.min_action);
}
__device__ __forceinline__ void stage_act_const(const GopsEnv& e, float* s_ac, int tid) {
    if (tid < GOPS_MAX_ACT) {
        float* c = s_ac + tid * 8;
        c[0] = (e.policy_high[tid] - e.policy_low[tid]) / 2.f;
        c[1] = (e.policy_high[tid] + e.policy_low[tid]) / 2.f;
        c[2] = e.min_action[tid]; c[3] = e.max_action[tid];
        c[4] = e.act_low[tid]; c[5] = e.act_high[tid];
        c[6] = c[7] = 0.f;
    }
}
__device__ __forceinline__ ActC act_const(const float* s_ac, int a) {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(s_ac + a * 8), c1 = *reinterpret_cast<const f32x4*>(s_ac + a * 8 + 4);
    return ActC{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], exact_inverse_or_zero(c0[3] - c0[2])};
}
__device__ __forceinline__ float wrap_action(const ActC& e, float abar) {   // same arithmetic as the GopsEnv form below
    const float a1 = clampf(abar, e.min_action, e.max_action);
    const float a2 = e.act_low + (e.act_high - e.act_low) * div_range(e, a1 - e.min_action);
    const float a3 = clampf(a2, e.act_low, e.act_high);
    return clampf(a3, e.act_low, e.act_high);
}

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

// the same adjoint on per-action constants held in registers (ActC: the sweep of the plane-split kernels pins them to SGPRs)
__device__ __forceinline__ float wrap_action_bwd(const ActC& e, float abar, float g) {
    const float a1 = clampf(abar, e.min_action, e.max_action);
    const float a2 = e.act_low + (e.act_high - e.act_low) * div_range(e, a1 - e.min_action);   // (the forward's value: the clamp decision below)
    if (!(a2 >= e.act_low && a2 <= e.act_high)) g = 0.f;   // both clamps see the same range
    g = div_range(e, g * (e.act_high - e.act_low));
    if (!(abar >= e.min_action && abar <= e.max_action)) g = 0.f;
    return g;
}

// ScaleObservationModel (scale_observation.py:107-119): what the model steps / what the caller sees
__device__ __forceinline__ float obs_unscale(const GopsEnv& e, int i, float o) { return e.scale_obs ? o / e.obs_scale[i] - e.obs_shift[i] : o; }
__device__ __forceinline__ float obs_rescale(const GopsEnv& e, int i, float x) { return e.scale_obs ? (x + e.obs_shift[i]) * e.obs_scale[i] : x; }

// ((x + pi) mod 2pi) - pi with Python/torch remainder semantics (gops/utils/math_utils.py:8-11)
__device__ __forceinline__ float angle_normalize(float x) {
    const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
    const float y = x + pi;
    float r;
    // fmodf(y, 2pi) is exact; so are these shortcuts on their ranges (Sterbenz), which cover every
    // angle the env models produce in practice - the libm loop is only the fallback.
    if (y >= 0.f && y < two_pi) r = y;
    else if (y >= two_pi && y < 2.f * two_pi) r = y - two_pi;
    else if (y < 0.f && y > -two_pi) r = y;
    else r = fmodf(y, two_pi);
    if (r < 0.f) r += two_pi;
    return r - pi;
}

// Sum over each aligned group of 16 lanes, result in every lane: four DPP adds (quad_perm x2,
// row_half_mirror, row_mirror) - no LDS crossbar round trips like __shfl_xor.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror
    return v;
}

// ---- exact bf16 planes of fp32 values --------------------------------------------------------------------
// (a, b) -> three packed bf16 pairs (low half from a, high half from b); a == the sum of its three planes EXACTLY, and
// so is b.  Each plane is the TRUNCATION of the running residual to its top 16 bits (8 significant bits): the residual
// after one plane has at most 16 significant bits left, after two at most 8, so the third truncation is exact - same
// guarantee as rounding to nearest, but on full-rate integer / add instructions only (v_perm_b32 packs the two high
// halves; v_cvt_pk_bf16_f32 is a quarter-rate instruction).
__device__ __forceinline__ void split3(float a, float b, unsigned (&pl)[3]) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    pl[0] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    pl[1] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
    pl[2] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
// (a, b) -> packed half pair, round to nearest (v_cvt_pk_f16_f32), saturating at +-65504 instead of producing inf
__device__ __forceinline__ unsigned pk_half(float a, float b) {
    const f16x2 h = {(_Float16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
    return __builtin_bit_cast(unsigned, h);
}

// (a, b) * s -> packed half pairs hi, lo with x = hi + lo / 2^11 to 2^-22 |x| (x = a s): hi = f16(x) (round to nearest),
// lo = f16((x - hi) 2^11) - x - hi is exact in fp32, |lo| <= |x|: lo never overflows before hi does
// `ovf`: running packed maximum of |hi| as 16-bit patterns (v_pk_max_u16): any half >= 0x7c00 (inf / nan) means a value left the
// half range - split_overflowed().  Non-finite values do NOT reliably reach the outputs on their own (relu's fmaxf drops a nan, tanh
// squashes an inf), so the kernels test this flag at the end of a tile and poison the tile's results (rollout_fwd.hip / _bwd.hip).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(float a, float b, float s, unsigned& hi, unsigned& lo, unsigned& ovf) {
    const float xa = a * s, xb = b * s;      // (no clamp: beyond the half range hi = inf, lo = nan - see SPLIT_FWD_SA)
    const f16x2 h = {(_Float16)xa, (_Float16)xb};
    hi = __builtin_bit_cast(unsigned, h);
    ovf = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, ovf), __builtin_bit_cast(u16x2, hi & 0x7fff7fffu)));
    const f16x2 l = {(_Float16)((xa - (float)h[0]) * 2048.f), (_Float16)((xb - (float)h[1]) * 2048.f)};
    lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ bool split_overflowed(unsigned ovf) { return (ovf & 0xffffu) >= 0x7c00u || (ovf >> 16) >= 0x7c00u; }
// workgroup-wide OR of the per-thread overflow flags through four floats of LDS scratch the caller has free at that point
// (no static LDS: the plane-split kernels size their dynamic LDS to the last KiB of the CU's 160); two barriers
__device__ __forceinline__ bool split_overflow_any(unsigned ovf, float* scratch4, int tid) {
    const bool wave_any = __builtin_amdgcn_ballot_w64(split_overflowed(ovf)) != 0ull;
    if ((tid & 63) == 0) scratch4[tid >> 6] = wave_any ? 1.f : 0.f;
    __syncthreads();
    const bool any = (scratch4[0] != 0.f) || (scratch4[1] != 0.f) || (scratch4[2] != 0.f) || (scratch4[3] != 0.f);
    __syncthreads();
    return any;
}

// Plane image of a hidden tile, written from the MFMA result layout: lane (n = lane & 15, g = lane >> 4) of wave w holds
// v[q][r] = element (row 4 g + r, feature 64 w + 16 q + n) of its four n-tiles q.  The four features of a row go out as
// ONE 8-byte store per plane (16 consecutive lanes = 128 contiguous bytes), i.e. position 64 w + 4 n + q of the row holds
// feature 64 w + 16 q + n: split_perm().  The weights of the consuming GEMM are packed in the same order.
// `s16`: F16X2 - the scale of BOTH planes; bf16x3 + f16 form - the scale of the half plane.
__device__ __forceinline__ void plane_store(char* planes, int rowb, int wave, int lane, const f32x4 (&v)[4], float s16, unsigned& ovf) {
    const int pstride = TB * rowb;
    char* base = planes + ((lane >> 4) << 2) * rowb + 128 * wave + 8 * (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#if GOPS_SPLIT_F16X2
        u32x2 hi, lo;
        unsigned h0, l0, h1, l1;
        split2h(v[0][r], v[1][r], s16, h0, l0, ovf);
        split2h(v[2][r], v[3][r], s16, h1, l1, ovf);
        hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
        *reinterpret_cast<u32x2*>(base + r * rowb) = hi;
        *reinterpret_cast<u32x2*>(base + r * rowb + pstride) = lo;
#else
        unsigned p01[3], p23[3];
        split3(v[0][r], v[1][r], p01);
        split3(v[2][r], v[3][r], p23);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const u32x2 w = {p01[pl], p23[pl]};
            *reinterpret_cast<u32x2*>(base + r * rowb + pl * pstride) = w;
        }
        const u32x2 h = {pk_half(v[0][r] * s16, v[1][r] * s16), pk_half(v[2][r] * s16, v[3][r] * s16)};
        *reinterpret_cast<u32x2*>(base + r * rowb + 3 * pstride) = h;
#endif
    }
}

// Plane image of the fp32 input tile xs [TB][ldx] (columns < kp valid / zero) in natural column order, kp32 columns.
__device__ __forceinline__ void plane_convert_x(const float* xs, int ldx, int kp, int kp32, char* planes, int rowb, int tid, float s16, unsigned& ovf) {
    const int upr = kp32 >> 3, pstride = TB * rowb;
    for (int idx = tid; idx < TB * upr; idx += NTHREADS) {
        const int m = idx / upr, u = idx - m * upr, c = u << 3;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (c < kp) { a = *reinterpret_cast<const f32x4*>(xs + m * ldx + c); b = *reinterpret_cast<const f32x4*>(xs + m * ldx + c + 4); }
        char* dst = planes + m * rowb + 16 * u;
#if GOPS_SPLIT_F16X2
        u32x4 hi, lo;
        unsigned h, l;
        split2h(a[0], a[1], s16, h, l, ovf); hi[0] = h; lo[0] = l;
        split2h(a[2], a[3], s16, h, l, ovf); hi[1] = h; lo[1] = l;
        split2h(b[0], b[1], s16, h, l, ovf); hi[2] = h; lo[2] = l;
        split2h(b[2], b[3], s16, h, l, ovf); hi[3] = h; lo[3] = l;
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + pstride) = lo;
#else
        unsigned p0[3], p1[3], p2[3], p3[3];
        split3(a[0], a[1], p0); split3(a[2], a[3], p1); split3(b[0], b[1], p2); split3(b[2], b[3], p3);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const u32x4 w = {p0[pl], p1[pl], p2[pl], p3[pl]};
            *reinterpret_cast<u32x4*>(dst + pl * pstride) = w;
        }
        const u32x4 h = {pk_half(a[0] * s16, a[1] * s16), pk_half(a[2] * s16, a[3] * s16), pk_half(b[0] * s16, b[1] * s16), pk_half(b[2] * s16, b[3] * s16)};
        *reinterpret_cast<u32x4*>(dst + 3 * pstride) = h;
#endif
    }
}

// A value pinned to the accumulator half of the register file (v_accvgpr_write_b32 with an "a"-constrained result): the
// MFMA reads its B operand from there directly.  Left to itself hipcc keeps as many weight fragments as fit in VGPRs,
// parks the rest in AGPRs as SPILLS and copies each back (4 x v_accvgpr_read_b32 = 16 issue cycles) in front of the MFMA
// that needs it - as long as the 16-cycle MFMA itself.
template <class V>
__device__ __forceinline__ V pin_agpr(const V& x) {
    static_assert(sizeof(V) == 16, "128-bit fragments");
    const u32x4 u = __builtin_bit_cast(u32x4, x);
    u32x4 r;
    unsigned t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(t) : "v"(u[i]));
        r[i] = t;
    }
    return __builtin_bit_cast(V, r);
}

// Plane-split weights of one layer for this wave's NT n-tiles x KCH chunks of 32 inputs.  The bf16 plane is register-
// stationary; the half residual plane is register-stationary too (RLDS = false) or lives in LDS (RLDS = true: `rl` points
// at the workgroup's copy of the packed plane, fragment (n-tile nt, chunk c) of lane l at rl[((nt * KCH + c) * 64 + l)],
// read with conflict-free ds_read_b128 a few MFMAs ahead of its use).  One 256 x 256 layer is 256 registers per lane with
// both planes resident; two such layers do not fit beside the rest of the kernel, so layer 0 keeps its residual plane in LDS.
template <int KCH, int NT, bool RLDS, int PIN = 0>   // PIN: planes pinned to AGPRs (pin_agpr): bit 0 the residual plane, bit 1 the bf16 plane
struct StatQ {
    bf16x8 w[KCH * NT];
    f16x8 r[RLDS ? 1 : KCH * NT];
    const f16x8* rl;   // RLDS: this lane's first fragment of this wave's first n-tile
    float inv[NT];     // 1 / s_r of the n-tiles
    // lds_r (RLDS): LDS region of nt_tot * KCH * 64 fragments, filled here by all NTHREADS threads (caller: barrier before use)
    //
    // Launch-time cost (round 5, ISA of the round-4 build): written as one loop over fragments, every PINNED fragment went
    // global_load -> s_waitcnt vmcnt(0) -> 4 x v_accvgpr_write through the same four VGPRs - 32 serialized L2 round trips for a
    // 256 x 256 layer -, and the LDS fill of the residual plane was a load / wait / ds_write loop of 16 more: ~20 of the forward
    // kernel's 22 - 27 us of per-launch fixed cost (tools: kernel time at H = 1, 2, 5 .. 30).  Now every load of a group is issued
    // before the first result is touched: fragments in groups of LOAD_GROUP (64 temporaries), the LDS fill in batches of 8.
    static constexpr int LOAD_GROUP = 16;
    __device__ __forceinline__ void load(const bf16x8* __restrict__ W1, const f16x8* __restrict__ R, const float* __restrict__ inv_r,
                                         int nt_tot, int tid, f16x8* lds_r = nullptr) {
        const int lane = tid & 63, nt0 = (tid >> 6) * NT;
        size_t at0[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bool ok = nt0 + j < nt_tot;
            const float iv = gptr(inv_r)[ok ? nt0 + j : 0];   // (unconditional load: a guarded one is load, s_waitcnt, per n-tile)
            inv[j] = ok ? iv : 0.f;
            at0[j] = (size_t)(ok ? nt0 + j : 0) * KCH * 64 + lane;
        }
        if constexpr (!RLDS) {   // residual plane: plain loads (hipcc targets the accumulator registers directly, all in flight)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int c = 0; c < KCH; ++c) {
                    if constexpr ((PIN & 1) != 0) r[c * NT + j] = pin_agpr(gptr(R)[at0[j] + (size_t)c * 64]);
                    else r[c * NT + j] = gptr(R)[at0[j] + (size_t)c * 64];
                }
        }
        if constexpr ((PIN & 2) != 0) {
#pragma unroll
            for (int g0 = 0; g0 < KCH * NT; g0 += LOAD_GROUP) {
                bf16x8 tmp[LOAD_GROUP];
#pragma unroll
                for (int i = 0; i < LOAD_GROUP; ++i) {
                    const int f = g0 + i, c = f / NT, j = f - c * NT;
                    if (f < KCH * NT) tmp[i] = gptr(W1)[at0[j] + (size_t)c * 64];
                }
#pragma unroll
                for (int i = 0; i < LOAD_GROUP; ++i)
                    if (g0 + i < KCH * NT) w[g0 + i] = pin_agpr(tmp[i]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int c = 0; c < KCH; ++c) w[c * NT + j] = gptr(W1)[at0[j] + (size_t)c * 64];
        }
        if constexpr (RLDS) {
            const int total = nt_tot * KCH * 64;
            for (int base = tid; base < total; base += 8 * NTHREADS) {
                f16x8 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = gptr(R)[min(base + u * NTHREADS, total - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (base + u * NTHREADS < total) lds_r[base + u * NTHREADS] = v[u];
            }
            rl = lds_r + (size_t)(nt0 < nt_tot ? nt0 : 0) * KCH * 64 + lane;
        }
    }
};

// Launch-time fills of LDS from global memory: U loads in flight per thread before the first store.  (Left as
// `for (i = tid; i < n; i += NTHREADS) lds[i] = glob[i]`, hipcc emits load, s_waitcnt vmcnt(0), ds_write per iteration: one L2
// round trip each - see StatQ::load.)  `load(i)` must be safe for every i < n; `store(i, v)` is only called for i < n.
template <int U, class L, class S>
__device__ __forceinline__ void batched_fill(int n, int tid, L&& load, S&& store) {
    for (int base = tid; base < n; base += U * NTHREADS) {
        decltype(load(0)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load(min(base + u * NTHREADS, n - 1));
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * NTHREADS < n) store(base + u * NTHREADS, v[u]);
    }
}

// acc (bf16 main term) / accr (f16 residual term, still scaled) of this wave's NT n-tiles over the plane image `planes`.
// The four A fragments of chunk c + 1 (and an LDS-resident residual plane's fragments of chunk c) are requested ahead of
// chunk c's 4 NT MFMAs - pinned there with a scheduling barrier, because left alone hipcc sinks every ds_read to just
// in front of its first use and each group of NT MFMAs then waits out a full LDS round trip.
template <int KCH, int NT, bool RLDS, int PIN>
__device__ __forceinline__ void gemm_split(const char* planes, int rowb, const StatQ<KCH, NT, RLDS, PIN>& W, int lane,
                                           f32x4 (&acc)[NT], f32x4 (&accr)[NT]) {
    const int pstride = TB * rowb;
    const char* arow = planes + (lane & 15) * rowb + (lane >> 4) * 16;
#if GOPS_SPLIT_F16X2
    // (the `w` plane of StatQ holds half values in this form: its bf16x8 type is the 16-byte container)
    f16x8 ah = *reinterpret_cast<const f16x8*>(arow), al = *reinterpret_cast<const f16x8*>(arow + pstride);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        f16x8 rr[NT];
        if constexpr (RLDS) {
#pragma unroll
            for (int j = 0; j < NT; ++j) rr[j] = W.rl[(j * KCH + c) * 64];
        }
        f16x8 nh = ah, nl = al;
        if (c + 1 < KCH) {
            nl = *reinterpret_cast<const f16x8*>(arow + 64 * (c + 1) + pstride);
            nh = *reinterpret_cast<const f16x8*>(arow + 64 * (c + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, W.w[c * NT + j]), acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, __builtin_bit_cast(f16x8, W.w[c * NT + j]), accr[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if constexpr (RLDS) accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, rr[j], accr[j], 0, 0, 0);
            else accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, W.r[c * NT + j], accr[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        ah = nh; al = nl;
    }
#else
    bf16x8 a1 = *reinterpret_cast<const bf16x8*>(arow), a2 = *reinterpret_cast<const bf16x8*>(arow + pstride);
    bf16x8 a3 = *reinterpret_cast<const bf16x8*>(arow + 2 * pstride);
    f16x8 af = *reinterpret_cast<const f16x8*>(arow + 3 * pstride);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        f16x8 rr[NT];
        if constexpr (RLDS) {
#pragma unroll
            for (int j = 0; j < NT; ++j) rr[j] = W.rl[(j * KCH + c) * 64];
        }
        bf16x8 n1 = a1, n2 = a2, n3 = a3;
        f16x8 nf = af;
        if (c + 1 < KCH) {
            n3 = *reinterpret_cast<const bf16x8*>(arow + 64 * (c + 1) + 2 * pstride);
            n2 = *reinterpret_cast<const bf16x8*>(arow + 64 * (c + 1) + pstride);
            n1 = *reinterpret_cast<const bf16x8*>(arow + 64 * (c + 1));
            nf = *reinterpret_cast<const f16x8*>(arow + 64 * (c + 1) + 3 * pstride);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, W.w[c * NT + j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, W.w[c * NT + j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, W.w[c * NT + j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if constexpr (RLDS) accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, rr[j], accr[j], 0, 0, 0);
            else accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, W.r[c * NT + j], accr[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a1 = n1; a2 = n2; a3 = n3; af = nf;
    }
#endif
}

// Plane-split weights of a layer that does NOT stay on the CU (layer 0 of policies with more than 128 inputs: its planes
// would need 2 x 56 .. 64 KiB beside layer 1's 256 registers): the fragments stream from L2 - every CU of an XCD reads the
// same <= 256 KiB, so they are L2 hits - two n-tiles at a time through a ring of PF chunks per wave (PF x 2 x 2 fragments =
// 48 registers).  prime() issues the first PF chunks of a pair of n-tiles; gemm_split consumes chunk c and refills its slot
// with chunk c + PF right behind the MFMAs that read it.  Same packed layout as StatQ::load reads.
#define GOPS_SPLIT_PF 3
template <int KCH>
struct StreamRing {
    static constexpr int PF = GOPS_SPLIT_PF < KCH ? GOPS_SPLIT_PF : KCH;
    bf16x8 w[PF][2];
    f16x8 r[PF][2];
};
template <int KCH, int NT>   // NT n-tiles per wave (even), visited as NT / 2 pairs
struct StreamQ {
    // (uniform base pointers + 32-bit per-lane fragment indices: the loads take the scalar-base addressing form; per-lane
    //  64-bit pointers made hipcc precompute one address pair per fragment and spill them)
    const GLOBAL_AS bf16x8* w1;
    const GLOBAL_AS f16x8* r;
    unsigned at[NT];                  // this lane's fragment of chunk 0 of its n-tile j
    float inv[NT];
    __device__ __forceinline__ void load(const bf16x8* __restrict__ W1, const f16x8* __restrict__ R, const float* __restrict__ inv_r,
                                         int nt_tot, int tid) {
        const int lane = tid & 63, nt0 = (tid >> 6) * NT;
        w1 = gptr(W1);
        r = gptr(R);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bool ok = nt0 + j < nt_tot;   // (a wave's surplus n-tiles compute tile 0 again: their results are not stored)
            const float iv = gptr(inv_r)[ok ? nt0 + j : 0];
            inv[j] = ok ? iv : 0.f;
            at[j] = (unsigned)((ok ? nt0 + j : 0) * KCH * 64 + lane);
        }
    }
    // The fragment addresses are invariant across the rollout's steps, so hipcc hoists all 2 NT KCH of them (64-bit pairs) out
    // of the step loop, spills them, and reloads each from scratch in front of its load - behind an s_waitcnt vmcnt(0) that
    // also drains the ring.  off() hands out the lane's fragment index through an empty asm the optimizer cannot see
    // through: the address arithmetic (one 32-bit add per fragment) stays next to the load.
    __device__ __forceinline__ int off(int j) const {
        int o = (int)at[j];
        asm volatile("" : "+v"(o));
        return o;
    }
    __device__ __forceinline__ bf16x8 frag_w(int o, int c) const { return *(w1 + o + c * 64); }
    __device__ __forceinline__ f16x8 frag_r(int o, int c) const { return *(r + o + c * 64); }
    __device__ __forceinline__ void prime(StreamRing<KCH>& ring, int pair) const {
        const int o[2] = {off(2 * pair), off(2 * pair + 1)};
#pragma unroll
        for (int d = 0; d < StreamRing<KCH>::PF; ++d)
#pragma unroll
            for (int j = 0; j < 2; ++j) { ring.w[d][j] = frag_w(o[j], d); ring.r[d][j] = frag_r(o[j], d); }
    }
};

// n-tiles 2 pair, 2 pair + 1 of the wave: acc / accr are those two tiles' accumulators; the ring must hold the pair's first
// PF chunks (prime)
template <int KCH, int NT>
__device__ __forceinline__ void gemm_split_pair(const char* planes, int rowb, const StreamQ<KCH, NT>& W, StreamRing<KCH>& ring, int pair,
                                                int lane, f32x4 (&acc)[2], f32x4 (&accr)[2]) {
    constexpr int PF = StreamRing<KCH>::PF;
    const int pstride = TB * rowb;
    const int o[2] = {W.off(2 * pair), W.off(2 * pair + 1)};
    const char* arow = planes + (lane & 15) * rowb + (lane >> 4) * 16;
#if GOPS_SPLIT_F16X2
    f16x8 ah = *reinterpret_cast<const f16x8*>(arow), al = *reinterpret_cast<const f16x8*>(arow + pstride);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int slot = c % PF;
        f16x8 nh = ah, nl = al;
        if (c + 1 < KCH) {
            nl = *reinterpret_cast<const f16x8*>(arow + 64 * (c + 1) + pstride);
            nh = *reinterpret_cast<const f16x8*>(arow + 64 * (c + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, ring.w[slot][j]), acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, __builtin_bit_cast(f16x8, ring.w[slot][j]), accr[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ring.r[slot][j], accr[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + PF < KCH) {   // refill the slot behind the MFMAs that read it: in flight during the next PF - 1 chunks
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ring.w[slot][j] = W.frag_w(o[j], c + PF);
                ring.r[slot][j] = W.frag_r(o[j], c + PF);
            }
        }
        ah = nh; al = nl;
    }
#else
    bf16x8 a1 = *reinterpret_cast<const bf16x8*>(arow), a2 = *reinterpret_cast<const bf16x8*>(arow + pstride);
    bf16x8 a3 = *reinterpret_cast<const bf16x8*>(arow + 2 * pstride);
    f16x8 af = *reinterpret_cast<const f16x8*>(arow + 3 * pstride);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const int slot = c % PF;
        bf16x8 n1 = a1, n2 = a2, n3 = a3;
        f16x8 nf = af;
        if (c + 1 < KCH) {
            n3 = *reinterpret_cast<const bf16x8*>(arow + 64 * (c + 1) + 2 * pstride);
            n2 = *reinterpret_cast<const bf16x8*>(arow + 64 * (c + 1) + pstride);
            n1 = *reinterpret_cast<const bf16x8*>(arow + 64 * (c + 1));
            nf = *reinterpret_cast<const f16x8*>(arow + 64 * (c + 1) + 3 * pstride);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, ring.w[slot][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, ring.w[slot][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, ring.w[slot][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) accr[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, ring.r[slot][j], accr[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c + PF < KCH) {   // refill the slot behind the MFMAs that read it: in flight during the next PF - 1 chunks
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ring.w[slot][j] = W.frag_w(o[j], c + PF);
                ring.r[slot][j] = W.frag_r(o[j], c + PF);
            }
        }
        a1 = n1; a2 = n2; a3 = n3; af = nf;
    }
#endif
}

// One layer of the streamed-split kernels: this wave's four n-tiles (of nt_tot; surplus tiles recompute tile 0) over KCH
// chunks of the plane image `planes`, two n-tiles at a time through the ring.
template <int KCH>
__device__ __forceinline__ void ss_layer_gemm(const char* planes, int rowb, const bf16x8* W1, const f16x8* R, const float* inv_r,
                                              int nt_tot, int tid, f32x4 (&acc)[4], f32x4 (&accr)[4], float (&inv)[4]) {
    StreamQ<KCH, 4> Q;
    StreamRing<KCH> ring;
    Q.load(W1, R, inv_r, nt_tot, tid);
#pragma unroll
    for (int q = 0; q < 4; ++q) inv[q] = Q.inv[q];
    // narrow operands (g_x of a 4- or 6-column observation: ONE n-tile): a wave without a valid tile in a pair skips it
    // (wave-uniform), instead of recomputing tile 0 - acc / accr stay zero there and nothing of them is stored
    const int mine = nt_tot - 4 * (tid >> 6);
    if (mine <= 0) return;
    Q.prime(ring, 0);
    const int lane = tid & 63;
    f32x4 pa[2] = {}, pr[2] = {};
    gemm_split_pair(planes, rowb, Q, ring, 0, lane, pa, pr);
    acc[0] = pa[0]; acc[1] = pa[1]; accr[0] = pr[0]; accr[1] = pr[1];
    if (mine <= 2) return;
    Q.prime(ring, 1);
    f32x4 pb[2] = {}, ps[2] = {};
    gemm_split_pair(planes, rowb, Q, ring, 1, lane, pb, ps);
    acc[2] = pb[0]; acc[3] = pb[1]; accr[2] = ps[0]; accr[3] = ps[1];
}

__device__ __forceinline__ float row16_max(float v) {   // max over each aligned group of 16 lanes, result in every lane
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
// Backward sweep: power of two s that brings the tile's largest |delta_y| = mx into [8, 16) (half-precision plane of the
// deltas: whatever the magnitude of the gradients, the correction term sees mid-range halfs); out[0] = s, out[1] = 1 / s.
__device__ __forceinline__ void split_delta_scale(float mx, float* out) {
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);   // biased exponent; mx >= 0
    e = e < 4 ? 4 : (e > 250 ? 250 : e);                  // (zero / denormal / inf / nan tiles: any finite power of two will do)
    out[0] = __uint_as_float((unsigned)(257 - e) << 23);   // 2^(130 - e): mx * s in [8, 16)
    out[1] = __uint_as_float((unsigned)(e - 3) << 23);     // 2^(e - 130)
}

// ---- MFMA tile GEMM: acc[j] (16 x 16 tile nt0+j) += A[16 x 16*kchunks] * Wp ------------------
// A lives in LDS row-major with leading dimension lda (floats, lda % 4 == 0); lane l supplies
// A[m = l&15][16c + 4*(l>>4) + i] to the i-th v_mfma_f32_16x16x4_f32 of chunk c, and the packed
// operand holds the matching B values so that one dwordx4 load per lane feeds four MFMAs.
// Streamed B operand: PF chunks of fragments are kept in flight ahead of the MFMAs (L2 latency is
// 500+ cycles under load).  prime() may be called well before run() to hide the first fetch.
template <int NT>
struct StreamB {
    static constexpr int PF = (NT >= 4) ? 1 : 4;
    f32x4 ring[PF][NT];
    const GLOBAL_AS f32x4* wbase;
    __device__ __forceinline__ void prime(const f32x4* Wp, int kchunks, int nt0, int lane, int c_begin) {
        wbase = gptr(Wp) + (size_t)nt0 * kchunks * 64 + lane;
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int cd = (c_begin + d < kchunks) ? c_begin + d : kchunks - 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) ring[d][j] = wbase[((size_t)j * kchunks + cd) * 64];
        }
    }
    __device__ __forceinline__ void run(const float* A, int lda, int kchunks, int lane, f32x4 (&acc)[NT], int c_begin) {
        const float* arow = A + (lane & 15) * lda + 4 * (lane >> 4);
        for (int c0 = c_begin; c0 < kchunks; c0 += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int c = c0 + d;
                if (c < kchunks) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * c);
                    f32x4 bcur[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bcur[j] = ring[d][j];
#ifdef GOPS_STREAMB_EXACT_REFILL
                    // refill only while there is a chunk to fetch: a clamped (redundant) load stays pending on its
                    // registers and makes the next GEMM that re-uses them drain vmcnt(0) in front of its first MFMA
                    if (c + PF < kchunks) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) ring[d][j] = wbase[((size_t)j * kchunks + c + PF) * 64];
                    }
#else
                    const int cn = (c + PF < kchunks) ? c + PF : kchunks - 1;
#pragma unroll
                    for (int j = 0; j < NT; ++j) ring[d][j] = wbase[((size_t)j * kchunks + cn) * 64];
#endif
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bcur[j][i], acc[j], 0, 0, 0);
                    }
                }
            }
        }
    }
};

template <int NT>
__device__ __forceinline__ void mfma_gemm(const float* __restrict__ A, int lda, int kchunks,
                                          const f32x4* __restrict__ Wp, int nt0, int lane,
                                          f32x4 (&acc)[NT]) {
    StreamB<NT> sb;
    sb.prime(Wp, kchunks, nt0, lane, 0);
    sb.run(A, lda, kchunks, lane, acc, 0);
}

// One dense layer on the tile: out tiles are dealt to the 4 waves in contiguous groups, each wave
// walks its group 4 / 2 / 1 MFMA n-tiles at a time and hands the finished 16x16 accumulators
// (tile nt0+q: rows 4*(lane>>4)+r, column 16*(nt0+q) + (lane&15)) to `epi.operator()<COUNT>(acc, nt0)`.
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
            epi.template operator()<4>(acc, nt);
            nt += 4;
        } else if (left >= 2) {
            f32x4 acc[4] = {};
            mfma_gemm<2>(A, lda, kch, Wp, nt, lane, reinterpret_cast<f32x4(&)[2]>(acc));
            epi.template operator()<2>(acc, nt);
            nt += 2;
        } else {
            f32x4 acc[4] = {};
            mfma_gemm<1>(A, lda, kch, Wp, nt, lane, reinterpret_cast<f32x4(&)[1]>(acc));
            epi.template operator()<1>(acc, nt);
            nt += 1;
        }
    }
}

// ---- register-stationary weights ---------------------------------------------------------------
// With one workgroup per CU (B = 4096 -> 256 tiles) each wave owns a whole SIMD's 512-entry
// register file, 512 KiB per CU: more than LDS (160 KiB) and enough to keep a wave's slice of a
// 256-wide layer (KCH x NT dwordx4 fragments = 4*KCH*NT registers) resident for all H steps, so
// the weights are read from HBM/L2 once per launch instead of once per timestep.  hipcc places the
// fragments in AGPRs and feeds them to v_mfma directly (gfx950 unified VGPR/AGPR file).
template <int KCH, int NT>
struct StatW {
    f32x4 w[KCH * NT];
    __device__ __forceinline__ void load(const f32x4* __restrict__ Wp, int nt_tot, int tid, int kch_total = KCH) {
        const int lane = tid & 63, nt0 = (tid >> 6) * NT;
#pragma unroll
        for (int c = 0; c < KCH; ++c)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                w[c * NT + j] = (nt0 + j < nt_tot) ? gptr(Wp)[((size_t)(nt0 + j) * kch_total + c) * 64 + lane] : z;
            }
    }
};
struct NoW {};   // placeholder for a streamed layer
// Narrow nets (every hidden layer of the policy <= 64 wide or so: the shapes of the reference's example scripts) on the streamed fp32
// kernels: at one n-tile per wave a layer is a handful of MFMAs behind the L2 round trip of its weight fragments (prime -> first
// MFMA, once per layer per step).  The packed fragments of all hidden layers (<= NARROW_MAX_FLOATS) are copied to LDS once per
// launch instead; fragment (nt, c) of a layer is the 1 KiB at Wl + (nt * kch + c) * 64 (one 16-byte read per lane, conflict-free).
// Same products in the same order as the streamed path: bit-identical results (tests/test_narrow_gpu.py).  cfg1 (idpendulum,
// 64-64 net) per step: forward GEMM phases 4.1 k -> 3.0 k cycles, sweep 3.1 k + 2.2 k -> 2.1 k + 1.4 k; update 175 -> 164 us.
// (Reading a layer's fragments in one batch instead of chunk by chunk was measured: no gain.)
#define NARROW_MAX_FLOATS 8192   // 32 KiB: (16 + 64) x 64 is 5120
template <class Epi>
__device__ __forceinline__ void gemm_layer_lds(const float* A, int lda, int kch, int nt_tot, const f32x4* Wl, int tid, Epi&& epi) {
    const int lane = tid & 63, wave = tid >> 6;
    const int per = (nt_tot + 3) >> 2;
    const int nt_end = min(nt_tot, (wave + 1) * per);
    const float* arow = A + (lane & 15) * lda + 4 * (lane >> 4);
    for (int nt = wave * per; nt < nt_end; ++nt) {
        const f32x4* wl = Wl + (size_t)nt * kch * 64 + lane;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        f32x4 a = *reinterpret_cast<const f32x4*>(arow), b = wl[0];
        for (int c = 0; c < kch; ++c) {
            const int cn = (c + 1 < kch) ? c + 1 : c;
            const f32x4 an = *reinterpret_cast<const f32x4*>(arow + 16 * cn), bn = wl[cn * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
            a = an; b = bn;
        }
        f32x4 out[4] = {acc, {}, {}, {}};
        epi.template operator()<1>(out, nt);
    }
}
// copies `n4` 16-byte vectors of packed weights to LDS (all threads; the caller's next barrier publishes them)
__device__ __forceinline__ void narrow_fill(f32x4* dst, const f32x4* src, int n4, int tid) {
    for (int idx = tid; idx < n4; idx += NTHREADS) dst[idx] = gptr(src)[idx];
}
// every load issued so far complete (a real S_WAITCNT vmcnt(0) that the waitcnt pass accounts for)
__device__ __forceinline__ void settle_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// K-chunks 0..KCH-1 come from the stationary fragments, chunks KCH..kch_total-1 (if any) are
// streamed from the packed weights like mfma_gemm does.
template <int KCH, int NT, class Epi>
__device__ __forceinline__ void gemm_layer_stat(const float* A, int lda, const StatW<KCH, NT>& W,
                                                int nt_tot, int tid, Epi&& epi, int kch_total = KCH,
                                                const f32x4* Wp = nullptr) {
    const int lane = tid & 63, nt0 = (tid >> 6) * NT;
    if (nt0 >= nt_tot) return;
    const float* arow = A + (lane & 15) * lda + 4 * (lane >> 4);
    f32x4 acc[NT] = {};
    StreamB<NT> sb;
    if (kch_total > KCH) sb.prime(Wp, kch_total, nt0, lane, KCH);   // fetched behind the stationary MFMAs
    f32x4 a_cur = *reinterpret_cast<const f32x4*>(arow);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        // the next chunk's A fragment is fetched from LDS while this chunk's MFMAs run
        f32x4 a_nxt = a_cur;
        if (c + 1 < KCH) a_nxt = *reinterpret_cast<const f32x4*>(arow + 16 * (c + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], W.w[c * NT + j][i], acc[j], 0, 0, 0);
        a_cur = a_nxt;
    }
    if (kch_total > KCH) sb.run(A, lda, kch_total, lane, acc, KCH);
    f32x4 out[4] = {};
#pragma unroll
    for (int q = 0; q < NT; ++q) out[q] = acc[q];
    epi.template operator()<NT>(out, nt0);   // stationary layers: nt_tot is a multiple of NT (or < 4 with NT = 1)
}

// L2 warm-up of the stash tiles the NEXT backward step will read.  Plain loads whose values are
// consumed one step later (XOR into a sink), so the compiler's s_waitcnt for them lands after a whole
// step of work: the lines travel HBM -> L2 in the background and the real reads hit L2.  (Inline-asm
// loads are not an option: hipcc drains vmcnt(0) - i.e. every pending stash store - before each asm.)
#define TOUCH_SLOTS 4
// One warm-up line per (thread, slot): line g of the concatenation [H_1 .. H_L (Z_1 .. Z_L if GELU),
// env rows, first line of each X row] of the tile at stash row `prow`.  Fully unrolled over constant
// layer indices - nothing here may live in (scratch) memory.
__device__ __forceinline__ unsigned touch_fetch(const MlpDev& M, const StashDev& st, size_t prow, int g) {
    const int Lh = M.nl - 1;
    const bool gelu = M.act == GOPS_ACT_GELU;
#pragma unroll
    for (int j = 1; j < GOPS_MAX_LAYERS; ++j) {
        if (j <= Lh) {
            const int n = M.dims[j], nl = TB * n / 32;
            if (g >= 0 && g < nl) return *gptr(reinterpret_cast<const unsigned*>(st.h[j] + prow * n) + g * 32);
            g -= nl;
            if (gelu) {
                if (g >= 0 && g < nl) return *gptr(reinterpret_cast<const unsigned*>(st.z[j] + prow * n) + g * 32);
                g -= nl;
            }
        }
    }
    if (g >= 0 && g < TB * ENV_STASH / 32) return *gptr(reinterpret_cast<const unsigned*>(st.env + prow * ENV_STASH) + g * 32);
    g -= TB * ENV_STASH / 32;
    if (g >= 0 && g < 4) return *gptr(reinterpret_cast<const unsigned*>(st.x + prow * M.kp[0]) + g * 32);   // FM: columns 0..7
    return 0u;
}

// Asynchronous 16-byte-per-lane copy global -> LDS (global_load_lds_dwordx4): lane i's 16 bytes land at
// LDS address `lds_base` + 16 i (wave-uniform base in M0), no VGPR holds the data, completion is
// counted by vmcnt.  Issued as inline asm on purpose: the builtin makes hipcc wait vmcnt(0) before the
// next LDS read that MAY alias the destination - i.e. within a few dozen instructions - which turns the
// prefetch into a blocking load.  The caller owns the ordering: `s_waitcnt vmcnt(0)` + barrier before
// anyone reads the destination.  (An asm-issued VMEM op only makes the compiler's own counted waits
// more conservative, never unsafe: vmcnt retires in order.)
// NT: non-temporal policy - for streams that this launch reads once (the sweep's stash rows): MI355X_MICROARCH.md measures such
// fills landing ~18 % earlier, and the lines do not displace what the CU re-reads from L2.
template <bool NT = false>
__device__ __forceinline__ void async_copy16_to_lds(const float* gsrc, const float* lds_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds_base);
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gptr(gsrc)), "s"(m0v)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gptr(gsrc)), "s"(m0v)
                     : "memory");
}

// Copy a [TB][ncols] row-major LDS tile (leading dim ld, ld % 16 == 4) to the FM stash tile at `g`
// (g[n * 16 + m]): thread = (feature n, row group), four conflict-free ds_read_b32 down a column and one
// 16-byte store; consecutive threads write consecutive 16-byte units.
__device__ __forceinline__ void stash_tile_fm(const float* lds, int ld, int ncols, float* g, int tid) {
    for (int idx = tid; idx < ncols * 4; idx += NTHREADS) {
        const int n = idx >> 2, m0 = (idx & 3) << 2;
        const f32x4 v = {lds[m0 * ld + n], lds[(m0 + 1) * ld + n], lds[(m0 + 2) * ld + n], lds[(m0 + 3) * ld + n]};
        __builtin_nontemporal_store(v, gptr(reinterpret_cast<f32x4*>(g) + idx));
    }
}

// Copy a [TB][ncols] LDS tile (leading dim ld) to global rows g[(row0+m)*ncols ...], coalesced.
__device__ __forceinline__ void stash_tile(const float* lds, int ld, int ncols, float* g, size_t row0,
                                           int nrows_valid, int tid) {
    const int vec_per_row = ncols >> 2;   // ncols % 4 == 0
    if ((vec_per_row & (vec_per_row - 1)) == 0) {   // power of two: shifts instead of a division
        const int sh = 31 - __builtin_clz(vec_per_row);
#pragma unroll 4
        for (int idx = tid; idx < TB * vec_per_row; idx += NTHREADS) {
            const int m = idx >> sh, c4 = idx & (vec_per_row - 1);
            if (m < nrows_valid) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(lds + m * ld + 4 * c4);
                __builtin_nontemporal_store(v, gptr(reinterpret_cast<f32x4*>(g + (row0 + m) * ncols + 4 * c4)));
            }
        }
        return;
    }
    for (int idx = tid; idx < TB * vec_per_row; idx += NTHREADS) {
        const int m = idx / vec_per_row, c4 = idx - m * vec_per_row;
        if (m < nrows_valid) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lds + m * ld + 4 * c4);
            __builtin_nontemporal_store(v, gptr(reinterpret_cast<f32x4*>(g + (row0 + m) * ncols + 4 * c4)));
        }
    }
}

// Floats of LDS one hidden-activation / delta tile takes: [TB][ldh] floats, or [TB][ldh + 4] halfs in the F16 kernels.
__host__ __device__ inline int hidden_tile_floats(int ldh, bool f16) { return f16 ? TB * (ldh + 4) / 2 : TB * ldh; }

// Launch with up to 160 KiB of dynamic LDS (the default cap is 64 KiB).  The attribute belongs to a
// kernel, and every template instantiation is its own kernel: the "already raised" flag is keyed on the
// kernel's address (a function-local static would be shared by all instantiations of one signature).
bool lds_attr_needed(const void* kernel);   // api.hip: true the first time it sees `kernel`
template <class K, class... Args>
inline void launch_with_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    if (lds_attr_needed(reinterpret_cast<const void*>(kernel)))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}
