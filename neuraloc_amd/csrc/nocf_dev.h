// nocf_dev.h -- device-side types and small helpers shared by the translation units of libnocf.so
// (nocf_kernels.hip: tile / lane / mono / f64 / adjoint kernels and the C ABI; nocf_duo.hip: the split-role
// weight-stationary kernel).  Everything here is header-only (__forceinline__ device code, plain structs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nocf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// All LDS lives in one dynamic array that every device function names directly (never through a
// stored pointer), so each access is a ds_* instruction, not a flat one.
extern __shared__ __attribute__((aligned(16))) float lds[];

#define ZQLD 16              // row stride of the z = A s rows (r <= 16)

struct DevProb {
    int kind, obstacle, nAgents, training, agentDim;
    double r, alphQ, alphW, mass, grav;
    const float* xtarget;
};

struct DevPhi {
    const float *K0, *b0, *K, *b, *w, *A, *cw;
    const float* cbp;                // device address of c.bias, or null (then the plan carries the host value)
};

__device__ __forceinline__ float sigma_act(float o) {      // src/Phi.py:8-9
    const float ao = fabsf(o);
    return ao + logf(1.f + expf(-2.f * ao));
}

// sigma(o) and tanh(o) from one exponential: e = exp(-2|o|) in (0,1];
//   sigma = |o| + log(1+e)           (the reference's own overflow-safe form, src/Phi.py:8-9)
//   tanh  = sign(o) (1-e)/(1+e)      (absolute error <= ~1e-7, like any fp32 rounding of an O(1) value)
// hardware v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp) instead of the ~60-instruction libm calls.
__device__ __forceinline__ void act_pair(float o, float& sig, float& th) {
    const float ao = fabsf(o);
    const float e = __builtin_amdgcn_exp2f(ao * -2.885390081777927f);       // exp(-2|o|) = 2^(-2 log2(e) |o|)
    const float p = 1.f + e;
    sig = ao + 0.6931471805599453f * __builtin_amdgcn_logf(p);             // v_log_f32 is log2
    th = copysignf((1.f - e) * __builtin_amdgcn_rcpf(p), o);
}

__device__ __forceinline__ float tanh_fast(float o) {
    const float e = __builtin_amdgcn_exp2f(fabsf(o) * -2.885390081777927f);
    return copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), o);
}

// ------------------------------------------------------------------------------------------
// lane reductions on the DPP path (VALU operand swizzles, a few cycles each) instead of __shfl_xor, which hipcc turns
// into ds_bpermute_b32: an LDS round trip (>100 cycles) per step of a dependent chain.
//   quad_perm [1,0,3,2] = lane^1, [2,3,0,1] = lane^2, row_half_mirror = 7-i within 8 lanes, row_mirror = 15-i within 16
// After k steps every lane of an aligned 2^k group holds the group's sum (fixed order: deterministic).
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_peer(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum2(float v) { return v + dpp_peer<0xB1>(v); }
__device__ __forceinline__ float sum4(float v) { v = sum2(v); return v + dpp_peer<0x4E>(v); }
__device__ __forceinline__ float sum8(float v) { v = sum4(v); return v + dpp_peer<0x141>(v); }
__device__ __forceinline__ float sum16(float v) { v = sum8(v); return v + dpp_peer<0x140>(v); }
// all 64 lanes (every lane must be active): the four row totals meet through scalar registers
__device__ __forceinline__ float sum64(float v) {
    v = sum16(v);
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// aligned groups of seg = 2, 4, ..., 64 lanes
__device__ __forceinline__ float sum_seg(float v, int seg) {
    if (seg >= 64) return sum64(v);
    if (seg >= 2) v += dpp_peer<0xB1>(v);
    if (seg >= 4) v += dpp_peer<0x4E>(v);
    if (seg >= 8) v += dpp_peer<0x141>(v);
    if (seg >= 16) v += dpp_peer<0x140>(v);
    if (seg >= 32) v += __shfl_xor(v, 16);            // (no 32-lane DPP pattern on gfx9: one crossbar step)
    return v;
}

#define TWO_PI_D 6.283185307179586

__device__ __forceinline__ float gauss2(float x0, float x1, float m0, float m1, float cov, float denom) {
    const float e0 = x0 - m0, e1 = x1 - m1;
    return expf(-0.5f * ((e0 * e0) / cov + (e1 * e1) / cov)) / denom;
}

__device__ __forceinline__ float obstacle_cross2d(const DevProb& pb, float x0, float x1) {
    if (pb.obstacle == NOCF_OBS_SOFTCORRIDOR) {
        const float cov = 0.2f;
        const float denom = (float)TWO_PI_D * sqrtf(cov * cov);
        return ((gauss2(x0, x1, -2.5f, 0.f, cov, denom) + gauss2(x0, x1, 2.5f, 0.f, cov, denom))
                + gauss2(x0, x1, -1.5f, 0.f, cov, denom)) + gauss2(x0, x1, 1.5f, 0.f, cov, denom);
    }
    if (pb.obstacle == NOCF_OBS_HARDCORRIDOR) {
        const float denom = (float)TWO_PI_D * 1.0f;
        const float n1 = sqrtf(x0 * x0 + (x1 - 4.f) * (x1 - 4.f));
        const float n2 = sqrtf(x0 * x0 + (x1 + 3.5f) * (x1 + 3.5f));
        if (pb.training) {
            const float thr = (float)(2.0 + pb.r);
            if ((n1 < thr) || (n2 < thr))
                return gauss2(x0, x1, 0.f, 4.f, 1.f, denom) + gauss2(x0, x1, 0.f, -3.5f, 1.f, denom);
            return 0.f;
        }
        return ((n1 < 2.0f) || (n2 < 2.0f)) ? 1.f : 0.f;     // eval: the mask itself is summed (a count)
    }
    return 0.f;
}

__device__ __forceinline__ float obstacle_swarm(const DevProb& pb, float x0, float x1, float x2) {
    if (pb.obstacle != NOCF_OBS_BLOCKS) return 0.f;
    if (pb.training) {
        const double r = pb.r;
        const bool in1 = (x0 < (float)(2.0 + r)) && (x0 > (float)(-2.0 - r)) && (x1 < (float)(0.5 + r)) &&
                         (x1 > (float)(-0.5 - r)) && (x2 < (float)(7.0 + r));
        const bool in2 = (x0 < (float)(4.0 + r)) && (x0 > (float)(2.0 - r)) && (x1 < (float)(1.0 + r)) &&
                         (x1 > (float)(-1.0 - r)) && (x2 < (float)(4.0 + r));
        if (!(in1 || in2)) return 0.f;
        const float c15 = (float)15.749609945722419;          // (2 pi)^1.5
        const float den1 = c15 * sqrtf(243.f), den2 = c15 * sqrtf(81.f);
        float e0 = x0, e1 = x1, e2 = x2 - 2.f;
        const float q1 = expf(-0.5f * (((e0 * e0) / 9.f + (e1 * e1) / 3.f) + (e2 * e2) / 9.f)) / den1;
        e0 = x0 - 2.5f;
        const float q2 = expf(-0.5f * (((e0 * e0) / 9.f + (e1 * e1) / 3.f) + (e2 * e2) / 3.f)) / den2;
        return (q1 + q2) + 999.f;
    }
    const bool in1 = (x0 < 2.0f) && (x0 > -2.0f) && (x1 < 0.5f) && (x1 > -0.5f) && (x2 < 7.0f);
    const bool in2 = (x0 < 4.0f) && (x0 > 2.0f) && (x1 < 1.0f) && (x1 > -1.0f) && (x2 < 4.0f);
    return (in1 || in2) ? 1.f : 0.f;
}

__device__ __forceinline__ bool want_W(const DevProb& pb) {
    return (pb.kind == NOCF_PROB_QUADCOPTER) ? (pb.alphW > 0.0) : (pb.alphW != 0.0);
}

// Several time segments in ONE launch (include/nocf.h, nocf_rollout_segments_f32; BASELINE config 5's shock sweep: the second segments of all
// shock times at once): rows [k * rows, (k + 1) * rows) of the batch are integrated over [t0[k], RollArgs::t1] with nt[k] steps.  n = 0: one
// segment, RollArgs::t0 / nt / h.  Passed by value (272 bytes of kernel arguments; only the one-CU kernel reads it).
#define NOCF_MAX_SEG 16
struct SegTab { int n, rows; double t0[NOCF_MAX_SEG]; int nt[NOCF_MAX_SEG]; int slot0[NOCF_MAX_SEG]; };      // slot0: first time slot of the segment in zFull / ctrlFull

struct RollArgs {
    const float* x; long n;
    double t0, t1, h; int nt, stepper;
    float a0;
    float* z_out; float* persample; float* zFull; float* ctrlFull; int cdim;
    unsigned long long* stamps;
    float* sAll;                     // training: stage inputs s=[x,t] of every RK evaluation, [nt*nstage][n][d+1]
    // training, optional: the activation record (include/nocf.h, nocf_rollout_record_act_f32).  Four sections of actRows x m floats
    // (u0 = sigma(o), tanh(o), tanh(q), a = w + hN K1' v of every RK evaluation and sample) and one of actRows x (d+1) (grad Phi);
    // actRows = nt * nstage * n.  Only the split-role kernel writes it.
    float* act; long actRows;
    // training tape for the split-role adjoint (nocf_duo_bwd.inc; include/nocf.h, nocf_rollout_tape_f32): the record above with the
    // TERMINAL evaluation as one more block (actRows = (nt * nstage + 1) * n; sAll then has that block too), plus u_1 = u_0 + hN sigma(q)
    // of the terminal evaluation ([n][m]) and four scalars per evaluation and sample ([actRows][4]: dPhi/dt - H, the x-only cost terms
    // q and w of that state, 0; terminal block: Phi - alph0 G, 0, 0, 0).  Null: no tape.
    float* tapeU1; float* tapeSc;
    SegTab seg;
};
