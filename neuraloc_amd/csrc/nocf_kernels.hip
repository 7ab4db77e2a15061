// nocf_kernels.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI of the OCflow rollout hot path.
//
// One persistent launch integrates a whole rollout: every workgroup owns T = 4*S samples
// from t0 to t1 (samples never interact, src/OCflow.py:80-86 is the only cross-sample op),
// and for each RK stage evaluates grad Phi (src/Phi.py:99-138), the problem physics
// (src/problem/*.py) and the RK update (src/OCflow.py:143-184) out of LDS, never touching
// HBM between the initial load of x and the final cost rows.
//
// Matrix work uses v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per
// instruction = a [4 samples] x [64 hidden columns] tile with K=1, exact fp32.  A lane owns
// one hidden column (B operand = one packed weight per k), the 4 accumulator registers are
// the 4 samples, the A operand is the activation [sample = lane&3][k] read from LDS.  That
// is the wave64-native shape for a 4..16-row batch tile: nothing is padded to 16/32 rows.
//
// Weights are re-laid out once per call (pack_kernel) into lane-linear float4 images so a
// wave's weight fetch is one fully coalesced 1 KiB global_load_dwordx4 per 4 k-steps; the
// images (<= 2.9 MB for swarm50) live in the XCD L2s for the whole rollout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>
#include <vector>
#include "nocf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define UN 8                 // k-quads (4 k-steps each) per software-pipelined chunk
#define MAX_SK 4             // max split-K factor of a GEMM phase
#define MAX_NTH 12

// ------------------------------------------------------------------------------------------
// plan: shapes, packed-image offsets, LDS carve.  Built on the host, passed by value.
// ------------------------------------------------------------------------------------------
struct DevPlan {
    int d, D1, m, r, ME, nTh;
    int MB, MBE, DB;                 // 64-column blocks: hidden, hidden+A rows, d+1
    int KQ1, KQ6, KQm;               // k-quads of the three GEMM shapes, padded to UN
    int SK1, SK6, SKm;               // split-K factors
    int LD, LDs, PLD, GLD;           // LDS row strides (floats)
    int T, nwaves;
    long oW0f, oW0b, oWf, oWb, strideW;   // float4 offsets of the images in the workspace
    long ob0, ob, ow, ocw;                // float offsets of the padded vectors
    float hN, cb;
    // LDS carve (float offsets)
    int lSB, lU0, lU1, lTH, lAV, lV, lPART, lG, lZ0, lZA, lDZ, lRED, lSC, lPHI, lTRIG;
    int ZLD;
    int ldsFloats;
};

struct DevProb {
    int kind, obstacle, nAgents, training, agentDim;
    double r, alphQ, alphW, mass, grav;
    const float* xtarget;
};

struct DevPhi {
    const float *K0, *b0, *K, *b, *w, *A, *cw;
};

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float k0ext(const DevPhi& P, int m, int r, int D1, int j, int k) {
    // rows 0..m-1 = K0, rows m..m+r-1 = A  (the low-rank quadratic rides along the opening layer)
    if (k >= D1) return 0.f;
    if (j < m) return P.K0[(long)j * D1 + k];
    if (j < m + r) return P.A[(long)(j - m) * D1 + k];
    return 0.f;
}

__global__ void pack_kernel(DevPlan pl, DevPhi P, float* __restrict__ ws) {
    float4* ws4 = reinterpret_cast<float4*>(ws);
    const long nW0f = (long)pl.MBE * pl.KQ1 * 64;
    const long nW0b = (long)pl.DB * pl.KQ6 * 64;
    const long nWl = (long)pl.MB * pl.KQm * 64;
    const long nLayers = pl.nTh - 1;
    const long total4 = nW0f + nW0b + 2 * nLayers * nWl;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += stride) {
        float v[4];
        long dst;
        if (idx < nW0f) {                               // GEMM1 image: B[k][j] = K0ext[j][k]
            long q = idx;
            int lane = q & 63; q >>= 6;
            int kq = q % pl.KQ1; int cb = q / pl.KQ1;
            for (int e = 0; e < 4; ++e) v[e] = k0ext(P, pl.m, pl.r, pl.D1, cb * 64 + lane, kq * 4 + e);
            dst = pl.oW0f + idx;
        } else if (idx < nW0f + nW0b) {                 // GEMM6 image: B[k][i] = K0ext[k][i]
            long q = idx - nW0f;
            int lane = q & 63; q >>= 6;
            int kq = q % pl.KQ6; int cb = q / pl.KQ6;
            for (int e = 0; e < 4; ++e) v[e] = k0ext(P, pl.m, pl.r, pl.D1, kq * 4 + e, cb * 64 + lane);
            dst = pl.oW0b + (idx - nW0f);
        } else {
            long q = idx - nW0f - nW0b;
            int layer = q / (2 * nWl);                  // 0 -> reference layer 1
            long w = q - (long)layer * 2 * nWl;
            int back = w >= nWl;
            if (back) w -= nWl;
            int lane = w & 63; long ww = w >> 6;
            int kq = ww % pl.KQm; int cb = ww / pl.KQm;
            const float* Kl = P.K + (long)layer * pl.m * pl.m;
            for (int e = 0; e < 4; ++e) {
                int a = cb * 64 + lane, k = kq * 4 + e;
                float val = 0.f;
                if (a < pl.m && k < pl.m) val = back ? Kl[(long)k * pl.m + a]   // B[j][k] = K[j][k]: (jq*4+e, col)
                                                     : Kl[(long)a * pl.m + k];  // B[k][j] = K[j][k]
                v[e] = val;
            }
            dst = (back ? pl.oWb : pl.oWf) + (long)layer * pl.strideW + w;
        }
        ws4[dst] = make_float4(v[0], v[1], v[2], v[3]);
    }
    // padded vectors
    const long nb0 = (long)pl.MBE * 64, nb = nLayers * pl.MB * 64, nw = (long)pl.MB * 64, ncw = (long)pl.DB * 64;
    const long totalv = nb0 + nb + nw + ncw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < totalv; idx += stride) {
        float val = 0.f;
        long dst;
        if (idx < nb0) { if (idx < pl.m) val = P.b0[idx]; dst = pl.ob0 + idx; }
        else if (idx < nb0 + nb) {
            long q = idx - nb0; int layer = q / (pl.MB * 64); int c = q % (pl.MB * 64);
            if (c < pl.m) val = P.b[(long)layer * pl.m + c];
            dst = pl.ob + q;
        } else if (idx < nb0 + nb + nw) { long q = idx - nb0 - nb; if (q < pl.m) val = P.w[q]; dst = pl.ow + q; }
        else { long q = idx - nb0 - nb - nw; if (q < pl.D1) val = P.cw[q]; dst = pl.ocw + q; }
        ws[dst] = val;
    }
}

// ------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    // D[blk][i][j] += A[blk][i] * B[blk][j];  lane = 4*blk + (i for A | j for B,D);  D reg = i
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float sigma_act(float o) {      // src/Phi.py:8-9
    const float ao = fabsf(o);
    return ao + logf(1.f + expf(-2.f * ao));
}

struct Ctx {
    float* lds;
    const float* ws;
    int tid, nthreads, wave, lane;
};

// part[ks][t][col] = sum_{k in split ks} act[t][k] * B[k][col]
template <int S>
__device__ __forceinline__ void gemm_phase(const Ctx& c, const DevPlan& pl, const float4* __restrict__ img,
                                           int nblk, int KQ, int SK, const float* __restrict__ act, int ld) {
    float* part = c.lds + pl.lPART;
    const int pstride = pl.T * pl.PLD;
    const int chunks = KQ / UN;
    const int units = nblk * SK;
    const int arow = c.lane & 3;
    for (int u = c.wave; u < units; u += pl.nwaves) {
        const int cb = u / SK, ks = u - cb * SK;
        const int c0 = (chunks * ks) / SK, c1 = (chunks * (ks + 1)) / SK;
        f32x4 acc[S][2];
#pragma unroll
        for (int s = 0; s < S; ++s) { acc[s][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[s][1] = acc[s][0]; }
        const float4* wp = img + ((long)cb * KQ + (long)c0 * UN) * 64 + c.lane;
        const float* ap = act + arow * ld + c0 * UN * 4;
        float4 wc[UN], wn[UN];
        if (c0 < c1) {
#pragma unroll
            for (int i = 0; i < UN; ++i) wc[i] = wp[i * 64];
        }
        for (int ch = c0; ch < c1; ++ch) {
            wp += UN * 64;
            if (ch + 1 < c1) {
#pragma unroll
                for (int i = 0; i < UN; ++i) wn[i] = wp[i * 64];
            }
#pragma unroll
            for (int i = 0; i < UN; ++i) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float4 a = *reinterpret_cast<const float4*>(ap + s * 4 * ld + i * 4);
                    acc[s][0] = mfma4(a.x, wc[i].x, acc[s][0]);
                    acc[s][1] = mfma4(a.y, wc[i].y, acc[s][1]);
                    acc[s][0] = mfma4(a.z, wc[i].z, acc[s][0]);
                    acc[s][1] = mfma4(a.w, wc[i].w, acc[s][1]);
                }
            }
            ap += UN * 4;
#pragma unroll
            for (int i = 0; i < UN; ++i) wc[i] = wn[i];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const f32x4 rsum = acc[s][0] + acc[s][1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                part[ks * pstride + (4 * s + q) * pl.PLD + cb * 64 + c.lane] = rsum[q];
        }
    }
}

__device__ __forceinline__ float part_sum(const float* part, int pstride, int SK, int off) {
    float v = part[off];
    for (int k = 1; k < SK; ++k) v += part[k * pstride + off];
    return v;
}

// Segmented block reduction of NV values per thread.  Threads are split into T groups of
// G = nthreads/T consecutive threads (group t serves sample t).  Result: out[t*NV + v].
// Contains two barriers; every thread of the workgroup must call it.
template <int NV>
__device__ __forceinline__ void group_reduce(const Ctx& c, const DevPlan& pl, float (&v)[NV], float* out) {
    const int G = c.nthreads / pl.T;
    const int seg = G < 64 ? G : 64;
    for (int off = seg >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += __shfl_xor(v[i], off);
    }
    float* red = c.lds + pl.lRED;
    const int subseg = c.tid / seg;
    if ((c.tid & (seg - 1)) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[subseg * NV + i] = v[i];
    }
    __syncthreads();
    const int R = G / seg;                 // sub-segments per sample (>=1)
    if (c.tid < pl.T * NV) {
        const int t = c.tid / NV, i = c.tid - t * NV;
        float s = red[(t * R) * NV + i];
        for (int q = 1; q < R; ++q) s += red[(t * R + q) * NV + i];
        out[t * NV + i] = s;
    }
    __syncthreads();
}


// ------------------------------------------------------------------------------------------
// grad Phi (and optionally Phi) for the T samples whose s=[x,t] rows sit in SB.
// src/Phi.py:99-138 restated per sample row; results: G[t][0..d]; if need_value PHI[t]=Phi(s).
// Every thread of the workgroup must call it (it contains barriers).
// ------------------------------------------------------------------------------------------
template <int S>
__device__ void phi_eval(const Ctx& c, const DevPlan& pl, bool need_value) {
    const float4* ws4 = reinterpret_cast<const float4*>(c.ws);
    float* L = c.lds;
    float* SB = L + pl.lSB;
    float* U[2] = {L + pl.lU0, L + pl.lU1};
    float* TH = L + pl.lTH;
    float* AV = L + pl.lAV;
    float* V = L + pl.lV;
    float* PART = L + pl.lPART;
    float* G = L + pl.lG;
    const int T = pl.T, LD = pl.LD, PLD = pl.PLD, pstride = T * PLD;
    const float hN = pl.hN;
    const float* b0 = c.ws + pl.ob0;
    const float* wv = c.ws + pl.ow;
    const float* cw = c.ws + pl.ocw;
    const int lastLayer = pl.nTh - 1;

    // ---- opening layer: o = s K0^T + b0 ; z = A s
    __syncthreads();
    gemm_phase<S>(c, pl, ws4 + pl.oW0f, pl.MBE, pl.KQ1, pl.SK1, SB, pl.LDs);
    __syncthreads();
    for (int t = 0; t < T; ++t)
        for (int col = c.tid; col < pl.ME; col += c.nthreads) {
            const float raw = part_sum(PART, pstride, pl.SK1, t * PLD + col);
            if (col < pl.m) {
                const float o = raw + b0[col];
                U[0][t * LD + col] = sigma_act(o);
                TH[t * LD + col] = tanhf(o);
            } else {
                V[t * LD + col] = raw;                 // z = A s, consumed by the closing GEMM
            }
        }
    int cur = 0;
    // ---- residual layers, forward
    for (int i = 1; i <= lastLayer; ++i) {
        __syncthreads();
        gemm_phase<S>(c, pl, ws4 + pl.oWf + (long)(i - 1) * pl.strideW, pl.MB, pl.KQm, pl.SKm, U[cur], LD);
        __syncthreads();
        const float* bi = c.ws + pl.ob + (long)(i - 1) * pl.MB * 64;
        float* THi = TH + (long)i * T * LD;
        for (int t = 0; t < T; ++t)
            for (int col = c.tid; col < pl.m; col += c.nthreads) {
                const float q = part_sum(PART, pstride, pl.SKm, t * PLD + col) + bi[col];
                const float th = tanhf(q);
                if (i < lastLayer) {
                    THi[t * LD + col] = th;
                    U[cur ^ 1][t * LD + col] = U[cur][t * LD + col] + hN * sigma_act(q);
                } else {
                    const float wj = wv[col];
                    V[t * LD + col] = th * wj;
                    AV[t * LD + col] = wj;
                    if (need_value) U[cur ^ 1][t * LD + col] = U[cur][t * LD + col] + hN * sigma_act(q);
                }
            }
        cur ^= 1;
    }
    // ---- Phi itself (final time only): w.u + 1/2 |A s|^2 + c.s + cb   (src/Phi.py:91-96)
    if (need_value) {
        __syncthreads();
        const int Gsz = c.nthreads / T;
        const int t = c.tid / Gsz, j0 = c.tid - t * Gsz;
        float acc[1] = {0.f};
        for (int col = j0; col < pl.m; col += Gsz) acc[0] += wv[col] * U[cur][t * LD + col];
        for (int q = pl.m + j0; q < pl.ME; q += Gsz) { const float z = V[t * LD + q]; acc[0] += 0.5f * z * z; }
        for (int i = j0; i < pl.D1; i += Gsz) acc[0] += cw[i] * SB[t * pl.LDs + i];
        group_reduce<1>(c, pl, acc, L + pl.lPHI);
        if (c.tid < T) L[pl.lPHI + c.tid] += pl.cb;
    }
    // ---- backward sweep: a <- a + hN K_i^T (tanh(.) . a)
    for (int i = lastLayer; i >= 1; --i) {
        __syncthreads();
        gemm_phase<S>(c, pl, ws4 + pl.oWb + (long)(i - 1) * pl.strideW, pl.MB, pl.KQm, pl.SKm, V, LD);
        __syncthreads();
        const float* THp = TH + (long)(i - 1) * T * LD;
        for (int t = 0; t < T; ++t)
            for (int col = c.tid; col < pl.m; col += c.nthreads) {
                const float a = AV[t * LD + col] + hN * part_sum(PART, pstride, pl.SKm, t * PLD + col);
                AV[t * LD + col] = a;
                V[t * LD + col] = THp[t * LD + col] * a;
            }
    }
    // ---- closing: g = K0^T (tanh(o) . a) + A^T (A s) + c
    __syncthreads();
    gemm_phase<S>(c, pl, ws4 + pl.oW0b, pl.DB, pl.KQ6, pl.SK6, V, LD);
    __syncthreads();
    for (int t = 0; t < T; ++t)
        for (int i = c.tid; i < pl.D1; i += c.nthreads)
            G[t * pl.GLD + i] = part_sum(PART, pstride, pl.SK6, t * PLD + i) + cw[i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// problem physics: (x = SB rows, p = G rows) -> DZ[t] = [-grad_p H | L | |dPhi/dt - H| | Q | W]
// Cross2D.py:69-162, SwarmTraj.py:68-164, Quadcopter.py:65-113, utils.py:70-86.
// LHQW[t*4..] receives (L,H,Q,W) as the reference's calcLHQW returns them.
// ------------------------------------------------------------------------------------------
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

// Every thread of the workgroup must call it.  want_ctrl: also leave the quadcopter thrust in SC.
__device__ void physics_eval(const Ctx& c, const DevPlan& pl, const DevProb& pb, float* LHQW) {
    float* Lm = c.lds;
    const float* X = Lm + pl.lSB;
    const float* P = Lm + pl.lG;
    float* DZ = Lm + pl.lDZ;
    float* SC = Lm + pl.lSC;
    float* TRIG = Lm + pl.lTRIG;
    const int T = pl.T, d = pl.d, N = pb.nAgents, ad = pb.agentDim;
    const int Gsz = c.nthreads / T;
    const int t = c.tid / Gsz, j0 = c.tid - t * Gsz;
    const float* x = X + t * pl.LDs;
    const float* p = P + t * pl.GLD;

    float v[3] = {0.f, 0.f, 0.f};                  // sum p^2, raw obstacle sum, raw interaction sum
    for (int i = j0; i < d; i += Gsz) v[0] += p[i] * p[i];
    if (pb.kind == NOCF_PROB_CROSS2D) {
        if (pb.obstacle != NOCF_OBS_NONE)
            for (int a = j0; a < N; a += Gsz) v[1] += obstacle_cross2d(pb, x[2 * a], x[2 * a + 1]);
    } else if (pb.kind == NOCF_PROB_SWARMTRAJ) {
        if (pb.obstacle != NOCF_OBS_NONE && pb.alphQ > 0.0)
            for (int a = j0; a < N; a += Gsz) v[1] += obstacle_swarm(pb, x[3 * a], x[3 * a + 1], x[3 * a + 2]);
    }
    const bool wantW = (pb.kind == NOCF_PROB_QUADCOPTER) ? (pb.alphW > 0.0) : (pb.alphW != 0.0);
    if (wantW && N >= 2) {
        const float den = (float)(2.0 * pb.r * pb.r);
        const int pd = (pb.kind == NOCF_PROB_CROSS2D) ? 2 : 3;      // position components per agent
        if (N == 2) {
            if (j0 == 0) {
                float s2 = 0.f;
                for (int k = 0; k < pd; ++k) { const float e = x[k] - x[ad + k]; s2 += e * e; }
                const float dist = sqrtf(s2);
                const double fac = (pb.kind != NOCF_PROB_QUADCOPTER && pb.training) ? 2.2 : 2.0;
                if (dist < (float)(fac * pb.r)) v[2] = expf(-(dist * dist) / den);
            }
        } else if (pb.kind != NOCF_PROB_QUADCOPTER) {
            const double fac = pb.training ? (pb.kind == NOCF_PROB_SWARMTRAJ ? 3.2 : 2.2) : 2.0;
            const float thr = (float)(fac * pb.r);
            for (int idx = j0; idx < N * N; idx += Gsz) {
                const int i = idx / N, j = idx - i * N;
                if (i < j) {
                    float s2 = 0.f;
                    for (int k = 0; k < pd; ++k) { const float e = x[ad * i + k] - x[ad * j + k]; s2 += e * e; }
                    const float dist = sqrtf(s2);
                    if (dist < thr) {
                        const float e = expf(-(dist * dist) / den);
                        if (e != 1.f) v[2] += e;             // the reference drops entries equal to 1.
                    }
                }
            }
        }
    }
    if (pb.kind == NOCF_PROB_QUADCOPTER && j0 < 3 * N) {       // sin/cos of (psi, theta, phi) per agent
        const int a = j0 / 3, q = j0 - 3 * a;
        float sn, cs;
        sincosf(x[12 * a + 3 + q], &sn, &cs);
        TRIG[(t * N + a) * 6 + q] = sn;
        TRIG[(t * N + a) * 6 + 3 + q] = cs;
    }
    group_reduce<3>(c, pl, v, SC);                              // SC[t*3 + {0,1,2}]

    if (pb.kind != NOCF_PROB_QUADCOPTER) {
        for (int tt = 0; tt < T; ++tt)
            for (int i = c.tid; i < d; i += c.nthreads) DZ[tt * pl.ZLD + i] = -P[tt * pl.GLD + i];
        if (c.tid < T) {
            const int s = c.tid;
            const float sp2 = SC[s * 3 + 0], Qraw = SC[s * 3 + 1], Wv = wantW ? SC[s * 3 + 2] : 0.f;
            float Lg, Qret;
            if (pb.kind == NOCF_PROB_CROSS2D) { Qret = (float)pb.alphQ * Qraw; Lg = 0.5f * sp2 + Qret; }
            else { Qret = Qraw; Lg = 0.5f * sp2 + (float)pb.alphQ * Qraw; }
            if (wantW) Lg = Lg + (float)pb.alphW * Wv;
            const float H = -Lg + sp2;
            float* dz = DZ + s * pl.ZLD;
            dz[d] = Lg;
            dz[d + 1] = fabsf(P[s * pl.GLD + d] - H);
            dz[d + 2] = Qret;
            dz[d + 3] = Wv;
            if (LHQW) { LHQW[s * 4 + 0] = Lg; LHQW[s * 4 + 1] = H; LHQW[s * 4 + 2] = Qret; LHQW[s * 4 + 3] = Wv; }
        }
    } else if (c.tid < T) {
        const int s = c.tid;
        const float* xs = X + s * pl.LDs;
        const float* ps = P + s * pl.GLD;
        float* dz = DZ + s * pl.ZLD;
        const float mass = (float)pb.mass, grav = (float)pb.grav;
        const float Qret = 0.f;                                  // Quadcopter.py:116-122: no obstacle implemented
        float Lg = (float)pb.alphQ * Qret;
        float Wv = 0.f;
        if (wantW) { Wv = SC[s * 3 + 2]; Lg = Lg + (float)pb.alphW * Wv; }
        float H = 0.f;
        for (int a = 0; a < N; ++a) {
            const float* xa = xs + 12 * a;
            const float* pa = ps + 12 * a;
            const float* tr = TRIG + (s * N + a) * 6;
            const float sps = tr[0], sth = tr[1], sph = tr[2], cps = tr[3], cth = tr[4], cph = tr[5];
            const float f7 = sps * sph + cps * sth * cph;
            const float f8 = -cps * sph + sps * sth * cph;
            const float f9 = cth * cph;
            const float fsum = f7 * pa[6] + f8 * pa[7] + f9 * pa[8];
            const float u = (float)(-1.0 / (2.0 * pb.mass)) * fsum;
            const float sq = pa[9] * pa[9] + pa[10] * pa[10] + pa[11] * pa[11];
            Lg = Lg + 2.f + u * u + 0.25f * sq;
            const float s1 = xa[6] * pa[0] + xa[7] * pa[1] + xa[8] * pa[2];
            const float s2 = xa[9] * pa[3] + xa[10] * pa[4] + xa[11] * pa[5];
            const float um = u / mass;
            H = H - Lg - s1 - s2 - um * fsum + grav * pa[8] + 0.5f * sq;
            float* da = dz + 12 * a;
            for (int k = 0; k < 6; ++k) da[k] = xa[6 + k];
            da[6] = um * f7; da[7] = um * f8; da[8] = um * f9 - grav;
            da[9] = -0.5f * pa[9]; da[10] = -0.5f * pa[10]; da[11] = -0.5f * pa[11];
            SC[T * 3 + s * N + a] = u;                           // thrust, for calcCtrls
        }
        dz[d] = Lg;
        dz[d + 1] = fabsf(ps[d] - H);
        dz[d + 2] = Qret;
        dz[d + 3] = Wv;
        if (LHQW) { LHQW[s * 4 + 0] = Lg; LHQW[s * 4 + 1] = H; LHQW[s * 4 + 2] = Qret; LHQW[s * 4 + 3] = Wv; }
    }
    __syncthreads();
}

// controls for the T samples from (x = SB, p = G) -> global rows; Cross2D.py:164-165,
// SwarmTraj.py:166-167, Quadcopter.py:165-174.  Needs physics_eval to have run on the same (x,p)
// for the quadcopter (thrust in SC).
__device__ void ctrl_write(const Ctx& c, const DevPlan& pl, const DevProb& pb, float* out, long row0, long n, int cdim) {
    const float* P = c.lds + pl.lG;
    const float* SC = c.lds + pl.lSC;
    for (int t = 0; t < pl.T; ++t) {
        if (row0 + t >= n) break;
        float* o = out + (row0 + t) * cdim;
        if (pb.kind != NOCF_PROB_QUADCOPTER) {
            for (int i = c.tid; i < pl.d; i += c.nthreads) o[i] = -P[t * pl.GLD + i];
        } else {
            for (int i = c.tid; i < cdim; i += c.nthreads) {
                const int a = i >> 2, q = i & 3;
                o[i] = (q == 0) ? SC[pl.T * 3 + t * pb.nAgents + a] : -0.5f * P[t * pl.GLD + 12 * a + 8 + q];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// the rollout kernel: src/OCflow.py:7-95 for T samples per workgroup
// ------------------------------------------------------------------------------------------
struct RollArgs {
    const float* x; long n;
    double t0, t1, h; int nt, stepper;
    float a0;
    float* z_out; float* persample; float* zFull; float* ctrlFull; int cdim;
};

template <int S>
__global__ void __launch_bounds__(512) rollout_kernel(DevPlan pl, DevProb pb, const float* __restrict__ ws, RollArgs ra) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Ctx c;
    c.lds = lds; c.ws = ws;
    c.tid = threadIdx.x; c.nthreads = blockDim.x; c.wave = threadIdx.x >> 6; c.lane = threadIdx.x & 63;
    const int T = pl.T, d = pl.d, ZLD = pl.ZLD;
    const long row0 = (long)blockIdx.x * T;
    float* SB = lds + pl.lSB;
    float* Z0 = lds + pl.lZ0;
    float* ZA = lds + pl.lZA;
    float* DZ = lds + pl.lDZ;

    for (int i = c.tid; i < pl.ldsFloats; i += c.nthreads) lds[i] = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        long row = row0 + t; if (row >= ra.n) row = ra.n - 1;          // tail rows replicate a valid sample
        for (int i = c.tid; i < d; i += c.nthreads) {
            const float v = ra.x[row * d + i];
            Z0[t * ZLD + i] = v;
            SB[t * pl.LDs + i] = v;
        }
        if (c.tid == 0) SB[t * pl.LDs + d] = (float)ra.t0;
    }
    if (ra.zFull) {
        for (int t = 0; t < T; ++t) {
            const long row = row0 + t;
            if (row < ra.n) {
                for (int i = c.tid; i < d + 4; i += c.nthreads)
                    ra.zFull[row * (d + 4) + i] = (i < d) ? ra.x[row * d + i] : 0.f;
                for (int i = c.tid; i < ra.cdim; i += c.nthreads) ra.ctrlFull[row * ra.cdim + i] = 0.f;
            }
        }
    }
    __syncthreads();

    const float c16 = (float)(1.0 / 6.0), c26 = (float)(2.0 / 6.0);
    double tk = ra.t0;
    const int nstage = (ra.stepper == NOCF_RK4) ? 4 : 1;
    const int nsub = nstage + (ra.zFull ? 1 : 0);
    // One call site for phi_eval/physics_eval: steps k < nt run the RK stages (plus, with
    // intermediates, the control evaluation); the extra pass k == nt is the terminal evaluation.
    for (int k = 0; k <= ra.nt; ++k) {
        const bool fin = (k == ra.nt);
        const double t1k = tk + ra.h;
        const double hsd = t1k - tk;                  // stepRK4 re-derives h = t1 - t0 (src/OCflow.py:170)
        const float hs = (float)hsd;
        if (fin) {
            if (c.tid < T) SB[c.tid * pl.LDs + d] = (float)ra.t1;      // src/OCflow.py:62
        }
        for (int st = 0; st < (fin ? 1 : nsub); ++st) {
            phi_eval<S>(c, pl, fin);
            if (fin) break;
            physics_eval(c, pl, pb, nullptr);
            if (st < nstage) {
                // ---- RK update (src/OCflow.py:143-184), elementwise over the d+4 components
                double tnext;
                if (nstage == 1) tnext = t1k;
                else tnext = (st < 2) ? (tk + hsd / 2) : (st == 2 ? (tk + hsd) : t1k);
                const bool last = (st == nstage - 1);
                // with intermediates the next evaluation is the control at (z_{k+1}, (tk+h)-h), src/OCflow.py:53
                if (last && ra.zFull) tnext = t1k - ra.h;
                for (int t = 0; t < T; ++t)
                    for (int i = c.tid; i < d + 4; i += c.nthreads) {
                        const float K = hs * DZ[t * ZLD + i];
                        const float z0 = Z0[t * ZLD + i];
                        float xs;
                        if (nstage == 1) { xs = z0 + K; Z0[t * ZLD + i] = xs; }
                        else if (st == 0) { ZA[t * ZLD + i] = z0 + c16 * K; xs = z0 + 0.5f * K; }
                        else if (st == 1) { ZA[t * ZLD + i] += c26 * K; xs = z0 + 0.5f * K; }
                        else if (st == 2) { ZA[t * ZLD + i] += c26 * K; xs = z0 + K; }
                        else { xs = ZA[t * ZLD + i] + c16 * K; Z0[t * ZLD + i] = xs; }
                        if (i < d) SB[t * pl.LDs + i] = xs;
                        else if (i == d) SB[t * pl.LDs + d] = (float)tnext;
                        if (last && ra.zFull && row0 + t < ra.n)
                            ra.zFull[((long)(k + 1) * ra.n + row0 + t) * (d + 4) + i] = xs;
                    }
            } else {
                ctrl_write(c, pl, pb, ra.ctrlFull + (long)(k + 1) * ra.n * ra.cdim, row0, ra.n, ra.cdim);
                if (c.tid < T) SB[c.tid * pl.LDs + d] = (float)t1k;
            }
            __syncthreads();
        }
        tk += ra.h;
    }

    // ---- terminal costs (src/OCflow.py:58-76)
    {
        const int Gsz = c.nthreads / T;
        const int t = c.tid / Gsz, j0 = c.tid - t * Gsz;
        const float* g = lds + pl.lG + t * pl.GLD;
        float v[2] = {0.f, 0.f};
        for (int i = j0; i < d; i += Gsz) {
            const float res = Z0[t * ZLD + i] - pb.xtarget[i];
            v[0] += res * res;
            v[1] += fabsf(g[i] - ra.a0 * res);
        }
        float* SC = lds + pl.lSC;
        group_reduce<2>(c, pl, v, SC);
        if (c.tid < T && row0 + c.tid < ra.n) {
            const int s = c.tid;
            const long row = row0 + s;
            const float cG = 0.5f * SC[s * 2 + 0];
            const float* z = Z0 + s * ZLD;
            if (ra.persample) {
                float* o = ra.persample + row * 7;
                o[0] = z[d]; o[1] = cG; o[2] = z[d + 1];
                o[3] = fabsf(lds[pl.lPHI + s] - ra.a0 * cG);
                o[4] = SC[s * 2 + 1];
                o[5] = z[d + 2]; o[6] = z[d + 3];
            }
        }
        if (ra.z_out)
            for (int tt = 0; tt < T; ++tt)
                if (row0 + tt < ra.n)
                    for (int i = c.tid; i < d + 4; i += c.nthreads) ra.z_out[(row0 + tt) * (d + 4) + i] = Z0[tt * ZLD + i];
    }
}

// ------------------------------------------------------------------------------------------
// stand-alone Phi.getGrad / Phi.forward and problem physics (same device code as the rollout)
// ------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(512) phi_kernel(DevPlan pl, const float* __restrict__ ws, const float* __restrict__ s,
                                                  long n, float* grad, float* value) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Ctx c;
    c.lds = lds; c.ws = ws;
    c.tid = threadIdx.x; c.nthreads = blockDim.x; c.wave = threadIdx.x >> 6; c.lane = threadIdx.x & 63;
    const int T = pl.T, D1 = pl.D1;
    const long row0 = (long)blockIdx.x * T;
    for (int i = c.tid; i < pl.ldsFloats; i += c.nthreads) lds[i] = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        long row = row0 + t; if (row >= n) row = n - 1;
        for (int i = c.tid; i < D1; i += c.nthreads) lds[pl.lSB + t * pl.LDs + i] = s[row * D1 + i];
    }
    phi_eval<S>(c, pl, value != nullptr);
    for (int t = 0; t < T; ++t) {
        if (row0 + t >= n) break;
        if (grad) for (int i = c.tid; i < D1; i += c.nthreads) grad[(row0 + t) * D1 + i] = lds[pl.lG + t * pl.GLD + i];
        if (value && c.tid == 0) value[row0 + t] = lds[pl.lPHI + t];
    }
}

__global__ void __launch_bounds__(512) prob_kernel(DevPlan pl, DevProb pb, const float* __restrict__ x,
                                                   const float* __restrict__ p, long n,
                                                   float* lhqw, float* gradpH, float* ctrls, int cdim) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Ctx c;
    c.lds = lds; c.ws = nullptr;
    c.tid = threadIdx.x; c.nthreads = blockDim.x; c.wave = threadIdx.x >> 6; c.lane = threadIdx.x & 63;
    const int T = pl.T, d = pl.d;
    const long row0 = (long)blockIdx.x * T;
    for (int i = c.tid; i < pl.ldsFloats; i += c.nthreads) lds[i] = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        long row = row0 + t; if (row >= n) row = n - 1;
        for (int i = c.tid; i < d; i += c.nthreads) {
            lds[pl.lSB + t * pl.LDs + i] = x[row * d + i];
            lds[pl.lG + t * pl.GLD + i] = p[row * d + i];
        }
    }
    __syncthreads();
    float* LH = lds + pl.lPART;                  // scratch for (L,H,Q,W)
    physics_eval(c, pl, pb, LH);
    for (int t = 0; t < T; ++t) {
        if (row0 + t >= n) break;
        if (lhqw && c.tid < 4) lhqw[(row0 + t) * 4 + c.tid] = LH[t * 4 + c.tid];
        if (gradpH) for (int i = c.tid; i < d; i += c.nthreads) gradpH[(row0 + t) * d + i] = -lds[pl.lDZ + t * pl.ZLD + i];
    }
    if (ctrls) ctrl_write(c, pl, pb, ctrls, row0, n, cdim);
}

// deterministic reduction of the per-sample table: 7 column sums (fp64 accumulation, fixed order) + n
__global__ void __launch_bounds__(256) cost_sum_kernel(const float* __restrict__ tab, long n, float* __restrict__ out) {
    __shared__ double sh[256 * 7];
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (long row = threadIdx.x; row < n; row += 256)
        for (int j = 0; j < 7; ++j) acc[j] += (double)tab[row * 7 + j];
    for (int j = 0; j < 7; ++j) sh[threadIdx.x * 7 + j] = acc[j];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int j = 0; j < 7; ++j) sh[threadIdx.x * 7 + j] += sh[(threadIdx.x + w) * 7 + j];
        __syncthreads();
    }
    if (threadIdx.x < 7) out[threadIdx.x] = (float)sh[threadIdx.x];
    if (threadIdx.x == 7) out[7] = (float)n;
}

__global__ void mfma_selftest_kernel(const float* __restrict__ a, const float* __restrict__ b, int K, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) acc = mfma4(a[(lane & 3) * K + k], b[k * 64 + lane], acc);
    for (int q = 0; q < 4; ++q) out[q * 64 + lane] = acc[q];
}

// ------------------------------------------------------------------------------------------
// host side: plan construction and the C ABI
// ------------------------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int rup(int a, int b) { return cdiv(a, b) * b; }

static int choose_sk(int nblk, int chunks, int nwaves) {
    int best = 1; long bestCost = -1;
    for (int sk = 1; sk <= MAX_SK && sk <= chunks; ++sk) {
        const long rounds = cdiv(nblk * sk, nwaves);
        const long cost = rounds * cdiv(chunks, sk) * 8 + sk;       // makespan in chunks, mild penalty per split
        if (bestCost < 0 || cost < bestCost) { bestCost = cost; best = sk; }
    }
    return best;
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// fills the shape / image part of the plan; returns 0 or an NOCF_E_* code
static int make_plan(int d, int m, int nTh, int r, int n_agents, DevPlan* out) {
    if (d < 1 || m < 1 || nTh < 2 || nTh > MAX_NTH || r < 1 || r > d + 1) return NOCF_E_SHAPE;
    DevPlan pl;
    memset(&pl, 0, sizeof(pl));
    pl.d = d; pl.D1 = d + 1; pl.m = m; pl.r = r; pl.ME = m + r; pl.nTh = nTh;
    pl.MB = cdiv(m, 64); pl.MBE = cdiv(pl.ME, 64); pl.DB = cdiv(pl.D1, 64);
    pl.KQ1 = rup(cdiv(pl.D1, 4), UN); pl.KQ6 = rup(cdiv(pl.ME, 4), UN); pl.KQm = rup(cdiv(m, 4), UN);
    // geometry: waves per workgroup and sample sub-tiles
    int nw = 1;
    while (nw < 8 && nw < pl.MBE) nw *= 2;
    int S = 1;
    nw = env_int("NOCF_NWAVES", nw);
    S = env_int("NOCF_SUBTILES", S);
    if (!(nw == 1 || nw == 2 || nw == 4 || nw == 8) || !(S == 1 || S == 2 || S == 4)) return NOCF_E_SHAPE;
    pl.nwaves = nw; pl.T = 4 * S;
    pl.SK1 = choose_sk(pl.MBE, pl.KQ1 / UN, nw);
    pl.SK6 = choose_sk(pl.DB, pl.KQ6 / UN, nw);
    pl.SKm = choose_sk(pl.MB, pl.KQm / UN, nw);
    const int skmax = std::max(pl.SK1, std::max(pl.SK6, pl.SKm));
    // LDS row strides: 64j+4 floats keeps the four sample rows of an A-operand read on distinct 16-B slots
    const int kmaxH = std::max(pl.KQ6, pl.KQm) * 4;
    pl.LD = rup(std::max(kmaxH, pl.MBE * 64), 64) + 4;
    pl.LDs = rup(pl.KQ1 * 4, 64) + 4;
    pl.PLD = std::max(pl.MBE, pl.DB) * 64;
    pl.GLD = pl.DB * 64;
    pl.ZLD = rup(d + 4, 4);
    // packed images
    long o4 = 0;
    pl.oW0f = o4; o4 += (long)pl.MBE * pl.KQ1 * 64;
    pl.oW0b = o4; o4 += (long)pl.DB * pl.KQ6 * 64;
    pl.strideW = (long)pl.MB * pl.KQm * 64;
    pl.oWf = o4; o4 += (long)(nTh - 1) * pl.strideW;
    pl.oWb = o4; o4 += (long)(nTh - 1) * pl.strideW;
    long of = o4 * 4;
    pl.ob0 = of; of += (long)pl.MBE * 64;
    pl.ob = of; of += (long)(nTh - 1) * pl.MB * 64;
    pl.ow = of; of += (long)pl.MB * 64;
    pl.ocw = of; of += (long)pl.DB * 64;
    // LDS carve
    const int T = pl.T;
    int l = 0;
    auto take = [&](int nfl) { int o = l; l += rup(nfl, 4); return o; };
    pl.lSB = take(T * pl.LDs);
    pl.lU0 = take(T * pl.LD); pl.lU1 = take(T * pl.LD);
    pl.lTH = take((nTh - 1) * T * pl.LD);
    pl.lAV = take(T * pl.LD); pl.lV = take(T * pl.LD);
    pl.lPART = take(std::max(skmax * T * pl.PLD, T * 4));
    pl.lG = take(T * pl.GLD);
    pl.lZ0 = take(T * pl.ZLD); pl.lZA = take(T * pl.ZLD); pl.lDZ = take(T * pl.ZLD);
    pl.lRED = take(std::max(T, nw) * 4);
    pl.lSC = take(T * 3 + T * std::max(1, n_agents) + 8);
    pl.lPHI = take(T);
    pl.lTRIG = take(T * std::max(1, n_agents) * 6);
    pl.ldsFloats = l;
    if ((size_t)l * 4 > 160 * 1024) return NOCF_E_LDS;
    pl.hN = (float)(1.0 / (nTh - 1));
    *out = pl;
    return 0;
}

static size_t plan_ws_bytes(const DevPlan& pl) {
    return (size_t)(pl.ocw + (long)pl.DB * 64) * sizeof(float);
}

static int fill_prob(const NocfProb* prob, int d, DevProb* pb) {
    if (!prob) return NOCF_E_NULL;
    int ad;
    switch (prob->kind) {
        case NOCF_PROB_CROSS2D: ad = 2; break;
        case NOCF_PROB_SWARMTRAJ: ad = 3; break;
        case NOCF_PROB_QUADCOPTER: ad = 12; break;
        default: return NOCF_E_PROB;
    }
    if (prob->n_agents < 1 || prob->n_agents * ad != d) return NOCF_E_PROB;
    const int ob = prob->obstacle;
    const bool ok = ob == NOCF_OBS_NONE ||
                    (prob->kind == NOCF_PROB_CROSS2D && (ob == NOCF_OBS_SOFTCORRIDOR || ob == NOCF_OBS_HARDCORRIDOR)) ||
                    (prob->kind == NOCF_PROB_SWARMTRAJ && ob == NOCF_OBS_BLOCKS);
    if (!ok) return NOCF_E_PROB;
    if (prob->kind == NOCF_PROB_QUADCOPTER && prob->alph_W > 0.0 && prob->n_agents > 2) return NOCF_E_PROB;
    pb->kind = prob->kind; pb->obstacle = ob; pb->nAgents = prob->n_agents; pb->training = prob->training ? 1 : 0;
    pb->agentDim = ad;
    pb->r = prob->r; pb->alphQ = prob->alph_Q; pb->alphW = prob->alph_W; pb->mass = prob->mass; pb->grav = prob->grav;
    pb->xtarget = prob->xtarget;
    return 0;
}

template <typename KernelT>
static hipError_t set_lds(KernelT kern, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static int pack_weights(const DevPlan& pl, const NocfPhi* phi, float* ws, hipStream_t st) {
    DevPhi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw};
    const long total4 = pl.oWb + (long)(pl.nTh - 1) * pl.strideW;
    int blocks = (int)std::min<long>((total4 + 255) / 256, 2048);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, pl, P, ws);
    return (int)hipGetLastError();
}

static int check_phi(const NocfPhi* phi) {
    if (!phi) return NOCF_E_NULL;
    if (!phi->K0 || !phi->b0 || !phi->K || !phi->b || !phi->w || !phi->A || !phi->cw) return NOCF_E_NULL;
    return 0;
}

// ---- optional in-library timing of the rollout kernel (bench.py): HIP events recorded on the
// launch stream immediately around the kernel, so the figure is the kernel's own duration.
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static bool g_prof_on = false;

extern "C" {

int nocf_version(void) { return NOCF_VERSION; }

int nocf_profile_begin(void) {
    for (auto& pr : g_prof_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    g_prof_events.clear();
    g_prof_on = true;
    return 0;
}

int nocf_profile_end(double* total_ms, int32_t* launches) {
    g_prof_on = false;
    double tot = 0.0;
    for (auto& pr : g_prof_events) {
        hipError_t e = hipEventSynchronize(pr.second);
        if (e) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, pr.first, pr.second);
        if (e) return (int)e;
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int32_t)g_prof_events.size();
    for (auto& pr : g_prof_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    g_prof_events.clear();
    return 0;
}

size_t nocf_workspace_bytes(int32_t d, int32_t m, int32_t nTh) {
    DevPlan pl;
    const int r = std::min(10, d + 1);
    if (make_plan(d, m, nTh, r, 1, &pl) != 0) return 0;
    return plan_ws_bytes(pl);
}

int nocf_ctrl_dim(const NocfProb* prob, int32_t d) {
    if (!prob) return NOCF_E_NULL;
    return prob->kind == NOCF_PROB_QUADCOPTER ? 4 * prob->n_agents : d;
}

int nocf_rollout_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                     double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                     float* z_out, float* persample, float* cost_sums, float* zFull, float* ctrlFull,
                     void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_phi(phi);
    if (rc) return rc;
    if (!x || !alph || !workspace) return NOCF_E_NULL;
    if (n < 1 || nt < 1) return NOCF_E_SHAPE;
    if (stepper != NOCF_RK4 && stepper != NOCF_RK1) return NOCF_E_STEPPER;
    if ((zFull != nullptr) != (ctrlFull != nullptr)) return NOCF_E_NULL;
    if (cost_sums && !persample) return NOCF_E_NULL;
    DevProb pb;
    rc = fill_prob(prob, phi->d, &pb);
    if (rc) return rc;
    DevPlan pl;
    rc = make_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, &pl);
    if (rc) return rc;
    pl.cb = phi->cb;
    if (workspace_bytes < plan_ws_bytes(pl)) return NOCF_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    rc = pack_weights(pl, phi, ws, st);
    if (rc) return rc;
    RollArgs ra;
    ra.x = x; ra.n = n; ra.t0 = t0; ra.t1 = t1; ra.h = (t1 - t0) / nt; ra.nt = nt; ra.stepper = stepper;
    ra.a0 = alph[0];
    ra.z_out = z_out; ra.persample = persample; ra.zFull = zFull; ra.ctrlFull = ctrlFull;
    ra.cdim = nocf_ctrl_dim(prob, phi->d);
    const size_t ldsBytes = (size_t)pl.ldsFloats * 4;
    const int grid = (int)((n + pl.T - 1) / pl.T);
    const int block = pl.nwaves * 64;
    hipError_t e;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_prof_on) {
        if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
        hipEventRecord(ev0, st);
    }
    switch (pl.T / 4) {
        case 1: e = set_lds(rollout_kernel<1>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(rollout_kernel<1>, dim3(grid), dim3(block), ldsBytes, st, pl, pb, ws, ra); break;
        case 2: e = set_lds(rollout_kernel<2>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(rollout_kernel<2>, dim3(grid), dim3(block), ldsBytes, st, pl, pb, ws, ra); break;
        default: e = set_lds(rollout_kernel<4>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(rollout_kernel<4>, dim3(grid), dim3(block), ldsBytes, st, pl, pb, ws, ra); break;
    }
    e = hipGetLastError();
    if (e) return (int)e;
    if (g_prof_on) { hipEventRecord(ev1, st); g_prof_events.emplace_back(ev0, ev1); }
    if (cost_sums) {
        hipLaunchKernelGGL(cost_sum_kernel, dim3(1), dim3(256), 0, st, persample, (long)n, cost_sums);
        e = hipGetLastError();
        if (e) return (int)e;
    }
    return 0;
}

static int phi_common(const NocfPhi* phi, const float* s, int64_t n, float* grad, float* value,
                      void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_phi(phi);
    if (rc) return rc;
    if (!s || !workspace || (!grad && !value)) return NOCF_E_NULL;
    if (n < 1) return NOCF_E_SHAPE;
    DevPlan pl;
    rc = make_plan(phi->d, phi->m, phi->nTh, phi->r, 1, &pl);
    if (rc) return rc;
    pl.cb = phi->cb;
    if (workspace_bytes < plan_ws_bytes(pl)) return NOCF_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    rc = pack_weights(pl, phi, ws, st);
    if (rc) return rc;
    const size_t ldsBytes = (size_t)pl.ldsFloats * 4;
    const int grid = (int)((n + pl.T - 1) / pl.T);
    const int block = pl.nwaves * 64;
    hipError_t e;
    switch (pl.T / 4) {
        case 1: e = set_lds(phi_kernel<1>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(phi_kernel<1>, dim3(grid), dim3(block), ldsBytes, st, pl, ws, s, (long)n, grad, value); break;
        case 2: e = set_lds(phi_kernel<2>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(phi_kernel<2>, dim3(grid), dim3(block), ldsBytes, st, pl, ws, s, (long)n, grad, value); break;
        default: e = set_lds(phi_kernel<4>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(phi_kernel<4>, dim3(grid), dim3(block), ldsBytes, st, pl, ws, s, (long)n, grad, value); break;
    }
    return (int)hipGetLastError();
}

int nocf_phi_grad_f32(const NocfPhi* phi, const float* s, int64_t n, float* grad,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!grad) return NOCF_E_NULL;
    return phi_common(phi, s, n, grad, nullptr, workspace, workspace_bytes, stream);
}

int nocf_phi_forward_f32(const NocfPhi* phi, const float* s, int64_t n, float* value,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!value) return NOCF_E_NULL;
    return phi_common(phi, s, n, nullptr, value, workspace, workspace_bytes, stream);
}

int nocf_prob_eval_f32(const NocfProb* prob, int32_t d, const float* x, const float* p, int64_t n,
                       float* lhqw, float* gradpH, float* ctrls, void* stream) {
    if (!x || !p) return NOCF_E_NULL;
    if (n < 1 || d < 1) return NOCF_E_SHAPE;
    DevProb pb;
    int rc = fill_prob(prob, d, &pb);
    if (rc) return rc;
    DevPlan pl;
    // physics only needs the state/gradient rows; a width-1 network keeps the LDS plan minimal
    rc = make_plan(d, 1, 2, 1, pb.nAgents, &pl);
    if (rc) return rc;
    const size_t ldsBytes = (size_t)pl.ldsFloats * 4;
    const int grid = (int)((n + pl.T - 1) / pl.T);
    hipError_t e = set_lds(prob_kernel, ldsBytes);
    if (e) return (int)e;
    hipLaunchKernelGGL(prob_kernel, dim3(grid), dim3(pl.nwaves * 64), ldsBytes, (hipStream_t)stream, pl, pb, x, p, (long)n,
                       lhqw, gradpH, ctrls, nocf_ctrl_dim(prob, d));
    return (int)hipGetLastError();
}

int nocf_selftest_mfma(const float* a, const float* b, int32_t K, float* out, void* stream) {
    if (!a || !b || !out) return NOCF_E_NULL;
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, K, out);
    return (int)hipGetLastError();
}

}  // extern "C"
