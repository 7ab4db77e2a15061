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
// images (2.7 MB for swarm50) stay in the XCD L2s for the whole rollout.  Each wave streams its
// column block through a 16-deep register ring (up to 16 KiB in flight per wave): at 4..8 samples
// per workgroup the stream is bound by L2->CU bandwidth/latency, not by the MFMA pipe.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>
#include <vector>
#include "nocf.h"

#include "nocf_dev.h"
#ifndef NOCF_JIT_ONLY
#include "nocf_duo.h"
#endif

#ifndef NOCF_MAXTHREADS
#define NOCF_MAXTHREADS 512   // waves per workgroup * 64; 256 gives each wave the whole 512-register file
#endif
#define HALF 8               // k-quads (4 k-steps each) per third of the weight ring
#define MAX_SK 8             // max split-K factor of a GEMM phase
#define MAX_NTH 12

// ------------------------------------------------------------------------------------------
// plan: shapes, packed-image offsets, LDS carve.  Built on the host, passed by value.
// ------------------------------------------------------------------------------------------
struct DevPlan {
    int d, D1, m, r, nTh;
    int MB, DB;                      // 64-column blocks: hidden width, d+1
    int KQ1, KQm;                    // k-quads of the two contraction lengths (d+1, m), padded to HALF
    int SK1, SK6, SKm;               // split-K factors of the opening / closing / residual GEMMs
    int LD, LDs, GLD, ZLD;           // LDS row strides (floats)
    int T, nwaves;
    long oW0f, oW0b, oWf, oWb, strideW;   // float4 offsets of the images in the workspace
    long ob0, ob, ow, ocw, oA, oPlan;     // float offsets of the padded vectors, the copy of A, the plan copy
    float hN, cb;
    // LDS carve (float offsets)
    int lSB, lU0, lU1, lTH, lAV, lV0, lV1, lPART, lG, lZQ, lZ0, lZA, lDZ, lRED, lSC, lPHI, lTRIG, lPT;
    int lVEC, nVEC;                  // biases, w, c.weight and A, copied once per launch (ws floats [ob0, oPlan))
    int bwd;                         // 1: training plan (keeps tanh of the last layer, adjoint arrays, see nocf_bwd.inc)
    int lGB, lAB, lT0B, lQB, lOB, lSBAR, lZQB, lLAM, lXS, lXP, lXD, lSCB, lUB;   // adjoint
    int pMB, pKQc;                   // column blocks per hidden phase / k-quads of the closing phase (diagnostic: halves)
    int ldsFloats;
    int lPW;                         // x-only cost partials formed in the shadow of the residual phases: [2 sets][T][2 waves][2]
    int nAg, pad_;                   // agents of the problem the plan was made for (a literal in the specialised kernels)
};

static_assert(sizeof(DevPlan) % 4 == 0 && sizeof(DevPlan) / 4 <= 256, "plan copy is done by one 256-thread block");
static_assert(sizeof(DevPlan) == 4 * 60 + 8 * 11, "no implicit padding: plans are compared with memcmp");

// ------------------------------------------------------------------------------------------
// plan layout.  constexpr: the host builds the plan of any shape at run time; for the shapes named in
// FIXED_SHAPES the same function is evaluated at COMPILE time and the kernels are instantiated with every
// stride, offset and trip count as a literal (no scalar loads of plan fields between the phases).
// ------------------------------------------------------------------------------------------
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int rup(int a, int b) { return cdiv(a, b) * b; }
constexpr int imax(int a, int b) { return a > b ? a : b; }

// split-K factor minimising the makespan of nblk column blocks over nwaves waves
constexpr int choose_sk(int nblk, int halves, int nwaves, int cap) {
    // makespan in ring halves (8 k-quads); a wave's first unit is prefetched by the previous phase, every
    // further unit pays ~3 halves of pipeline fill; a split costs one more barrier + an LDS pass
    int best = 1; long bestCost = -1;
    for (int sk = 1; sk <= cap && sk <= halves; ++sk) {
        const long rounds = cdiv(nblk * sk, nwaves);
        const long cost = (rounds * cdiv(halves, sk) + (rounds - 1) * 3) * 16 + (sk > 1 ? 24 + 2 * sk : 0);
        if (bestCost < 0 || cost < bestCost) { bestCost = cost; best = sk; }
    }
    return best;
}

// fills the shape / image part of the plan; returns 0 or an NOCF_E_* code.  nw_req / S_req / diagHalf: 0 = default.
constexpr int plan_layout(int d, int m, int nTh, int r, int n_agents, int bwd, int nw_req, int S_req, int diagHalf, DevPlan& out) {
    if (d < 1 || m < 1 || nTh < 2 || nTh > MAX_NTH || r < 1 || r > d + 1 || r > ZQLD) return NOCF_E_SHAPE;
    if (n_agents > 255) return NOCF_E_SHAPE;
    DevPlan pl{};
    pl.d = d; pl.D1 = d + 1; pl.m = m; pl.r = r; pl.nTh = nTh; pl.bwd = bwd; pl.nAg = n_agents;
    pl.MB = cdiv(m, 64); pl.DB = cdiv(pl.D1, 64);
    pl.KQ1 = rup(cdiv(pl.D1, 4), HALF); pl.KQm = rup(cdiv(m, 4), HALF);
    // geometry: waves per workgroup and sample sub-tiles
    int nw = 1;
    while (nw < NOCF_MAXTHREADS / 64 && nw < pl.MB) nw *= 2;
    int S = 1;
    if (nw_req) nw = nw_req;
    if (S_req) S = S_req;
    if (!(nw == 1 || nw == 2 || nw == 4 || nw == 8) || nw * 64 > NOCF_MAXTHREADS || !(S == 1 || S == 2 || S == 4)) return NOCF_E_SHAPE;
    pl.nwaves = nw; pl.T = 4 * S;
    // LDS row strides: 64j+4 floats keeps the four sample rows of an A-operand read on distinct 16-B slots
    pl.LD = rup(imax(pl.KQm * 4, pl.MB * 64), 64) + 4;
    pl.LDs = rup(pl.KQ1 * 4, 64) + 4;
    pl.GLD = pl.DB * 64;
    pl.ZLD = rup(d + 4, 4);
    // packed images
    long o4 = 0;
    pl.oW0f = o4; o4 += (long)pl.MB * pl.KQ1 * 64;
    pl.oW0b = o4; o4 += (long)pl.DB * pl.KQm * 64;
    pl.strideW = (long)pl.MB * pl.KQm * 64;
    pl.oWf = o4; o4 += (long)(nTh - 1) * pl.strideW;
    pl.oWb = o4; o4 += (long)(nTh - 1) * pl.strideW;
    long of = o4 * 4;
    pl.ob0 = of; of += (long)pl.MB * 64;
    pl.ob = of; of += (long)(nTh - 1) * pl.MB * 64;
    pl.ow = of; of += (long)pl.MB * 64;
    pl.ocw = of; of += (long)pl.DB * 64;
    pl.oA = of; of += rup(r * (d + 1), 4);
    pl.oPlan = of;
    // LDS carve; the split-K cap shrinks until the partial-sum slots fit next to the activations
    const int T = pl.T;
    (void)n_agents;
    int l = 0;
    for (int cap = bwd ? 4 : MAX_SK; cap >= 1; cap >>= 1) {
        pl.pMB = diagHalf ? imax(1, pl.MB / 2) : pl.MB;          // diagHalf: timing experiment only (results are wrong)
        pl.pKQc = diagHalf ? imax(HALF, (pl.KQm / 2) / HALF * HALF) : pl.KQm;
        pl.SK1 = choose_sk(pl.pMB, pl.KQ1 / HALF, nw, cap);
        pl.SK6 = choose_sk(pl.DB, pl.pKQc / HALF, nw, cap);
        pl.SKm = choose_sk(pl.pMB, pl.KQm / HALF, nw, cap);
        int partFloats = 4;
        if (pl.SK1 > 1) partFloats = imax(partFloats, pl.SK1 * T * pl.MB * 64);
        if (pl.SKm > 1) partFloats = imax(partFloats, pl.SKm * T * pl.MB * 64);
        if (pl.SK6 > 1) partFloats = imax(partFloats, pl.SK6 * T * pl.DB * 64);
        l = 0;
        const int Lr = nTh - 1, extra = (bwd && nTh > 2) ? (nTh - 2) * T * pl.LD : 0;
        pl.lSB = l; l += rup(T * pl.LDs, 4);
        pl.lU0 = l; l += rup(T * pl.LD, 4);
        pl.lU1 = l; l += rup(T * pl.LD + extra, 4);                 // adjoint plans: u_0 .. u_L
        pl.lTH = l; l += rup((nTh - 1 + (bwd ? 1 : 0)) * T * pl.LD, 4);
        pl.lAV = l; l += rup((bwd ? Lr : 1) * T * pl.LD, 4);        // adjoint plans: a_{L-1} .. a_0
        pl.lV0 = l; l += rup(T * pl.LD, 4);
        pl.lV1 = l; l += rup(T * pl.LD + extra, 4);                 // adjoint plans: v_L .. v_1, y
        pl.lPART = l; l += rup(partFloats, 4);
        pl.lG = l; l += rup(T * pl.GLD, 4);
        pl.lZQ = l; l += rup(T * ZQLD, 4);
        pl.lZ0 = l; l += rup(T * pl.ZLD, 4);
        pl.lZA = l; l += rup(T * pl.ZLD, 4);
        pl.lDZ = l; l += rup(T * pl.ZLD, 4);
        pl.lRED = l; l += rup(imax(T, nw) * 4, 4);
        pl.lSC = l; l += rup(imax(T * imax(1, n_agents) + 8, T * 4 + 8), 4);
        pl.lPHI = l; l += rup(T, 4);
        pl.lTRIG = l; l += rup(T * imax(1, n_agents) * 6, 4);
        pl.lPT = l; l += 4;                                         // (spare)
        pl.lPW = l; l += rup(2 * T * 4 + T * 4, 4);                 // two sets (the deferred cost side reads set e while e+1 fills
                                                                    // the other), then sum p^2 per sample and column block [T][<=4]
        pl.nVEC = (int)(pl.oPlan - pl.ob0);
        pl.lVEC = l; l += rup(pl.nVEC, 4);
        if (bwd) {
            pl.lGB = l; l += rup(T * pl.LDs, 4);
            pl.lAB = l; l += rup(Lr * T * pl.LD, 4);
            pl.lT0B = l; l += rup(Lr * T * pl.LD, 4);
            pl.lQB = l; l += rup(Lr * T * pl.LD, 4);
            pl.lOB = l; l += rup(T * pl.LD, 4);
            pl.lUB = l; l += rup(Lr > 1 ? T * pl.LD : 4, 4);
            pl.lSBAR = l; l += rup(T * pl.GLD, 4);
            pl.lZQB = l; l += rup(T * ZQLD, 4);
            pl.lLAM = l; l += rup(T * pl.ZLD, 4);
            pl.lXS = l; l += rup(T * pl.ZLD, 4);
            pl.lXP = l; l += rup(T * pl.ZLD, 4);
            pl.lXD = l; l += rup(T * pl.ZLD, 4);
            pl.lSCB = l; l += rup(T * 4 + 8, 4);
        }
        l += 64;                                    // slack: the activation ring's last prefetch reads 32 floats past a row
        if ((long)l * 4 <= 160 * 1024) break;
    }
    pl.ldsFloats = l;
    if ((long)l * 4 > 160 * 1024) return NOCF_E_LDS;
    pl.hN = (float)(1.0 / (nTh - 1));
    out = pl;
    return 0;
}

// where a kernel takes its plan from: the record in the workspace (any shape) or a compile-time constant
struct DynPlan { static constexpr bool fixed = false; };
template <int D, int M, int NTH, int R, int NAG, int BWD>
struct FixedPlan {
    static constexpr bool fixed = true;
    static constexpr DevPlan make() { DevPlan p{}; (void)plan_layout(D, M, NTH, R, NAG, BWD, 0, 0, 0, p); return p; }
};
// shapes with a specialised instantiation: (d, m, nTh, r, agents).  First the BASELINE.json tile-kernel configurations
// (swarm50, singlequad), then the initProb problems whose d+1 exceeds the lane kernel's 32 at the reference's default
// width m = 32 (midcross20, midcross30, swarm).
// A build with -DNOCF_XS_D=d -DNOCF_XS_M=m -DNOCF_XS_T=nTh -DNOCF_XS_R=r -DNOCF_XS_A=agents adds one more (evaluation,
// record and adjoint); with -DNOCF_JIT_ONLY it carries only that one: neuraloc_amd/_lib.py builds such a library per
// shape on request (NOCF_JIT=1).
#ifdef NOCF_XS_D
#define FIXED_SHAPES_EXTRA(X) X(NOCF_XS_D, NOCF_XS_M, NOCF_XS_T, NOCF_XS_R, NOCF_XS_A)
#else
#define FIXED_SHAPES_EXTRA(X)
#endif
#ifdef NOCF_JIT_ONLY
#define FIXED_SHAPES(X)
#define FIXED_SHAPES_TRAIN(X)
#else
#define FIXED_SHAPES(X) X(150, 512, 2, 10, 50) X(12, 128, 2, 10, 1) X(40, 32, 2, 10, 20) X(60, 32, 2, 10, 30) X(96, 32, 2, 10, 32)
// small shapes: evaluation takes the lane kernel, TRAINING (record + adjoint) takes the tile kernels -- the other
// BASELINE configs (swap2, softcorridor = midcross2 = swap12_1pair, swap12) and the remaining initProb problems at m = 32
#define FIXED_SHAPES_TRAIN(X) X(4, 16, 2, 5, 2) X(4, 32, 2, 5, 2) X(24, 32, 2, 10, 12) \
    X(8, 32, 2, 9, 4) X(12, 32, 2, 10, 6) X(16, 32, 2, 10, 8) X(20, 32, 2, 10, 10)
#endif

template <class SP>
static bool plan_is(const DevPlan& run) {
    static const DevPlan fixed = SP::make();
    DevPlan a = run, b = fixed;
    a.cb = 0.f; b.cb = 0.f;
    return memcmp(&a, &b, sizeof(DevPlan)) == 0;
}


// ------------------------------------------------------------------------------------------
// weight packing: image[cb][kq][lane] = float4 of B[k = 4kq..4kq+3][col = 64cb + lane]
// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(DevPlan pl, DevPhi P, float* __restrict__ ws) {
    float4* ws4 = reinterpret_cast<float4*>(ws);
    const long nW0f = (long)pl.MB * pl.KQ1 * 64;       // opening:  B[k][j] = K0[j][k]
    const long nW0b = (long)pl.DB * pl.KQm * 64;       // closing:  B[k][i] = K0[k][i]
    const long nWl = (long)pl.MB * pl.KQm * 64;
    const long nLayers = pl.nTh - 1;
    const long total4 = nW0f + nW0b + 2 * nLayers * nWl;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += stride) {
        float v[4];
        long dst;
        if (idx < nW0f) {
            long q = idx;
            const int lane = q & 63; q >>= 6;
            const int kq = q % pl.KQ1, cb = q / pl.KQ1;
            const int j = cb * 64 + lane;
            for (int e = 0; e < 4; ++e) {
                const int k = kq * 4 + e;
                v[e] = (j < pl.m && k < pl.D1) ? P.K0[(long)j * pl.D1 + k] : 0.f;
            }
            dst = pl.oW0f + idx;
        } else if (idx < nW0f + nW0b) {
            long q = idx - nW0f;
            const int lane = q & 63; q >>= 6;
            const int kq = q % pl.KQm, cb = q / pl.KQm;
            const int i = cb * 64 + lane;
            for (int e = 0; e < 4; ++e) {
                const int k = kq * 4 + e;
                v[e] = (k < pl.m && i < pl.D1) ? P.K0[(long)k * pl.D1 + i] : 0.f;
            }
            dst = pl.oW0b + (idx - nW0f);
        } else {
            const long q = idx - nW0f - nW0b;
            const int layer = q / (2 * nWl);             // 0 -> reference layer 1
            long w = q - (long)layer * 2 * nWl;
            const int back = w >= nWl;
            if (back) w -= nWl;
            const int lane = w & 63; const long ww = w >> 6;
            const int kq = ww % pl.KQm, cb = ww / pl.KQm;
            const float* Kl = P.K + (long)layer * pl.m * pl.m;
            const int a = cb * 64 + lane;
            for (int e = 0; e < 4; ++e) {
                const int k = kq * 4 + e;
                float val = 0.f;
                if (a < pl.m && k < pl.m) val = back ? Kl[(long)k * pl.m + a]    // backward: B[j][col] = K[j][col]
                                                     : Kl[(long)a * pl.m + k];   // forward:  B[k][j]   = K[j][k]
                v[e] = val;
            }
            dst = (back ? pl.oWb : pl.oWf) + (long)layer * pl.strideW + w;
        }
        ws4[dst] = make_float4(v[0], v[1], v[2], v[3]);
    }
    // padded vectors
    const long nb0 = (long)pl.MB * 64, nb = nLayers * pl.MB * 64, nw = (long)pl.MB * 64, ncw = (long)pl.DB * 64;
    const long totalv = nb0 + nb + nw + ncw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < totalv; idx += stride) {
        float val = 0.f;
        long dst;
        if (idx < nb0) { if (idx < pl.m) val = P.b0[idx]; dst = pl.ob0 + idx; }
        else if (idx < nb0 + nb) {
            const long q = idx - nb0; const int layer = q / (pl.MB * 64); const int c = q % (pl.MB * 64);
            if (c < pl.m) val = P.b[(long)layer * pl.m + c];
            dst = pl.ob + q;
        } else if (idx < nb0 + nb + nw) { const long q = idx - nb0 - nb; if (q < pl.m) val = P.w[q]; dst = pl.ow + q; }
        else { const long q = idx - nb0 - nb - nw; if (q < pl.D1) val = P.cw[q]; dst = pl.ocw + q; }
        ws[dst] = val;
    }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)pl.r * pl.D1; idx += stride)
        ws[pl.oA + idx] = P.A[idx];
    // the plan itself, for the kernels that read it through a pointer
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevPlan) / 4) {
        const unsigned* src = reinterpret_cast<const unsigned*>(&pl);
        unsigned v = src[threadIdx.x];
        if (P.cbp && threadIdx.x == offsetof(DevPlan, cb) / 4) v = __float_as_uint(*P.cbp);
        reinterpret_cast<unsigned*>(ws + pl.oPlan)[threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    // D[blk][i][j] += A[blk][i] * B[blk][j];  lane = 4*blk + (i for A | j for B,D);  D reg = i
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}



struct Ctx {
    const float* ws;
    __amdgpu_buffer_rsrc_t wrs;      // buffer descriptor over the packed workspace (wave-uniform)
    int tid, nthreads, wave, lane;
    float cb;                        // c.bias (the only plan field that is not a function of the shape)
#ifdef NOCF_STAMPS
#if NOCF_STAMPS >= 2
    mutable unsigned long long acc[12];      // per-phase cycle accumulators (24 VGPRs: they distort the kernel, level 2 only)
    mutable unsigned long long last;
#endif
    unsigned long long* tl;          // timeline of ONE evaluation of one workgroup: [wave][64 points] (null otherwise)
#endif
};

__device__ __forceinline__ void ctx_init(Ctx& c, const float* ws, unsigned ws_bytes) {
    c.ws = ws;
    c.wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, ws_bytes, 0x00020000);
    c.tid = threadIdx.x; c.nthreads = blockDim.x; c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // provably wave-uniform: scalar loop control
#ifdef NOCF_STAMPS
#if NOCF_STAMPS >= 2
    for (int i = 0; i < 12; ++i) c.acc[i] = 0;
    c.last = clock64();
#endif
    c.tl = nullptr;
#endif
}

// one packed k-quad of weights for this lane: 16 B at (uniform byte offset soff) + voff, voff = lane*16, or
// WOOB for lanes whose column is padding: the descriptor's range check returns 0 without fetching
#define WOOB 0x7ffffff0
__device__ __forceinline__ float4 wload(const Ctx& c, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(c.wrs, voff, soff, 0);
    float4 f;
    f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}

// Diagnostic build only (-DNOCF_STAMPS): thread 0 of every workgroup accumulates the shader cycles
// each phase takes (barrier waits included).  The production library contains no stamp.
#ifdef NOCF_STAMPS
#if NOCF_STAMPS >= 2
#define STAMP(c, id) do { if ((c).tid == 0) { unsigned long long t_ = clock64(); (c).acc[id] += t_ - (c).last; (c).last = t_; } } while (0)
#else
#define STAMP(c, id) do { } while (0)
#endif
#define TL(c, id) do { if ((c).tl && (c).lane == 0) (c).tl[(c).wave * 64 + (id)] = clock64(); } while (0)
__device__ unsigned long long* g_tl_dev = nullptr;       // timeline buffer of the adjoint kernel (set with the stamp buffer)
#else
#define STAMP(c, id) do { } while (0)
#define TL(c, id) do { } while (0)
#endif

// The weight ring: two halves of 8 k-quads (8 KiB per wave each).  It lives in the kernel's scope, not
// in gemm_phase, because the first turn of the NEXT phase is loaded before the current phase's
// epilogue, barrier and (at the end of an evaluation) the physics: weights do not depend on data.
struct Ring { float4 A[HALF], B[HALF]; };
struct PhaseDesc { long img4; int nblk, KQ, SK, ncols; };

__device__ __forceinline__ void unit_range(const PhaseDesc& ph, int u, int& cb, int& ks, int& h0, int& nh) {
    cb = u / ph.SK; ks = u - cb * ph.SK;
    const int halves = ph.KQ / HALF;
    h0 = (halves * ks) / ph.SK;
    nh = (halves * (ks + 1)) / ph.SK - h0;
}

__device__ __forceinline__ void ring_preload(const Ctx& c, Ring& rg, const PhaseDesc& ph, int u) {
    int cb, ks, h0, nh;
    unit_range(ph, u, cb, ks, h0, nh);
    const int voff = (cb * 64 + c.lane < ph.ncols) ? c.lane * 16 : WOOB;
    const int wo = (int)((ph.img4 + ((long)cb * ph.KQ + (long)h0 * HALF) * 64) * 16);
    if (nh > 0) {
#pragma unroll
        for (int i = 0; i < HALF; ++i) rg.A[i] = wload(c, voff, wo + i * 1024);
    }
    if (nh > 1) {
#pragma unroll
        for (int i = 0; i < HALF; ++i) rg.B[i] = wload(c, voff, wo + (HALF + i) * 1024);
    }
}

// One half of the weight ring: 8 k-quads = 32 k-steps = 32*S MFMAs.  Two register rings feed it:
//   buf[8]  weights of this half; with RW each consumed entry is reloaded with the k-quad 16 further
//           on (the ring's other half is in flight meanwhile: 8..16 KiB per wave outstanding)
//   a[S][8] activations (LDS); with RA each consumed entry is reloaded with the k-quad 8 further on,
//           so no MFMA waits on an LDS read it has just issued.
// Weights come through buffer loads (scalar byte offset + lane*16: no per-load 64-bit address
// arithmetic); no branch sits between the loads, so the waits are counted vmcnt/lgkmcnt.
template <int S> struct ActRing { static constexpr int AH = (S == 1) ? HALF : HALF / 2; };   // k-quads of activations kept ahead

template <int S, bool RW, bool RA>
__device__ __forceinline__ void ring_half(const Ctx& c, float4 (&buf)[HALF], float4 (&a)[S][ActRing<S>::AH], f32x4 (&acc)[S][4],
                                          int a_cur /*float4 index of THIS half's quad 0, this lane's row*/, int ld /*floats*/,
                                          int voff, int w_next /*bytes*/) {
    constexpr int AH = ActRing<S>::AH;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float4 w = buf[i];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float4 av = a[s][i % AH];
            acc[s][0] = mfma4(av.x, w.x, acc[s][0]);
            acc[s][1] = mfma4(av.y, w.y, acc[s][1]);
            acc[s][2] = mfma4(av.z, w.z, acc[s][2]);
            acc[s][3] = mfma4(av.w, w.w, acc[s][3]);
            // quad i+AH of the stream (the next half when it runs past this one; RA says whether that half exists)
            if (RA || i + AH < HALF) a[s][i % AH] = reinterpret_cast<const float4*>(lds)[a_cur + s * ld + i + AH];
        }
        if (RW) buf[i] = wload(c, voff, w_next + i * 1024);
    }
}

__device__ __forceinline__ float part_sum(const float* part, int pstride, int SK, int off) {
    float v = part[off];
    for (int k = 1; k < SK; ++k) v += part[k * pstride + off];
    return v;
}

// out[t][col] = epi(t, col, sum_k act[t][k] * B[k][col]) for ph.nblk 64-column blocks.
//   ph.img4  float4 offset of the packed image [nblk][KQ][64] in the workspace
//   act_off  float offset in LDS of the activation rows [T][ld]
//   pre      the ring already holds the first turn of this wave's first unit (loaded by the previous phase)
//   nxt      the phase that follows (nxt.SK == 0: none); its first turn is loaded as soon as this wave has
//            consumed its last weights, i.e. before the epilogue and the barrier
// SK == 1: the wave that owns a column block applies epi straight from its accumulators.
// SK  > 1: split-K partials go through LDS (fixed-order sum, deterministic); contains one barrier.
// The caller puts a barrier after the call before anyone reads what epi wrote.
struct NoPost { __device__ __forceinline__ void operator()() const {} };

// post(): runs on every wave after its own units and epilogues, before the caller's barrier -- work placed there
// fills the time a wave that finished streaming early would otherwise spend waiting for the slowest one
struct NoIdle { __device__ __forceinline__ void operator()(int, int) const {} };

// idle(i, n): runs on the waves that own no unit of this phase (wave i of n such waves) while the others stream
template <int S, class Epi, class Post = NoPost, class Idle = NoIdle>
__device__ __forceinline__ void gemm_phase(const Ctx& c, const DevPlan& pl, Ring& rg, bool pre, const PhaseDesc& ph,
                                           const PhaseDesc& nxt, int act_off, int ld, Epi epi, int stamp_id = 11, Post post = Post(),
                                           Idle idle = Idle()) {
    const int partLD = ph.nblk * 64;
    const int pstride = pl.T * partLD;
    const int units = ph.nblk * ph.SK;
    const int arow = c.lane & 3;
    const bool want_next = nxt.SK > 0 && c.wave < nxt.nblk * nxt.SK;
    bool next_done = false;
    const int tlb = 8 + (stamp_id >> 1) * 8;             // timeline points of this phase: tlb .. tlb+5
    (void)tlb;
    TL(c, tlb);
    for (int u = c.wave; u < units; u += pl.nwaves) {
        int cb, ks, h0, nh;
        unit_range(ph, u, cb, ks, h0, nh);
        if (!(pre && u == c.wave)) ring_preload(c, rg, ph, u);
        f32x4 acc[S][4];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[s][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int voff = (cb * 64 + c.lane < ph.ncols) ? c.lane * 16 : WOOB;
        int wo = (int)((ph.img4 + ((long)cb * ph.KQ + (long)h0 * HALF) * 64) * 16);   // byte offset, wave-uniform
        int ao = (act_off + arow * ld) / 4 + h0 * HALF;                               // float4 index, per lane
        float4 av[S][ActRing<S>::AH];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int i = 0; i < ActRing<S>::AH; ++i) av[s][i] = reinterpret_cast<const float4*>(lds)[ao + s * ld + i];
        TL(c, tlb + 1);
        // halves alternate between rg.A and rg.B; half h is refilled with half h+2 while it is consumed
        int h = 0;
        // never unrolled: with a compile-time trip count the scheduler otherwise sinks every refill load down to
        // its first use (vmcnt(1) everywhere) and the ring's prefetch distance is gone
#pragma clang loop unroll(disable)
        for (; h + 3 < nh; h += 2) {
            ring_half<S, true, true>(c, rg.A, av, acc, ao, ld, voff, wo + 2 * HALF * 1024);
            ring_half<S, true, true>(c, rg.B, av, acc, ao + HALF, ld, voff, wo + 3 * HALF * 1024);
            wo += 2 * HALF * 1024;
            ao += 2 * HALF;
        }
        const int rem = nh - h;                         // 1, 2 or 3 halves left
        if (rem == 3) {
            ring_half<S, true, true>(c, rg.A, av, acc, ao, ld, voff, wo + 2 * HALF * 1024);
            ring_half<S, false, true>(c, rg.B, av, acc, ao + HALF, ld, voff, wo);
            ring_half<S, false, false>(c, rg.A, av, acc, ao + 2 * HALF, ld, voff, wo);
        } else if (rem == 2) {
            ring_half<S, false, true>(c, rg.A, av, acc, ao, ld, voff, wo);
            ring_half<S, false, false>(c, rg.B, av, acc, ao + HALF, ld, voff, wo);
        } else if (rem == 1) {
            ring_half<S, false, false>(c, rg.A, av, acc, ao, ld, voff, wo);
        }
        if (u + pl.nwaves >= units) {
            STAMP(c, stamp_id);                          // diagnostic: end of this wave's streaming
            TL(c, tlb + 2);
            if (want_next) { ring_preload(c, rg, nxt, c.wave); next_done = true; }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const f32x4 rsum = (acc[s][0] + acc[s][1]) + (acc[s][2] + acc[s][3]);
            if (ph.SK == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) epi(4 * s + q, cb * 64 + c.lane, rsum[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    lds[pl.lPART + ks * pstride + (4 * s + q) * partLD + cb * 64 + c.lane] = rsum[q];
            }
        }
    }
    if (want_next && !next_done) ring_preload(c, rg, nxt, c.wave);     // waves without a unit in this phase
    if (c.wave >= units) idle(c.wave - units, pl.nwaves - units);
    post();
    TL(c, tlb + 3);
    if (ph.SK > 1) {
        __syncthreads();
        TL(c, tlb + 4);
        // all T*partLD outputs in one flat sweep over the whole workgroup (a closing phase has 3 column blocks: a
        // per-sample loop would leave five of eight waves idle while the others walk the samples one after another)
        for (int j = c.tid; j < pstride; j += c.nthreads) {
            const int t = (j >> 6) / ph.nblk, col = j - t * partLD;
            epi(t, col, part_sum(lds + pl.lPART, pstride, ph.SK, j));
        }
    }
    TL(c, tlb + 5);
}

// Segmented reduction of NV values per thread.  Threads are split into T groups of
// G = nthreads/T consecutive threads (group t serves sample t).  After the call (it ends with a
// barrier) group_total(...) gives any thread the total of sample t.
template <int NV>
__device__ __forceinline__ void group_reduce(const Ctx& c, const DevPlan& pl, float (&v)[NV]) {
    const int G = c.nthreads / pl.T;
    const int seg = G < 64 ? G : 64;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = sum_seg(v[i], seg);
    float* red = lds + pl.lRED;
    if ((c.tid & (seg - 1)) == 0) {
        const int subseg = c.tid / seg;
#pragma unroll
        for (int i = 0; i < NV; ++i) red[subseg * NV + i] = v[i];
    }
    __syncthreads();
}

template <int NV>
__device__ __forceinline__ float group_total(const Ctx& c, const DevPlan& pl, int t, int i) {
    const int G = c.nthreads / pl.T;
    const int R = G < 64 ? 1 : G / 64;          // sub-segments (waves) per sample
    const float* red = lds + pl.lRED;
    float s = red[(t * R) * NV + i];
    for (int q = 1; q < R; ++q) s += red[(t * R + q) * NV + i];
    return s;
}

// biases, w, c.weight and A: workspace -> LDS, once per launch (before the first barrier of the kernel)
__device__ __forceinline__ void load_vectors(const Ctx& c, const DevPlan& pl) {
    for (int i = c.tid; i < pl.nVEC; i += c.nthreads) lds[pl.lVEC + i] = c.ws[pl.ob0 + i];
}

// ------------------------------------------------------------------------------------------
// grad Phi (and optionally Phi) for the T samples whose s=[x,t] rows sit in SB.
// src/Phi.py:99-138 restated per sample row; results: G[t][0..d]; if need_value PHI[t]=Phi(s).
// Every thread of the workgroup must call it (it contains barriers).
// ------------------------------------------------------------------------------------------
// z = A s for the T samples in SB (the low-rank quadratic's inner product; A is at most 16 x (d+1)), by the `nth`
// threads numbered tid_i: 8 lanes share a row; four strided products per turn so the LDS reads go out together.
// Every caller accumulates in the same order, so the value does not depend on who computes it.
__device__ __forceinline__ void z_from_s(const DevPlan& pl, int tid_i, int nth) {
    const int T = pl.T, r = pl.r, D1 = pl.D1;
    const float* Araw = lds + pl.lVEC - pl.ob0 + pl.oA;
    const int items = T * r * 8;
    for (int base = 0; base < items; base += nth) {
        const int id = base + tid_i;
        const int part = id & 7, tr = id >> 3;
        float acc = 0.f;
        if (id < items) {
            const int t = tr / r, rr = tr - t * r;
            const float* arow = Araw + rr * D1;
            const float* srow = lds + pl.lSB + t * pl.LDs;
            for (int i = part; i < D1; i += 32) {
                float av[4], sv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int iu = i + 8 * u;
                    const int ic = iu < D1 ? iu : D1 - 1;
                    av[u] = iu < D1 ? arow[ic] : 0.f;
                    sv[u] = srow[ic];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += av[u] * sv[u];
            }
        }
        acc = sum8(acc);
        if (id < items && part == 0) { const int t = tr / r; lds[pl.lZQ + t * ZQLD + (tr - t * r)] = acc; }
    }
}

struct NoShadow { __device__ __forceinline__ void operator()(int) const {} };

// shadow(k): extra work of the caller run by every wave at the end of the last forward residual phase (k = 0) and
// of the first backward residual phase (k = 1), i.e. while the slower waves of those long phases still stream;
// shadow(2) runs on the waves without a unit in the closing phase, after their share of z = A s
template <int S, class Shadow = NoShadow>
__device__ void phi_eval(const Ctx& c, const DevPlan& pl, bool need_value, Ring& rg, bool& ring_ready, bool more_evals,
                         bool z_ready = false /* z = A s already sits in ZQ (the rollout's RK tail computed it) */,
                         Shadow shadow = Shadow(),
                         bool want_p2 = false /* leave sum_i g_i^2 (i < d) per sample and column block behind the PW sets */) {
    const int T = pl.T, LD = pl.LD, m = pl.m, D1 = pl.D1, r = pl.r;
    const int oSB = pl.lSB, oTH = pl.lTH, oAV = pl.lAV, oG = pl.lG, oZQ = pl.lZQ;
    const float hN = pl.hN;
    // small vectors live in LDS (load_vectors): an epilogue must not issue a global load, it would queue
    // behind the next phase's weight prefetch (loads return in order)
    const float* vec = lds + pl.lVEC - pl.ob0;
    const float* b0 = vec + pl.ob0;
    const float* wv = vec + pl.ow;
    const float* cw = vec + pl.ocw;
    const float* Araw = vec + pl.oA;
    const int lastLayer = pl.nTh - 1;
    const PhaseDesc phOpen = {pl.oW0f, pl.pMB, pl.KQ1, pl.SK1, m};
    const PhaseDesc phClose = {pl.oW0b, pl.DB, pl.pKQc, pl.SK6, D1};
    const PhaseDesc phNone = {0, 0, 0, 0, 0};
    auto phFwd = [&](int i) { return PhaseDesc{pl.oWf + (long)(i - 1) * pl.strideW, pl.pMB, pl.KQm, pl.SKm, m}; };
    auto phBwd = [&](int i) { return PhaseDesc{pl.oWb + (long)(i - 1) * pl.strideW, pl.pMB, pl.KQm, pl.SKm, m}; };

    STAMP(c, 10);
    TL(c, 0);
    // the opening weights do not wait for z: start them first (unless the previous evaluation already did)
    if (!ring_ready && c.wave < phOpen.nblk * phOpen.SK) ring_preload(c, rg, phOpen, c.wave);
    // ---- z = A s: needed by the closing epilogue (and by Phi itself).  When the closing phase leaves waves without
    // a unit they compute it there, beside the weight stream; otherwise, or when Phi is wanted, it is done here.
    const int closeUnits = pl.DB * pl.SK6;
    const bool z_in_closing = closeUnits < pl.nwaves && pl.SK6 > 1 && !need_value;   // (the split-K barrier orders it before the epilogue)
    if (!z_ready && !z_in_closing) z_from_s(pl, c.tid, c.nthreads);
    TL(c, 1);
    // ---- opening layer: o = s K0^T + b0 ; u0 = sigma(o) ; gate0 = tanh(o)
    gemm_phase<S>(c, pl, rg, true, phOpen, phFwd(1), oSB, pl.LDs, [&](int t, int col, float v) {
        if (col < m) {
            const float o = v + b0[col];
            float sg, th;
            act_pair(o, sg, th);
            lds[pl.lU0 + t * LD + col] = sg;
            lds[oTH + t * LD + col] = th;
        }
    }, 1);
    __syncthreads();
    TL(c, 2);
    STAMP(c, 0);
    // adjoint plans keep every layer's u, a and v (phi_vjp reads them); forward plans ping-pong two arrays
    const int TLD = T * LD;
    auto oUs = [&](int i) { return pl.bwd ? pl.lU0 + i * TLD : ((i & 1) ? pl.lU1 : pl.lU0); };     // u_i
    auto oVs = [&](int k) { return pl.bwd ? pl.lV0 + k * TLD : ((k & 1) ? pl.lV1 : pl.lV0); };     // v_i at k = L-i (y at L)
    auto oAs = [&](int k) { return pl.bwd ? oAV + k * TLD : oAV; };                                  // a_i at k = max(L-1-i, 0)
    // ---- residual layers, forward
    for (int i = 1; i <= lastLayer; ++i) {
        const float* bi = vec + pl.ob + (long)(i - 1) * pl.MB * 64;
        const int oTHi = oTH + i * T * LD;
        const int oUc = oUs(i - 1), oUn = oUs(i);
        gemm_phase<S>(c, pl, rg, true, phFwd(i), (i < lastLayer) ? phFwd(i + 1) : phBwd(lastLayer), oUc, LD,
                      [&](int t, int col, float v) {
            if (col < m) {
                const float q = v + bi[col];
                if (i < lastLayer) {
                    float sg, th;
                    act_pair(q, sg, th);
                    lds[oTHi + t * LD + col] = th;
                    lds[oUn + t * LD + col] = lds[oUc + t * LD + col] + hN * sg;
                } else {
                    const float wj = wv[col];
                    const float th = tanh_fast(q);
                    lds[pl.lV0 + t * LD + col] = th * wj;
                    lds[oAV + t * LD + col] = wj;
                    if (pl.bwd) lds[oTHi + t * LD + col] = th;         // the adjoint needs tanh(q) itself
                    if (need_value) lds[oUn + t * LD + col] = lds[oUc + t * LD + col] + hN * sigma_act(q);
                }
            }
        }, 3, [&]() { if (i == lastLayer) shadow(0); });
        __syncthreads();
    }
    TL(c, 3);
    STAMP(c, 2);
    // ---- Phi itself (final time only): w.u + 1/2 |A s|^2 + c.s + cb   (src/Phi.py:91-96)
    if (need_value) {
        const int Gsz = c.nthreads / T;
        const int t = c.tid / Gsz, j0 = c.tid - t * Gsz;
        const int oU = oUs(lastLayer);
        float acc[1] = {0.f};
        for (int col = j0; col < m; col += Gsz) acc[0] += wv[col] * lds[oU + t * LD + col];
        for (int q = j0; q < r; q += Gsz) { const float z = lds[oZQ + t * ZQLD + q]; acc[0] += 0.5f * z * z; }
        for (int i = j0; i < D1; i += Gsz) acc[0] += cw[i] * lds[oSB + t * pl.LDs + i];
        group_reduce<1>(c, pl, acc);
        if (c.tid < T) lds[pl.lPHI + c.tid] = group_total<1>(c, pl, c.tid, 0) + c.cb;
        __syncthreads();
    }
    // ---- backward sweep: a <- a + hN K_i^T (tanh(.) . a)
    for (int i = lastLayer; i >= 1; --i) {
        const int oTHp = oTH + (i - 1) * T * LD;
        const int oVc = oVs(lastLayer - i), oVn = oVs(lastLayer - i + 1);
        const int oAc = oAs((lastLayer - 1 - i) > 0 ? (lastLayer - 1 - i) : 0), oAn = oAs(lastLayer - i);
        gemm_phase<S>(c, pl, rg, true, phBwd(i), (i > 1) ? phBwd(i - 1) : phClose, oVc, LD,
                      [&](int t, int col, float v) {
            if (col < m) {
                const float a = lds[oAc + t * LD + col] + hN * v;
                lds[oAn + t * LD + col] = a;
                lds[oVn + t * LD + col] = lds[oTHp + t * LD + col] * a;
            }
        }, 5, [&]() { if (i == lastLayer) shadow(1); });
        __syncthreads();
    }
    TL(c, 4);
    STAMP(c, 4);
    float p2acc[2] = {0.f, 0.f};                    // want_p2: this thread's share of sum g^2 in the two turns of the flat epilogue sweep
    // ---- closing: g = K0^T (tanh(o) . a) + A^T (A s) + c
    gemm_phase<S>(c, pl, rg, true, phClose, more_evals ? phOpen : phNone, oVs(lastLayer), LD, [&](int t, int i, float v) {
        if (i < D1) {
            // A^T z with a fixed trip count: the 2*ZQLD LDS reads are independent and issue back to back
            float aq[ZQLD], zq[ZQLD];
#pragma unroll
            for (int q = 0; q < ZQLD; ++q) { aq[q] = (q < r) ? Araw[q * D1 + i] : 0.f; zq[q] = lds[oZQ + t * ZQLD + q]; }
            float g = v + cw[i];
#pragma unroll
            for (int q = 0; q < ZQLD; ++q) g = fmaf(aq[q], zq[q], g);
            lds[oG + t * pl.GLD + i] = g;
            if (want_p2 && i < pl.d) p2acc[(t * pl.GLD + i) >= c.nthreads ? 1 : 0] += g * g;
        }
    }, 7, NoPost(), [&](int iw, int niw) { if (z_in_closing && !z_ready) z_from_s(pl, iw * 64 + c.lane, niw * 64); shadow(2); });
    if (want_p2) {
        // the flat split-K sweep gives every wave one (sample, column block) per turn: wave totals, fixed order
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int j = c.wave * 64 + kk * c.nthreads;
            if (j < T * pl.GLD) {
                const float v = sum64(p2acc[kk]);
                if (c.lane == 0) lds[pl.lPW + 2 * T * 4 + (j >> 6)] = v;
            }
        }
    }
    ring_ready = more_evals;
    __syncthreads();
    TL(c, 5);
    STAMP(c, 6);
}

// ------------------------------------------------------------------------------------------
// problem physics.  Cross2D.py:69-162, SwarmTraj.py:68-164, Quadcopter.py:65-113, utils.py:70-86.
// ------------------------------------------------------------------------------------------

// Interaction sum of one sample by cyclic pairing: agent a meets its partners (a+j) mod N, j = 1..(N-1)/2 (for
// even N the opposite agent j = N/2 as well, counted from the lower half only), so every unordered pair appears
// once, every agent has the same number of partners, x_a stays in registers and consecutive lanes read
// consecutive agents (stride PD floats: conflict-free).  The Gsz threads of the sample's group are used as
// P = Gsz / N parts of the partner range.  Returns this thread's partial sum.
template <int PD>
__device__ __forceinline__ float pair_sum_cyclic(const float* __restrict__ x, int N, int j0, int Gsz, float thr, float thr2, float den) {
    const int J = (N - 1) >> 1;                       // partners every agent meets
    const bool even = (N & 1) == 0;
    int lgP = 0;                                      // P = 2^lgP parts of the partner range (no integer division below)
    while ((2 * N << lgP) <= Gsz && (2 << lgP) <= J) ++lgP;
    const int P = 1 << lgP;
    float w = 0.f;
    for (int q = j0; q < N * P; q += Gsz) {
        int part = 0, a = q;
        while (a >= N) { a -= N; ++part; }
        float xa[PD];
#pragma unroll
        for (int k = 0; k < PD; ++k) xa[k] = x[PD * a + k];
        const int jlo = 1 + ((J * part) >> lgP), jhi = 1 + ((J * (part + 1)) >> lgP);   // [jlo, jhi)
        int b = a + jlo; if (b >= N) b -= N;
        for (int j = jlo; j < jhi; j += 4) {
            float s2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int bu = b + u; if (bu >= N) bu -= N;
                float a2 = 0.f;
#pragma unroll
                for (int k = 0; k < PD; ++k) { const float e = xa[k] - x[PD * bu + k]; a2 += e * e; }
                s2[u] = (j + u < jhi) ? a2 : 3.0e38f;
            }
            b += 4; if (b >= N) b -= N;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (s2[u] < thr2) {                                  // cheap reject; the exact test follows
                    const float dist = sqrtf(s2[u]);
                    if (dist < thr) {
                        const float e = expf(-(dist * dist) / den);
                        if (e != 1.f) w += e;                        // the reference drops entries equal to 1.
                    }
                }
            }
        }
        if (even && part == P - 1 && a < (N >> 1)) {                 // the opposite agent, once per pair
            float a2 = 0.f;
#pragma unroll
            for (int k = 0; k < PD; ++k) { const float e = xa[k] - x[PD * (a + (N >> 1)) + k]; a2 += e * e; }
            if (a2 < thr2) {
                const float dist = sqrtf(a2);
                if (dist < thr) { const float e = expf(-(dist * dist) / den); if (e != 1.f) w += e; }
            }
        }
    }
    return w;
}

// The x-only part of the running costs (obstacle sum, interaction sum) of sample t = 2*k + (wave >> 1), formed by
// waves 0..3 (two per sample) at the end of residual phase k (0: forward, 1: backward) of an 8-wave, 4-sample
// workgroup: those are the first waves of their SIMDs and finish streaming well before the second ones, so this runs
// while the phase is still waiting for its slowest wave.  Partials go to PW[t][wave & 1][0..1].
__device__ __forceinline__ void physics_x_shadow(const Ctx& c, const DevPlan& pl, const DevProb& pb, int k, int set = 0) {
    if (c.wave >= 4) return;
    const int N = pl.nAg;                           // (= pb.nAgents; a compile-time constant in the specialised kernels)
    const int t = 2 * k + (c.wave >> 1), half = c.wave & 1;
    const int j0 = half * 64 + c.lane;
    const float* x = lds + pl.lSB + t * pl.LDs;
    const bool wantW = want_W(pb);
    const float den = (float)(2.0 * pb.r * pb.r);
    const double fac = pb.training ? (pb.kind == NOCF_PROB_SWARMTRAJ ? 3.2 : 2.2) : 2.0;
    const float thr = (float)(fac * pb.r);
    const float thr2 = thr * thr * 1.000002f;
    float vq = 0.f, vw = 0.f;
    if (pb.kind == NOCF_PROB_CROSS2D) {
        if (pb.obstacle != NOCF_OBS_NONE)
            for (int a = j0; a < N; a += 128) vq += obstacle_cross2d(pb, x[2 * a], x[2 * a + 1]);
        if (wantW) vw = pair_sum_cyclic<2>(x, N, j0, 128, thr, thr2, den);
    } else {
        if (pb.obstacle != NOCF_OBS_NONE && pb.alphQ > 0.0)
            for (int a = j0; a < N; a += 128) vq += obstacle_swarm(pb, x[3 * a], x[3 * a + 1], x[3 * a + 2]);
        if (wantW) vw = pair_sum_cyclic<3>(x, N, j0, 128, thr, thr2, den);
    }
    vq = sum64(vq); vw = sum64(vw);
    if (c.lane == 0) { lds[pl.lPW + set * pl.T * 4 + t * 4 + half * 2] = vq; lds[pl.lPW + set * pl.T * 4 + t * 4 + half * 2 + 1] = vw; }
}

// Phase 1 (all threads): per-sample partial sums -> RED (ends with a barrier).
//   v0 = sum p^2, v1 = raw obstacle sum, v2 = raw interaction sum; quadcopter: sin/cos table.
// x_pre: the x-only terms (obstacle and interaction sums) were already formed in the shadow of the residual phases
// (physics_x_shadow); only sum p^2 is computed here and the partials are folded in.
__device__ void physics_sums(const Ctx& c, const DevPlan& pl, const DevProb& pb, bool x_pre = false) {
    float* Lm = lds;
    const float* X = Lm + pl.lSB;
    const float* P = Lm + pl.lG;
    float* TRIG = Lm + pl.lTRIG;
    const int T = pl.T, d = pl.d, N = pb.nAgents, ad = pb.agentDim;
    const int Gsz = c.nthreads / T;
    const int t = c.tid / Gsz, j0 = c.tid - t * Gsz;
    const float* x = X + t * pl.LDs;
    const float* p = P + t * pl.GLD;

    float v[3] = {0.f, 0.f, 0.f};
    TL(c, 6);
    for (int i = j0; i < d; i += Gsz) v[0] += p[i] * p[i];
    TL(c, 7);
    if (x_pre) {
        if (j0 == 0) { v[1] = Lm[pl.lPW + t * 4] + Lm[pl.lPW + t * 4 + 2]; v[2] = Lm[pl.lPW + t * 4 + 1] + Lm[pl.lPW + t * 4 + 3]; }
    } else if (pb.kind == NOCF_PROB_CROSS2D) {
        if (pb.obstacle != NOCF_OBS_NONE)
            for (int a = j0; a < N; a += Gsz) v[1] += obstacle_cross2d(pb, x[2 * a], x[2 * a + 1]);
    } else if (pb.kind == NOCF_PROB_SWARMTRAJ) {
        if (pb.obstacle != NOCF_OBS_NONE && pb.alphQ > 0.0)
            for (int a = j0; a < N; a += Gsz) v[1] += obstacle_swarm(pb, x[3 * a], x[3 * a + 1], x[3 * a + 2]);
    }
    TL(c, 14);
    if (!x_pre && want_W(pb) && N >= 2) {
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
            const float thr2 = thr * thr * 1.000002f;
            v[2] = (pd == 3) ? pair_sum_cyclic<3>(x, N, j0, Gsz, thr, thr2, den)
                             : pair_sum_cyclic<2>(x, N, j0, Gsz, thr, thr2, den);
        }
    }
    if (pb.kind == NOCF_PROB_QUADCOPTER && j0 < 3 * N) {       // sin/cos of (psi, theta, phi) per agent
        const int a = j0 / 3, q = j0 - 3 * a;
        float sn, cs;
        sincosf(x[12 * a + 3 + q], &sn, &cs);
        TRIG[(t * N + a) * 6 + q] = sn;
        TRIG[(t * N + a) * 6 + 3 + q] = cs;
    }
    TL(c, 40);
    group_reduce<3>(c, pl, v);
    TL(c, 41);
}

struct Costs { float L, H, Q, W; };

// Phase 2 (one thread per sample, after physics_sums): the scalars of calcLHQW.  For the quadcopter it
// also writes -grad_p H of its sample to DZ[s][0..d) and the thrusts to SC (for calcCtrls).
// calcLHQW of the point-agent problems from the three per-sample totals
__device__ __forceinline__ Costs costs_point(const DevProb& pb, float sp2, float Qraw, float Wv) {
    const bool wantW = want_W(pb);
    Costs o;
    float Lg;
    if (pb.kind == NOCF_PROB_CROSS2D) { o.Q = (float)pb.alphQ * Qraw; Lg = 0.5f * sp2 + o.Q; }
    else { o.Q = Qraw; Lg = 0.5f * sp2 + (float)pb.alphQ * Qraw; }
    if (wantW) Lg = Lg + (float)pb.alphW * Wv;
    o.L = Lg; o.H = -Lg + sp2; o.W = wantW ? Wv : 0.f;
    return o;
}

__device__ Costs physics_finish(const Ctx& c, const DevPlan& pl, const DevProb& pb, int s) {
    float* Lm = lds;
    const float* X = Lm + pl.lSB;
    const float* P = Lm + pl.lG;
    const int N = pb.nAgents;
    const bool wantW = want_W(pb);
    const float sp2 = group_total<3>(c, pl, s, 0), Qraw = group_total<3>(c, pl, s, 1);
    const float Wv = wantW ? group_total<3>(c, pl, s, 2) : 0.f;
    Costs o;
    if (pb.kind != NOCF_PROB_QUADCOPTER) {
        float Lg;
        if (pb.kind == NOCF_PROB_CROSS2D) { o.Q = (float)pb.alphQ * Qraw; Lg = 0.5f * sp2 + o.Q; }
        else { o.Q = Qraw; Lg = 0.5f * sp2 + (float)pb.alphQ * Qraw; }
        if (wantW) Lg = Lg + (float)pb.alphW * Wv;
        o.L = Lg; o.H = -Lg + sp2; o.W = Wv;
        return o;
    }
    const float* xs = X + s * pl.LDs;
    const float* ps = P + s * pl.GLD;
    float* dz = Lm + pl.lDZ + s * pl.ZLD;
    float* SC = Lm + pl.lSC;
    const float* TRIG = Lm + pl.lTRIG;
    const float mass = (float)pb.mass, grav = (float)pb.grav;
    o.Q = 0.f;                                                   // Quadcopter.py:116-122: no obstacle implemented
    float Lg = (float)pb.alphQ * o.Q;
    if (wantW) Lg = Lg + (float)pb.alphW * Wv;
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
        H = H - Lg - s1 - s2 - um * fsum + grav * pa[8] + 0.5f * sq;     // running L, like the reference loop
        float* da = dz + 12 * a;
        for (int k = 0; k < 6; ++k) da[k] = xa[6 + k];
        da[6] = um * f7; da[7] = um * f8; da[8] = um * f9 - grav;
        da[9] = -0.5f * pa[9]; da[10] = -0.5f * pa[10]; da[11] = -0.5f * pa[11];
        SC[s * N + a] = u;
    }
    o.L = Lg; o.H = H; o.W = Wv;
    return o;
}

// controls for the T samples from (x = SB, p = G) -> global rows; Cross2D.py:164-165,
// SwarmTraj.py:166-167, Quadcopter.py:165-174 (thrusts were left in SC by physics_finish).
__device__ void ctrl_write(const Ctx& c, const DevPlan& pl, const DevProb& pb, float* out, long row0, long n, int cdim) {
    const float* P = lds + pl.lG;
    const float* SC = lds + pl.lSC;
    for (int t = 0; t < pl.T; ++t) {
        if (row0 + t >= n) break;
        float* o = out + (row0 + t) * cdim;
        if (pb.kind != NOCF_PROB_QUADCOPTER) {
            for (int i = c.tid; i < pl.d; i += c.nthreads) o[i] = -P[t * pl.GLD + i];
        } else {
            for (int i = c.tid; i < cdim; i += c.nthreads) {
                const int a = i >> 2, q = i & 3;
                o[i] = (q == 0) ? SC[t * pb.nAgents + a] : -0.5f * P[t * pl.GLD + 12 * a + 8 + q];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// the rollout kernel: src/OCflow.py:7-95 for T samples per workgroup
// ------------------------------------------------------------------------------------------

template <int S>
__device__ __forceinline__ void rollout_body(const DevPlan& pl, const DevPlan* __restrict__ plp, const DevProb& pb,
                                             const float* __restrict__ ws, const RollArgs& ra) {
    Ctx c;
    ctx_init(c, ws, (unsigned)(pl.oPlan * 4));
    c.cb = plp->cb;
    const int T = pl.T, d = pl.d, ZLD = pl.ZLD;
    const long row0 = (long)blockIdx.x * T;
    float* SB = lds + pl.lSB;
    float* Z0 = lds + pl.lZ0;
    float* ZA = lds + pl.lZA;
    float* DZ = lds + pl.lDZ;
    const float* G = lds + pl.lG;

    for (int i = c.tid; i < pl.ldsFloats; i += c.nthreads) lds[i] = 0.f;
    __syncthreads();
    load_vectors(c, pl);
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
    Ring rg;
    bool ring_ready = false;
    bool z_next_ready = false;                      // ZQ already holds A s of the coming evaluation
    float costZ0 = 0.f, costZA = 0.f;               // point agents: cost integrals of lane (sample, component) of the finishing wave
    bool pend = false; float pend_hs = 0.f; int pend_st = 0;     // deferred cost side of the previous evaluation (see below)
    int evi = 0;                                                 // evaluation counter (parity selects the PW set)
    double tk = ra.t0;
    const int nstage = (ra.stepper == NOCF_RK4) ? 4 : 1;
    const int nsub = nstage + (ra.zFull ? 1 : 0);
    const bool quad = (pb.kind == NOCF_PROB_QUADCOPTER);
    // One call site for phi_eval: steps k < nt run the RK stages (plus, with intermediates, the
    // control evaluation); the extra pass k == nt is the terminal evaluation.
    for (int k = 0; k <= ra.nt; ++k) {
        const bool fin = (k == ra.nt);
        const double t1k = tk + ra.h;
        const double hsd = t1k - tk;                  // stepRK4 re-derives h = t1 - t0 (src/OCflow.py:170)
        const float hs = (float)hsd;
        if (fin) {
            if (c.tid < T) SB[c.tid * pl.LDs + d] = (float)ra.t1;      // src/OCflow.py:62
            __syncthreads();
        }
        for (int st = 0; st < (fin ? 1 : nsub); ++st) {
            if (ra.sAll && !fin && st < nstage) {           // record the stage input for the adjoint sweep
                float* dst = ra.sAll + ((long)(k * nstage + st) * ra.n) * (d + 1);
                for (int j = c.tid; j < T * (d + 1); j += c.nthreads) {
                    const int t = j / (d + 1), i = j - t * (d + 1);
                    if (row0 + t < ra.n) dst[(row0 + t) * (d + 1) + i] = SB[t * pl.LDs + i];
                }
            }
            // the ring is not carried across evaluations: keeping 64 registers alive through the physics cost spills
#ifdef NOCF_STAMPS
            c.tl = (ra.stamps && blockIdx.x == 7 && k == 40 && st == 1) ? ra.stamps + (long)gridDim.x * 12 : nullptr;
#endif
            // point agents, N > 2, 8 waves x 4 samples, one residual layer: the x-only cost terms are formed in the
            // shadow of the two long phases (physics_x_shadow)
            const bool xshape = !quad && pb.nAgents > 2 && pl.nwaves == 8 && pl.T == 4 && pl.nTh == 2;
            const bool xpre = !fin && xshape;
            // ... and, without intermediates, the cost side of an evaluation (L, |dPhi/dt - H|, Q, W and their RK
            // accumulation) is DEFERRED: nothing in it feeds the state, so evaluation e only leaves sum p^2 (from the
            // closing epilogue), the x-only partials and dPhi/dt behind, and 16 lanes of the last wave -- which has no
            // unit in the closing phase -- finish it during the closing phase of evaluation e+1, before that phase's
            // epilogue overwrites grad Phi.  The state update then follows the closing barrier directly.
            const bool deferred = xshape && pl.SK6 > 1 && pl.DB * pl.SK6 < pl.nwaves && T * pl.GLD <= 2 * c.nthreads && pl.DB <= 4 && !ra.zFull;
            phi_eval<S>(c, pl, fin, rg, ring_ready, false, z_next_ready,               // (one call site: see nocf_bwd.inc)
                        [&](int k2) {
                            if (k2 < 2) { if (xpre) physics_x_shadow(c, pl, pb, k2, deferred ? (evi & 1) : 0); return; }
                            if (!(deferred && pend && c.wave == pl.nwaves - 1 && c.lane < 4 * T)) return;
                            const int s = c.lane >> 2, q = c.lane & 3;
                            const float* P2 = lds + pl.lPW + 2 * T * 4;
                            float sp2 = P2[s * pl.DB];
                            for (int bk = 1; bk < pl.DB; ++bk) sp2 += P2[s * pl.DB + bk];
                            const float* PWp = lds + pl.lPW + ((evi & 1) ^ 1) * T * 4;        // the previous evaluation's set
                            const float Qraw = PWp[s * 4] + PWp[s * 4 + 2];
                            const float Wv = PWp[s * 4 + 1] + PWp[s * 4 + 3];
                            const Costs cs = costs_point(pb, sp2, Qraw, Wv);
                            const float val = (q == 0) ? cs.L : (q == 1) ? fabsf(G[s * pl.GLD + d] - cs.H) : (q == 2) ? cs.Q : cs.W;
                            const float K = pend_hs * val;
                            if (nstage == 1) { costZ0 = costZ0 + K; }
                            else if (pend_st == 0) { costZA = costZ0 + c16 * K; }
                            else if (pend_st == 1 || pend_st == 2) { costZA += c26 * K; }
                            else { costZ0 = costZA + c16 * K; }
                        }, deferred && !fin);
            z_next_ready = false;
            pend = false;
            ++evi;
            if (fin) break;
            if (deferred && st < nstage) {
                // ---- state update right behind the closing barrier (src/OCflow.py:143-184, x components), straight into SB
                double tnx;
                if (nstage == 1) tnx = t1k;
                else tnx = (st < 2) ? (tk + hsd / 2) : (st == 2 ? (tk + hsd) : t1k);
                for (int j = c.tid; j < T * d; j += c.nthreads) {
                    int t = 0, i = j;
                    while (i >= d) { i -= d; ++t; }
                    const float K = hs * -G[t * pl.GLD + i];
                    const float z0 = Z0[t * ZLD + i];
                    float xs;
                    if (nstage == 1) { xs = z0 + K; Z0[t * ZLD + i] = xs; }
                    else if (st == 0) { ZA[t * ZLD + i] = z0 + c16 * K; xs = z0 + 0.5f * K; }
                    else if (st == 1) { ZA[t * ZLD + i] += c26 * K; xs = z0 + 0.5f * K; }
                    else if (st == 2) { ZA[t * ZLD + i] += c26 * K; xs = z0 + K; }
                    else { xs = ZA[t * ZLD + i] + c16 * K; Z0[t * ZLD + i] = xs; }
                    SB[t * pl.LDs + i] = xs;
                }
                if (c.tid < T) SB[c.tid * pl.LDs + d] = (float)tnx;
                pend = true; pend_hs = hs; pend_st = st;
                TL(c, 43);
                __syncthreads();
                TL(c, 44);
                STAMP(c, 9);
                continue;
            }
            // ---- RK update (src/OCflow.py:143-184): z_next accumulates, SB receives the next stage state
            const bool stage = (st < nstage);
            const bool last = (st == nstage - 1);
            double tnext;
            if (nstage == 1) tnext = t1k;
            else tnext = (st < 2) ? (tk + hsd / 2) : (st == 2 ? (tk + hsd) : t1k);
            // with intermediates the next evaluation is the control at (z_{k+1}, (tk+h)-h), src/OCflow.py:53
            if (last && ra.zFull) tnext = t1k - ra.h;
            // point agents: the state moves with -p, independent of the cost terms, so the next state is formed
            // BEFORE the physics (into XN = DZ: the physics still reads the current x from SB) and the tail
            // below overlaps three jobs on different waves
            float* XN = DZ;
            auto rk = [&](int t, int i, float dzi) {
                const float K = hs * dzi;
                const float z0 = Z0[t * ZLD + i];
                float xs;
                if (nstage == 1) { xs = z0 + K; Z0[t * ZLD + i] = xs; }
                else if (st == 0) { ZA[t * ZLD + i] = z0 + c16 * K; xs = z0 + 0.5f * K; }
                else if (st == 1) { ZA[t * ZLD + i] += c26 * K; xs = z0 + 0.5f * K; }
                else if (st == 2) { ZA[t * ZLD + i] += c26 * K; xs = z0 + K; }
                else { xs = ZA[t * ZLD + i] + c16 * K; Z0[t * ZLD + i] = xs; }
                if (i < d) { if (quad) SB[t * pl.LDs + i] = xs; else XN[t * ZLD + i] = xs; }
                if (last && ra.zFull && row0 + t < ra.n)
                    ra.zFull[((long)(k + 1) * ra.n + row0 + t) * (d + 4) + i] = xs;
            };
            if (stage && !quad) {                           // dx = -grad_p H = -p, all T*d components in one flat sweep
                for (int j = c.tid; j < T * d; j += c.nthreads) {
                    int t = 0, i = j;
                    while (i >= d) { i -= d; ++t; }
                    rk(t, i, -G[t * pl.GLD + i]);
                }
            }
            physics_sums(c, pl, pb, xpre);                  // ends with a barrier
            STAMP(c, 8);
            if (stage) {
                if (!quad) {
                    TL(c, 42);
                    // tail, three jobs side by side: the last wave finishes the 4 cost components of every sample
                    // (one lane each); the other waves compute z = A s for the NEXT evaluation from XN and move XN
                    // into SB.  z is left to phi_eval when the next evaluation is not a plain stage.
                    const int fw = pl.nwaves - 1;
                    const bool split = pl.nwaves > 1;
                    const bool zpre = !(last && (k == ra.nt - 1 || ra.zFull)) && !(pl.DB * pl.SK6 < pl.nwaves && pl.SK6 > 1);
                    if (c.wave == fw && c.lane < 4 * T) {
                        const int s = c.lane >> 2, q = c.lane & 3;
                        const Costs cs = physics_finish(c, pl, pb, s);
                        const float val = (q == 0) ? cs.L : (q == 1) ? fabsf(G[s * pl.GLD + d] - cs.H) : (q == 2) ? cs.Q : cs.W;
                        // the running cost integrals of this lane's (sample, component) live in two registers for the
                        // whole rollout (costZ0 = z0 part, costZA = RK accumulator): no LDS read-modify-write chain
                        {
                            const float K = hs * val;
                            float xs;
                            if (nstage == 1) { xs = costZ0 + K; costZ0 = xs; }
                            else if (st == 0) { costZA = costZ0 + c16 * K; xs = costZ0 + 0.5f * K; }
                            else if (st == 1) { costZA += c26 * K; xs = costZ0 + 0.5f * K; }
                            else if (st == 2) { costZA += c26 * K; xs = costZ0 + K; }
                            else { xs = costZA + c16 * K; costZ0 = xs; }
                            if (last && ra.zFull && row0 + s < ra.n)
                                ra.zFull[((long)(k + 1) * ra.n + row0 + s) * (d + 4) + d + q] = xs;
                        }
                        if (q == 0) SB[s * pl.LDs + d] = (float)tnext;
                    }
                    if (!split || c.wave != fw) {
                        const int zt = split ? c.nthreads - 64 : c.nthreads;
                        if (zpre) {
                            const float* vecA = lds + pl.lVEC - pl.ob0 + pl.oA;
                            const int r = pl.r, D1 = pl.D1;
                            const float tn = (float)tnext;
                            const int items = T * r * 8;
                            for (int base = 0; base < items; base += zt) {
                                const int id = base + c.tid;
                                const int part = id & 7, tr = id >> 3;
                                float acc = 0.f;
                                if (id < items) {
                                    const int t = tr / r, rr = tr - t * r;
                                    const float* arow = vecA + (long)rr * D1;
                                    for (int i = part; i < D1; i += 8) acc += arow[i] * ((i < d) ? XN[t * ZLD + i] : tn);
                                }
                                acc = sum8(acc);
                                if (id < items && part == 0) { const int t = tr / r; lds[pl.lZQ + t * ZQLD + (tr - t * r)] = acc; }
                            }
                        }
                        for (int t = 0; t < T; ++t)
                            for (int i = c.tid; i < d; i += zt) SB[t * pl.LDs + i] = XN[t * ZLD + i];
                    }
                    z_next_ready = zpre;
                } else {
                    // quadcopter: one thread per sample forms the whole right-hand side (sequential agent loop of
                    // the reference), then every thread takes part in the RK update of the T*(d+4) components
                    if (c.tid < T) {
                        const int s = c.tid;
                        const Costs cs = physics_finish(c, pl, pb, s);
                        DZ[s * ZLD + d] = cs.L;
                        DZ[s * ZLD + d + 1] = fabsf(G[s * pl.GLD + d] - cs.H);
                        DZ[s * ZLD + d + 2] = cs.Q;
                        DZ[s * ZLD + d + 3] = cs.W;
                    }
                    __syncthreads();
                    for (int j = c.tid; j < T * (d + 4); j += c.nthreads) { const int t = j / (d + 4), i = j - t * (d + 4); rk(t, i, DZ[t * ZLD + i]); }
                    if (c.tid < T) SB[c.tid * pl.LDs + d] = (float)tnext;
                }
            } else {
                if (quad && c.tid < T) (void)physics_finish(c, pl, pb, c.tid);     // thrusts for calcCtrls
                if (quad) __syncthreads();
                ctrl_write(c, pl, pb, ra.ctrlFull + (long)(k + 1) * ra.n * ra.cdim, row0, ra.n, ra.cdim);
                if (c.tid < T) SB[c.tid * pl.LDs + d] = (float)t1k;
            }
            TL(c, 43);
            __syncthreads();
            TL(c, 44);
            STAMP(c, 9);
        }
        tk += ra.h;
    }
#if defined(NOCF_STAMPS) && NOCF_STAMPS >= 2
    if (ra.stamps && c.tid == 0)
        for (int i = 0; i < 12; ++i) ra.stamps[(long)blockIdx.x * 12 + i] = c.acc[i];
#endif

    if (!quad) {                                    // the cost integrals return from their registers
        if (c.wave == pl.nwaves - 1 && c.lane < 4 * T) Z0[(c.lane >> 2) * ZLD + d + (c.lane & 3)] = costZ0;
        __syncthreads();
    }
    // ---- terminal costs (src/OCflow.py:58-76)
    {
        const int Gsz = c.nthreads / T;
        const int t = c.tid / Gsz, j0 = c.tid - t * Gsz;
        const float* g = G + t * pl.GLD;
        float v[2] = {0.f, 0.f};
        for (int i = j0; i < d; i += Gsz) {
            const float res = Z0[t * ZLD + i] - pb.xtarget[i];
            v[0] += res * res;
            v[1] += fabsf(g[i] - ra.a0 * res);
        }
        group_reduce<2>(c, pl, v);
        if (c.tid < T && row0 + c.tid < ra.n) {
            const int s = c.tid;
            const long row = row0 + s;
            const float cG = 0.5f * group_total<2>(c, pl, s, 0);
            const float* z = Z0 + s * ZLD;
            if (ra.persample) {
                float* o = ra.persample + row * 7;
                o[0] = z[d]; o[1] = cG; o[2] = z[d + 1];
                o[3] = fabsf(lds[pl.lPHI + s] - ra.a0 * cG);
                o[4] = group_total<2>(c, pl, s, 1);
                o[5] = z[d + 2]; o[6] = z[d + 3];
            }
        }
        if (ra.z_out)
            for (int tt = 0; tt < T; ++tt)
                if (row0 + tt < ra.n)
                    for (int i = c.tid; i < d + 4; i += c.nthreads) ra.z_out[(row0 + tt) * (d + 4) + i] = Z0[tt * ZLD + i];
    }
}

// DynPlan: the plan is the record in the workspace (fields are scalar loads, not 60 pinned SGPRs).
// FixedPlan<...>: the plan is a compile-time constant of that shape.
template <int S, class SP>
__global__ void __launch_bounds__(NOCF_MAXTHREADS) rollout_kernel(const DevPlan* __restrict__ plp, DevProb pb, const float* __restrict__ ws, RollArgs ra) {
    if constexpr (SP::fixed) {
        constexpr DevPlan plc = SP::make();
        rollout_body<S>(plc, plp, pb, ws, ra);
    } else {
        rollout_body<S>(*plp, plp, pb, ws, ra);
    }
}

// ------------------------------------------------------------------------------------------
// stand-alone Phi.getGrad / Phi.forward and problem physics (same device code as the rollout)
// ------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(NOCF_MAXTHREADS) phi_kernel(const DevPlan* __restrict__ plp, const float* __restrict__ ws, const float* __restrict__ s,
                                                  long n, float* grad, float* value) {
    const DevPlan& pl = *plp;
    Ctx c;
    ctx_init(c, ws, (unsigned)(pl.oPlan * 4));
    c.cb = plp->cb;
    const int T = pl.T, D1 = pl.D1;
    const long row0 = (long)blockIdx.x * T;
    for (int i = c.tid; i < pl.ldsFloats; i += c.nthreads) lds[i] = 0.f;
    __syncthreads();
    load_vectors(c, pl);
    for (int t = 0; t < T; ++t) {
        long row = row0 + t; if (row >= n) row = n - 1;
        for (int i = c.tid; i < D1; i += c.nthreads) lds[pl.lSB + t * pl.LDs + i] = s[row * D1 + i];
    }
    Ring rg;
    bool ring_ready = false;
    __syncthreads();
    phi_eval<S>(c, pl, value != nullptr, rg, ring_ready, false);
    for (int t = 0; t < T; ++t) {
        if (row0 + t >= n) break;
        if (grad) for (int i = c.tid; i < D1; i += c.nthreads) grad[(row0 + t) * D1 + i] = lds[pl.lG + t * pl.GLD + i];
        if (value && c.tid == 0) value[row0 + t] = lds[pl.lPHI + t];
    }
}

__global__ void __launch_bounds__(512) prob_kernel(DevPlan pl, DevProb pb, const float* __restrict__ x,
                                                   const float* __restrict__ p, long n,
                                                   float* lhqw, float* gradpH, float* ctrls, int cdim) {
    Ctx c;
    ctx_init(c, x, 0);                              // no packed weights here; the descriptor is unused
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
    physics_sums(c, pl, pb);
    if (c.tid < T) {
        const Costs cs = physics_finish(c, pl, pb, c.tid);
        if (lhqw && row0 + c.tid < n) {
            float* o = lhqw + (row0 + c.tid) * 4;
            o[0] = cs.L; o[1] = cs.H; o[2] = cs.Q; o[3] = cs.W;
        }
    }
    __syncthreads();
    const bool quad = (pb.kind == NOCF_PROB_QUADCOPTER);
    if (gradpH)
        for (int t = 0; t < T; ++t) {
            if (row0 + t >= n) break;
            for (int i = c.tid; i < d; i += c.nthreads)     // grad_p H = p, or minus the quadcopter's state velocity
                gradpH[(row0 + t) * d + i] = quad ? -lds[pl.lDZ + t * pl.ZLD + i] : lds[pl.lG + t * pl.GLD + i];
        }
    if (ctrls) ctrl_write(c, pl, pb, ctrls, row0, n, cdim);
}

// deterministic reduction of the per-sample table: 7 column sums (fp64 accumulation, fixed order) + n
// (means != null: also the 7 means and Jc of cost_means_kernel, same arithmetic on the same rounded sums -- one launch less per call when no
// all-reduce stands between the sums and the means)
struct MeanArgs { float* means; float a0, a3, a4, a5; };
// (the body is shared with the lane kernel's last-workgroup reduction, nocf_lane.inc: same arithmetic, same order -> the same bits)
__device__ __forceinline__ void cost_sum_body(double* sh /* [256 * 7] LDS */, float* __restrict__ tab, long n, float* __restrict__ out,
                                              const unsigned* __restrict__ err, const MeanArgs& ma) {
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
    // a timed-out exchange in the split-role kernel must not pass as a result: poison the count (every mean is NaN) and every
    // per-sample row (the noMean path returns them); the host raises when it reads the error word (nocf_last_rollout_status_async)
    const bool failed = err && *err;
    if (threadIdx.x == 7) out[7] = failed ? __int_as_float(0x7fc00000) : (float)n;
    if (ma.means && threadIdx.x < 8) {
        const float cnt = failed ? __int_as_float(0x7fc00000) : (float)n;
        const int i = threadIdx.x;
        if (i < 7) ma.means[i] = (float)sh[i] / cnt;
        else {
            const float L = (float)sh[0] / cnt, G = (float)sh[1] / cnt, HJt = (float)sh[2] / cnt, HJfin = (float)sh[3] / cnt, HJgrad = (float)sh[4] / cnt;
            ma.means[7] = (((L + ma.a0 * G) + ma.a3 * HJt) + ma.a4 * HJfin) + ma.a5 * HJgrad;          // src/OCflow.py:88-90, same order as cost_means_kernel
        }
    }
    if (failed) for (long i = threadIdx.x; i < n * 7; i += 256) tab[i] = __int_as_float(0x7fc00000);
}
__global__ void __launch_bounds__(256) cost_sum_kernel(float* __restrict__ tab, long n, float* __restrict__ out,
                                                       const unsigned* __restrict__ err, MeanArgs ma = MeanArgs{nullptr, 0.f, 0.f, 0.f, 0.f}) {
    __shared__ double sh[256 * 7];
    cost_sum_body(sh, tab, n, out, err, ma);
}

// means of the 7 cost terms and Jc from the 8 sums, one launch instead of a dozen tiny elementwise ones
__global__ void cost_means_kernel(const float* __restrict__ sums, float a0, float a3, float a4, float a5, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < 7) out[i] = sums[i] / sums[7];
    if (i == 7) {
        const float n = sums[7];
        const float L = sums[0] / n, G = sums[1] / n, HJt = sums[2] / n, HJfin = sums[3] / n, HJgrad = sums[4] / n;
        out[7] = (((L + a0 * G) + a3 * HJt) + a4 * HJfin) + a5 * HJgrad;          // src/OCflow.py:88-90, same order
    }
}

// ------------------------------------------------------------------------------------------
// C[m x n] (+)= sum_k A[k][0..m) (x) B[k][0..n) for SMALL m, n (<= 64) and very many rows k: the weight-gradient
// contractions of the small networks (rows = samples x evaluations, 10^5..10^6).  A library GEMM with a 32x32 output
// runs on one workgroup there (0.4 ms each); this is HBM-bound work (read both row streams once).
// Stage 1: every workgroup owns a slice of the rows, stages 32 rows at a time in LDS, 16x16 threads keep a 4x4 tile
// of C in registers; stage 2 adds the per-workgroup partials in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------
#define CT_ROWS 32
// grid (row slices, 64x64 output tiles); tile (ti0, tj0) of C = columns [64 ti0, +64) of A against [64 tj0, +64) of B
__global__ void __launch_bounds__(256) contract_partial_kernel(const float* __restrict__ A, const float* __restrict__ B, long K, int m, int n,
                                                               long rows_per_block, int tiles_j, float* __restrict__ part) {
    __shared__ float sA[CT_ROWS][64 + 4], sB[CT_ROWS][64 + 4];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int rr0 = tid >> 6, cc = tid & 63;
    const int tile = blockIdx.y, i0 = (tile / tiles_j) * 64, j0 = (tile % tiles_j) * 64;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block < K) ? r0 + rows_per_block : K;
    for (long r = r0; r < r1; r += CT_ROWS) {
#pragma unroll
        for (int p = 0; p < CT_ROWS / 4; ++p) {
            const int rr = p * 4 + rr0;
            const long row = r + rr;
            sA[rr][cc] = (row < r1 && i0 + cc < m) ? A[row * m + i0 + cc] : 0.f;
            sB[rr][cc] = (row < r1 && j0 + cc < n) ? B[row * n + j0 + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < CT_ROWS; ++rr) {
            const float4 a = *reinterpret_cast<const float4*>(&sA[rr][4 * ti]);
            const float4 b = *reinterpret_cast<const float4*>(&sB[rr][4 * tj]);
            acc[0][0] = fmaf(a.x, b.x, acc[0][0]); acc[0][1] = fmaf(a.x, b.y, acc[0][1]); acc[0][2] = fmaf(a.x, b.z, acc[0][2]); acc[0][3] = fmaf(a.x, b.w, acc[0][3]);
            acc[1][0] = fmaf(a.y, b.x, acc[1][0]); acc[1][1] = fmaf(a.y, b.y, acc[1][1]); acc[1][2] = fmaf(a.y, b.z, acc[1][2]); acc[1][3] = fmaf(a.y, b.w, acc[1][3]);
            acc[2][0] = fmaf(a.z, b.x, acc[2][0]); acc[2][1] = fmaf(a.z, b.y, acc[2][1]); acc[2][2] = fmaf(a.z, b.z, acc[2][2]); acc[2][3] = fmaf(a.z, b.w, acc[2][3]);
            acc[3][0] = fmaf(a.w, b.x, acc[3][0]); acc[3][1] = fmaf(a.w, b.y, acc[3][1]); acc[3][2] = fmaf(a.w, b.z, acc[3][2]); acc[3][3] = fmaf(a.w, b.w, acc[3][3]);
        }
        __syncthreads();
    }
    float* o = part + ((long)blockIdx.x * gridDim.y + tile) * 4096;
#pragma unroll
    for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float4*>(&o[(4 * ti + a) * 64 + 4 * tj]) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
}

// stage 2: 16 outputs per workgroup, 16 threads per output each summing every 16th partial, then a fixed-order LDS sum
__global__ void __launch_bounds__(256) contract_reduce_kernel(const float* __restrict__ part, int nblocks, int m, int n, int tiles_j,
                                                              float* __restrict__ C, int accumulate) {
    __shared__ float sh[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + o;
    const int tile = blockIdx.y, ntiles = gridDim.y;
    float s = 0.f;
    for (int b = sl; b < nblocks; b += 16) s += part[((long)b * ntiles + tile) * 4096 + idx];
    sh[sl][o] = s;
    __syncthreads();
    if (sl == 0) {
        float t = sh[0][o];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += sh[q][o];
        const int i = (tile / tiles_j) * 64 + (idx >> 6), j = (tile % tiles_j) * 64 + (idx & 63);
        if (i < m && j < n) C[(long)i * n + j] = accumulate ? C[(long)i * n + j] + t : t;
    }
}

// column sums of a row stream X [K, n] (the bias / w / c gradients: sums over every sample and evaluation of the adjoint's rows):
// thread = column (consecutive threads on consecutive addresses), workgroup = a slice of rows, 8 rows in flight per thread;
// stage 2 adds the slices' partials in a fixed order
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ X, long K, int n, long rows_per_block, float* __restrict__ part) {
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col >= n) return;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block < K) ? r0 + rows_per_block : K;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long r = r0;
    for (; r + 8 <= r1; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += X[(r + u) * n + col];
    }
    for (; r < r1; ++r) a[0] += X[r * n + col];
    part[(long)blockIdx.x * n + col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
// 32 columns x 8 slabs of the partials per workgroup (one thread walking all ~1000 partials of a column was a 0.3 ms dependent chain,
// four times per training iteration); fixed order: 4 interleaved chains per slab, the slabs summed pairwise through LDS
__global__ void __launch_bounds__(256) colsum_reduce_kernel(const float* __restrict__ part, int nblocks, int n, float* __restrict__ out, int accumulate) {
    __shared__ float sh[8][32];
    const int cl = threadIdx.x & 31, gsl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (col < n) {
        const int per = (nblocks + 7) / 8;
        const int b0 = gsl * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += part[(long)(b + u) * n + col];
        }
        for (; b < b1; ++b) a[0] += part[(long)b * n + col];
        s = (a[0] + a[1]) + (a[2] + a[3]);
    }
    sh[gsl][cl] = s;
    __syncthreads();
    if (gsl == 0 && col < n) {
        const float t = ((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl])) + ((sh[4][cl] + sh[5][cl]) + (sh[6][cl] + sh[7][cl]));
        out[col] = accumulate ? out[col] + t : t;
    }
}

// turns a buffer into NaN when a rollout's error word is set (a timed-out exchange of a split-role kernel); returns at once otherwise
__global__ void poison_kernel(float* __restrict__ buf, long count, const unsigned* __restrict__ errp) {
    if (__builtin_nontemporal_load(errp) == 0u) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) buf[i] = __uint_as_float(0x7fc00000u);
}

__global__ void mfma_selftest_kernel(const float* __restrict__ a, const float* __restrict__ b, int K, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) acc = mfma4(a[(lane & 3) * K + k], b[k * 64 + lane], acc);
    for (int q = 0; q < 4; ++q) out[q * 64 + lane] = acc[q];
}

#include "nocf_mono.inc"
#include "nocf_f64.inc"
#include "nocf_f64_bwd.inc"
#include "nocf_lane.inc"
#include "nocf_bwd.inc"
#include "nocf_lane_bwd.inc"
#include "nocf_mono_bwd.inc"

// ------------------------------------------------------------------------------------------
// host side: plan construction and the C ABI
// ------------------------------------------------------------------------------------------
// The NOCF_* knobs are read from the environment ONCE (first use) and cached: a launch does no getenv.  Tests that change a knob between
// calls invalidate the cache with nocf_debug_reload_env() (neuraloc_amd/_lib.py does it when NOCF_ENV_WATCH=1, which tests/conftest.py sets).
#include <map>
#include <mutex>
#include <string>
static std::mutex g_env_mu;
static std::map<std::string, int> g_env_cache;
static std::map<std::string, int> g_env_override;          // nocf_set_knob: in-process overrides, not touched by a reload
int nocf_env_int(const char* name, int dflt) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    auto ov = g_env_override.find(name);
    if (ov != g_env_override.end()) return ov->second;
    auto it = g_env_cache.find(name);
    if (it != g_env_cache.end()) return it->second == INT32_MIN ? dflt : it->second;
    const char* v = getenv(name);
    const int val = (v && *v) ? atoi(v) : INT32_MIN;
    g_env_cache[name] = val;
    return val == INT32_MIN ? dflt : val;
}
static int env_int(const char* name, int dflt) { return nocf_env_int(name, dflt); }

// host wrapper: geometry knobs from the environment, then the (constexpr) layout
static int make_plan(int d, int m, int nTh, int r, int n_agents, DevPlan* out, int bwd = 0) {
#ifdef NOCF_STAMPS
    const int diagHalf = env_int("NOCF_DIAG_HALF", 0);    // diagnostic builds only: a timing experiment with wrong results
#else
    const int diagHalf = 0;
#endif
    return plan_layout(d, m, nTh, r, n_agents, bwd, env_int("NOCF_NWAVES", 0), env_int("NOCF_SUBTILES", 0), diagHalf, *out);
}

// Mono (one-CU weight-stationary) plan: returns 0 and fills *out when the shape has an instantiation, else an NOCF_E_* code.
#define MONO_SHAPES(X) X(8, 1) X(6, 1) X(4, 1) X(2, 1) X(8, 2) X(6, 2) X(4, 2)
static int make_mono_plan(const DevPlan& base, int n_agents, MonoPlan* out, bool bwd = false) {
    if (base.nTh != 2 || base.r > 16) return NOCF_E_SHAPE;
    const int KBD = cdiv(base.D1, 16);
    int KBM = cdiv(base.m, 16);
    if (KBM > 8) return NOCF_E_SHAPE;
    KBM = KBM <= 2 ? 2 : (KBM <= 4 ? 4 : (KBM <= 6 ? 6 : 8));   // hidden units are zero-padded up to an instantiated width (m <= 128)
    bool have = false;
#define NOCF_MONO_HAVE(M_, D_) if (KBM == M_ && KBD == D_) have = true;
    MONO_SHAPES(NOCF_MONO_HAVE)
#undef NOCF_MONO_HAVE
    if (!have) return NOCF_E_SHAPE;
    MonoPlan mp;
    memset(&mp, 0, sizeof(mp));
    mp.pp = base;
    mp.KBM = KBM; mp.KBD = KBD;
    DevPlan& pl = mp.pp;
    const int T = 16;
    pl.T = T; pl.nwaves = 4;
    int l = 0;
    auto take = [&](int nfl) { int o = l; l += rup(nfl, 4); return o; };
    mp.lSF = take(KBD * 256);
    mp.lUF = take(KBM * 256);
    mp.lVF = take(KBM * 256);
    mp.lK4 = take(KBD * KBM * 256);
    mp.lAT = take(KBD * 256);
    mp.lVO = take(3 * KBM * 16);
    mp.lCW = take(KBD * 16);
    mp.lZP = take(4 * 256);
    mp.lGP = take(KBD * 4 * 256);
    mp.lPHIP = take(64);
    pl.lSB = take(T * base.LDs);
    pl.lG = take(T * base.GLD);
    pl.lZQ = take(T * ZQLD);
    pl.lZ0 = take(T * base.ZLD); pl.lZA = take(T * base.ZLD); pl.lDZ = take(T * base.ZLD);
    pl.lRED = take(std::max(T, 4) * 4);
    pl.lSC = take(T * std::max(1, n_agents) + 8);
    pl.lPHI = take(T);
    pl.lTRIG = take(T * std::max(1, n_agents) * 6);
    pl.lPT = take(4);
    pl.lPW = take(4);
    // the kernels' LDS layouts are compile-time constants (nocf_mono.inc, nocf_mono_bwd.inc): the shape has to fit their strides and agent counts
    if (KBD > 2 || base.LDs != MONO_LDS_S || base.GLD != MONO_GLD || base.ZLD > MONO_ZLD(KBD) || n_agents > MONO_NAG(KBD)) return NOCF_E_SHAPE;
    l = bwd ? mono_bwd_lds(KBM, KBD).total : mono_fwd_lds(KBM, KBD).total;
    pl.ldsFloats = l;
    if ((size_t)l * 4 > (bwd ? 160 : 96) * 1024) return NOCF_E_LDS;     // (two input k-blocks with 128 hidden units: 70 KB; one workgroup per CU either way)
    long o = base.oPlan + (long)rup((int)(sizeof(MonoPlan) / 4), 64);         // floats
    const long nW = (long)KBM * KBM * 64, nK1 = (long)KBM * KBD * 64, nK4 = (long)KBD * KBM * 64, nA = (long)KBD * 64;   // float4s
    mp.oW2 = o / 4; o += nW * 4;
    mp.oW3 = o / 4; o += nW * 4;
    mp.oK1 = o / 4; o += nK1 * 4;
    mp.oK4 = o / 4; o += nK4 * 4;
    mp.oAZ = o / 4; o += nA * 4;
    mp.oAT = o / 4; o += nA * 4;
    pl.pad_ = (int)(o - base.oPlan);                           // floats behind the plan record this kernel needs
    *out = mp;
    return 0;
}

static size_t mono_ws_bytes(const MonoPlan& mp) {
    return (size_t)(mp.pp.oPlan + mp.pp.pad_) * sizeof(float);
}

static size_t plan_ws_bytes(const DevPlan& pl) {
    return (size_t)pl.oPlan * sizeof(float) + ((sizeof(DevPlan) + 15) / 16) * 16;
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
    DevPhi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev};
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
// Per-DEVICE call state (the error word of the last rollout, which kernel it was, the timing events of a profile window), keyed by the
// current HIP device of the calling thread: two host threads driving two GPUs never see each other's state, and the thread PyTorch's
// autograd engine runs a backward on shares the state of the thread that launched the forward on that device.  (Two threads driving
// ONE device concurrently are not supported: the weight-stationary kernels need the whole device.)  There is no other mutable state
// in the library besides the cached knobs, which are read under a mutex.
struct CallState {
    const unsigned* errp = nullptr;     // device address of the last rollout's error word (split-role kernels), or null
    const char* kernel = "none";        // measurement hook: which rollout kernel the last call launched (nocf_last_rollout_kernel)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    bool prof_on = false;
};
static CallState g_call_state[64];
static CallState& call_state() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return g_call_state[dev & 63];
}
#define g_last_errp (call_state().errp)
#define g_last_kernel (call_state().kernel)
#define g_prof_events (call_state().events)
#define g_prof_on (call_state().prof_on)
static unsigned long long* g_stamp_buf = nullptr;     // diagnostic builds only (nocf_debug_set_stamp_buffer)

extern "C" {

int nocf_version(void) { return NOCF_VERSION; }

const char* nocf_last_rollout_kernel(void) { return g_last_kernel; }

void nocf_debug_reload_env(void) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    g_env_cache.clear();
}

int nocf_set_knob(const char* name, int32_t value, int32_t clear) {
    if (!name || strncmp(name, "NOCF_", 5) != 0) return NOCF_E_NULL;
    std::lock_guard<std::mutex> lk(g_env_mu);
    if (clear) g_env_override.erase(name); else g_env_override[name] = value;
    return 0;
}

int nocf_last_rollout_status_async(uint32_t* host_word, void* stream) {
    if (!host_word) return NOCF_E_NULL;
    if (!g_last_errp) { *host_word = 0u; return 0; }
    const hipError_t e = hipMemcpyAsync(host_word, g_last_errp, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    return e ? (int)e : 1;
}

int nocf_debug_set_stamp_buffer(void* device_buf) {
#ifdef NOCF_STAMPS
    g_stamp_buf = (unsigned long long*)device_buf;
    return 0;
#else
    (void)device_buf;
    return NOCF_E_SHAPE;                                 // production build carries no stamps
#endif
}

int nocf_cost_means_f32(const float* cost_sums, const float* alph, float* out, void* stream) {
    if (!cost_sums || !alph || !out) return NOCF_E_NULL;
    hipLaunchKernelGGL(cost_means_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cost_sums, alph[0], alph[3], alph[4], alph[5], out);
    return (int)hipGetLastError();
}

int64_t nocf_small_grad_floats(int32_t d, int32_t m) {
    const long D1 = d + 1;
    return (long)m * D1 + m + (long)m * m + m + m + D1 + 1 + D1 * D1;
}

int nocf_rollout_bwd_small_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                               const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                               float* gpart, float* lam0, void* stream) {
    int rc = check_phi(phi);
    if (rc) return rc;
    if (!alph || !s_all || !z_final || !hs || !gpart) return NOCF_E_NULL;
    if (n < 1 || nt < 1) return NOCF_E_SHAPE;
    if (stepper != NOCF_RK4 && stepper != NOCF_RK1) return NOCF_E_STEPPER;
    DevProb pb;
    rc = fill_prob(prob, phi->d, &pb);
    if (rc) return rc;
    // eligibility = the forward lane kernel's, Cross2D agents only
    if (!(phi->nTh == 2 && phi->m <= 32 && phi->d + 1 <= 32 && pb.kind == NOCF_PROB_CROSS2D && pb.agentDim == 2 &&
          pb.nAgents * 2 == phi->d && env_int("NOCF_LANE", 1) != 0))
        return NOCF_E_SHAPE;
    LaneBwdArgs la;
    la.P = DevPhi{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev};
    la.d = phi->d; la.m = phi->m; la.r = phi->r; la.nAg = pb.nAgents; la.cb = phi->cb;
    la.sAll = s_all; la.zT = z_final; la.hs = hs; la.n = n; la.nt = nt; la.nstage = (stepper == NOCF_RK4) ? 4 : 1;
    la.t1 = (float)t1; la.a0 = alph[0]; la.a3 = alph[3]; la.a4 = alph[4]; la.a5 = alph[5]; la.inv_n = (float)inv_n;
    la.gpart = gpart; la.P_total = nocf_small_grad_floats(phi->d, phi->m); la.lam0 = lam0;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((n + 3) / 4);
    const int MPsel = phi->m <= 16 ? 16 : 32;
    const int DPsel = phi->d + 1 <= 8 ? 8 : (phi->d + 1 <= 16 ? 16 : 32);
    if (env_int("NOCF_DEBUG", 0)) fprintf(stderr, "[nocf] lane adjoint kernel\n");
#define NOCF_LANEB_LAUNCH(MPV, DPV) hipLaunchKernelGGL((rollout_lane_bwd_kernel<MPV, DPV>), dim3(grid), dim3(256), 0, st, la, pb)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_prof_on) {
        if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
        (void)hipEventRecord(ev0, st);
    }
    if (MPsel == 16) { if (DPsel == 8) NOCF_LANEB_LAUNCH(16, 8); else if (DPsel == 16) NOCF_LANEB_LAUNCH(16, 16); else NOCF_LANEB_LAUNCH(16, 32); }
    else             { if (DPsel == 8) NOCF_LANEB_LAUNCH(32, 8); else if (DPsel == 16) NOCF_LANEB_LAUNCH(32, 16); else NOCF_LANEB_LAUNCH(32, 32); }
#undef NOCF_LANEB_LAUNCH
    if (g_prof_on) { (void)hipEventRecord(ev1, st); g_prof_events.emplace_back(ev0, ev1); }
    g_last_kernel = "rollout_lane_bwd_kernel";
    g_last_errp = nullptr;
    return (int)hipGetLastError();
}

int nocf_contract_f32(const float* A, const float* B, int64_t K, int32_t m, int32_t n, float* C, int32_t accumulate,
                      float* scratch, size_t scratch_floats, void* stream) {
    if (!A || !B || !C || !scratch) return NOCF_E_NULL;
    if (K < 1 || m < 1 || n < 1 || m > 512 || n > 512) return NOCF_E_SHAPE;
    const int tiles_i = (m + 63) / 64, tiles_j = (n + 63) / 64, ntiles = tiles_i * tiles_j;
    if (ntiles > 64) return NOCF_E_SHAPE;
    long nblocks = (K + 255) / 256;                       // >= 256 rows per workgroup
    if (nblocks > 1024 / ntiles) nblocks = 1024 / ntiles;
    if ((size_t)nblocks * ntiles * 4096 > scratch_floats) nblocks = (long)(scratch_floats / ((size_t)ntiles * 4096));
    if (nblocks < 1) return NOCF_E_WORKSPACE;
    long rpb = (K + nblocks - 1) / nblocks;
    rpb = (rpb + CT_ROWS - 1) / CT_ROWS * CT_ROWS;
    nblocks = (K + rpb - 1) / rpb;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(contract_partial_kernel, dim3((int)nblocks, ntiles), dim3(256), 0, st, A, B, (long)K, m, n, rpb, tiles_j, scratch);
    hipLaunchKernelGGL(contract_reduce_kernel, dim3(256, ntiles), dim3(256), 0, st, scratch, (int)nblocks, m, n, tiles_j, C, accumulate);
    return (int)hipGetLastError();
}

int nocf_colsum_f32(const float* X, int64_t K, int32_t n, float* out, int32_t accumulate, float* scratch, size_t scratch_floats, void* stream) {
    if (!X || !out || !scratch) return NOCF_E_NULL;
    if (K < 1 || n < 1) return NOCF_E_SHAPE;
    const int cb = (n + 255) / 256;
    long nblocks = (K + 127) / 128;                       // >= 128 rows per workgroup
    if (nblocks > 2048 / cb) nblocks = 2048 / cb;
    if ((size_t)nblocks * n > scratch_floats) nblocks = (long)(scratch_floats / (size_t)n);
    if (nblocks < 1) return NOCF_E_WORKSPACE;
    const long rpb = (K + nblocks - 1) / nblocks;
    nblocks = (K + rpb - 1) / rpb;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((int)nblocks, cb), dim3(256), 0, st, X, (long)K, n, rpb, scratch);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, st, scratch, (int)nblocks, n, out, accumulate);
    return (int)hipGetLastError();
}

int nocf_debug_set_timeline_buffer(void* device_buf) {
#ifdef NOCF_STAMPS
    unsigned long long* p = (unsigned long long*)device_buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_dev), &p, sizeof(p));
#else
    (void)device_buf;
    return NOCF_E_SHAPE;                                 // production build carries no stamps
#endif
}

int nocf_profile_begin(void) {
    for (auto& pr : g_prof_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
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
    for (auto& pr : g_prof_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    g_prof_events.clear();
    return 0;
}

size_t nocf_workspace_bytes(int32_t d, int32_t m, int32_t nTh) {
    DevPlan pl;
    const int r = std::min(10, d + 1);
    if (make_plan(d, m, nTh, r, 1, &pl) != 0) return 0;
    return plan_ws_bytes(pl);
}

size_t nocf_rollout_workspace_bytes(int32_t d, int32_t m, int32_t nTh, int64_t n) {
    DevPlan pl;
    const int r = std::min(10, d + 1);
    if (make_plan(d, m, nTh, r, 1, &pl) != 0) return 0;
    size_t b = plan_ws_bytes(pl);
    MonoPlan mpl;
    if (make_mono_plan(pl, 1, &mpl) == 0) b = std::max(b, mono_ws_bytes(mpl));
#ifndef NOCF_JIT_ONLY
    size_t db = 0;
    if (n > 0 && duo_workspace_bytes(d, m, nTh, r, 1, n, &db) == 0) b = std::max(b, db);
#endif
    return b;
}

int nocf_ctrl_dim(const NocfProb* prob, int32_t d) {
    if (!prob) return NOCF_E_NULL;
    return prob->kind == NOCF_PROB_QUADCOPTER ? 4 * prob->n_agents : d;
}

int nocf_segments_supported(const NocfPhi* phi, const NocfProb* prob) {
    if (check_phi(phi) || !prob) return 0;
    DevProb pb;
    if (fill_prob(prob, phi->d, &pb)) return 0;
    DevPlan pl;
    if (make_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, &pl)) return 0;
    MonoPlan mpl;
    return env_int("NOCF_MONO", 1) != 0 && make_mono_plan(pl, pb.nAgents, &mpl) == 0 ? 1 : 0;
}

// Ticket words of the one-launch small rollouts (nocf_lane.inc: the last workgroup to finish reduces the per-sample table).  256 B per device,
// allocated and zeroed once; a word is self-resetting (atomicInc wraps at the grid size), and every (device, stream) pair gets its own word --
// launches on one stream are ordered, launches on different streams never share a word.  More than 64 streams: the two-launch path.
static unsigned* lane_ticket(hipStream_t st) {
    static std::mutex mu;
    static std::map<int, unsigned*> base;
    static std::map<std::pair<int, hipStream_t>, int> slot;
    static std::map<int, int> used;
    int dev = 0;
    if (hipGetDevice(&dev)) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!base.count(dev)) {
        unsigned* p = nullptr;
        if (hipMalloc((void**)&p, 64 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); base[dev] = nullptr; }
        else if (hipMemset(p, 0, 64 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); base[dev] = nullptr; }
        else base[dev] = p;
    }
    unsigned* b = base[dev];
    if (!b) return nullptr;
    const auto key = std::make_pair(dev, st);
    auto it = slot.find(key);
    if (it == slot.end()) {
        if (used[dev] >= 64) return nullptr;
        it = slot.emplace(key, used[dev]++).first;
    }
    return b + it->second;
}

static int rollout_impl(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                        double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                        float* z_out, float* persample, float* cost_sums, float* zFull, float* ctrlFull,
                        void* workspace, size_t workspace_bytes, void* stream, float* s_all,
                        float* act = nullptr, int32_t* act_recorded = nullptr, float* tapeU1 = nullptr, float* tapeSc = nullptr,
                        const SegTab* seg = nullptr, float* cost_means = nullptr) {
    if (act_recorded) *act_recorded = 0;
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
    RollArgs ra;
    ra.x = x; ra.n = n; ra.t0 = t0; ra.t1 = t1; ra.h = (t1 - t0) / nt; ra.nt = nt; ra.stepper = stepper;
    ra.a0 = alph[0];
    ra.z_out = z_out; ra.persample = persample; ra.zFull = zFull; ra.ctrlFull = ctrlFull;
    ra.cdim = nocf_ctrl_dim(prob, phi->d);
    ra.stamps = g_stamp_buf;
    ra.sAll = s_all;
    const MeanArgs mean_args{cost_sums ? cost_means : nullptr, alph[0], alph[3], alph[4], alph[5]};
    memset(&ra.seg, 0, sizeof(ra.seg));
    if (seg) ra.seg = *seg;                       // (several time segments in one launch: the one-CU kernel only, NOCF_E_SHAPE otherwise)
    ra.act = nullptr; ra.actRows = 0;             // (only the split-role kernel records activations: set below)
    ra.tapeU1 = nullptr; ra.tapeSc = nullptr;
    hipError_t e;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const unsigned* errp = nullptr;
    if (seg) {                                    // segments run on the one-CU kernel only: decided before anything is enqueued or reset
        MonoPlan probe;
        if (env_int("NOCF_MONO", 1) == 0 || make_mono_plan(pl, pb.nAgents, &probe) != 0 || workspace_bytes < mono_ws_bytes(probe))
            return NOCF_E_SHAPE;
    }
    g_last_errp = nullptr;
    // small networks: one wave per sample, everything in registers (nocf_lane.inc); needs no packed images
    const bool lane_ok = env_int("NOCF_LANE", 1) != 0 && phi->nTh == 2 && phi->m <= 32 && phi->d + 1 <= 32 &&
                         pb.kind != NOCF_PROB_QUADCOPTER && pb.nAgents <= 16 && !g_stamp_buf && !seg;
    if (lane_ok) {
        LaneArgs la;
        la.P = DevPhi{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev};
        la.d = phi->d; la.m = phi->m; la.r = phi->r; la.nAg = pb.nAgents; la.cb = phi->cb;
        // one launch per call (round 6, NOCF_LANE_ONE=1): with a ticket word the kernel's last workgroup forms the sums (and means) itself.
        // OFF by default -- measured on the MI355X (profiles/r6/07_lane_one_launch.txt): the agent-scope release every workgroup needs in front of
        // its ticket is an L2 write-back (buffer_wbl2 sc1), 256-512 of them cost 7-15 us, more than the 4-us kernel and launch gap they replace
        unsigned* ticket = (cost_sums && !s_all && env_int("NOCF_LANE_ONE", 0) != 0) ? lane_ticket(st) : nullptr;
        la.sums = ticket ? cost_sums : nullptr; la.ticket = ticket;
        la.means = mean_args.means; la.a0 = mean_args.a0; la.a3 = mean_args.a3; la.a4 = mean_args.a4; la.a5 = mean_args.a5;
        const int grid = (int)((n + 3) / 4);
        const int MPsel = phi->m <= 16 ? 16 : 32;
        const int DPsel = phi->d + 1 <= 8 ? 8 : (phi->d + 1 <= 16 ? 16 : 32);
        if (g_prof_on) {
            if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
            (void)hipEventRecord(ev0, st);
        }
        const size_t laneLds = ticket ? (size_t)(256 * 7 * 8 + 16) : 0;      // (the last-workgroup reduction's scratch: only with a ticket)
#define NOCF_LANE_LAUNCH(MPV, DPV) do { if (s_all) hipLaunchKernelGGL((rollout_lane_kernel<MPV, DPV, true>), dim3(grid), dim3(256), 0, st, la, pb, ra); \
                                         else if (ticket) hipLaunchKernelGGL((rollout_lane_kernel<MPV, DPV, false, true>), dim3(grid), dim3(256), laneLds, st, la, pb, ra); \
                                         else hipLaunchKernelGGL((rollout_lane_kernel<MPV, DPV, false>), dim3(grid), dim3(256), 0, st, la, pb, ra); } while (0)
        if (MPsel == 16) { if (DPsel == 8) { NOCF_LANE_LAUNCH(16, 8); } else if (DPsel == 16) { NOCF_LANE_LAUNCH(16, 16); } else { NOCF_LANE_LAUNCH(16, 32); } }
        else             { if (DPsel == 8) { NOCF_LANE_LAUNCH(32, 8); } else if (DPsel == 16) { NOCF_LANE_LAUNCH(32, 16); } else { NOCF_LANE_LAUNCH(32, 32); } }
#undef NOCF_LANE_LAUNCH
        e = hipGetLastError();
        if (e) return (int)e;
        g_last_kernel = "rollout_lane_kernel";
        if (g_prof_on) { (void)hipEventRecord(ev1, st); g_prof_events.emplace_back(ev0, ev1); }
        if (cost_sums && !ticket) {
            hipLaunchKernelGGL(cost_sum_kernel, dim3(1), dim3(256), 0, st, persample, (long)n, cost_sums, errp, mean_args);
            e = hipGetLastError();
            if (e) return (int)e;
        }
        return 0;
    }
#ifndef NOCF_JIT_ONLY
    // split-role weight-stationary kernel (nocf_duo.hip): wide two-layer networks (m = 512) on point-agent problems, any batch
    // size (chunks of 2048 rows), evaluation and the recording forward of training
    if (env_int("NOCF_DUO", 1) != 0 && !seg) {
        if (g_prof_on) {
            if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
        }
        if (act && s_all) { ra.act = act; ra.actRows = (long)nt * ((stepper == NOCF_RK4) ? 4 : 1) * n; }
        if (act && s_all && tapeSc) { ra.actRows += n; ra.tapeU1 = tapeU1; ra.tapeSc = tapeSc; }       // tape: the terminal block too
        rc = duo_launch(phi, pb, ra, ws, workspace_bytes, st, &errp, env_int("NOCF_DEBUG", 0), g_prof_on ? ev0 : nullptr, g_prof_on ? ev1 : nullptr);
        ra.act = nullptr; ra.actRows = 0; ra.tapeU1 = nullptr; ra.tapeSc = nullptr;
        if (rc == 0) {
            if (act && s_all && act_recorded) *act_recorded = 1;
            g_last_kernel = "rollout_duo_kernel";
            g_last_errp = errp;
            if (g_prof_on) g_prof_events.emplace_back(ev0, ev1);
            if (cost_sums) {
                hipLaunchKernelGGL(cost_sum_kernel, dim3(1), dim3(256), 0, st, persample, (long)n, cost_sums, errp, mean_args);
                e = hipGetLastError();
                if (e) return (int)e;
            }
            // a timed-out exchange: every other output the caller asked for becomes NaN too (one tiny launch each; they return at once
            // when the error word is clear)
            if (z_out) hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(256), 0, st, z_out, (long)n * (phi->d + 4), errp);
            if (zFull) hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(256), 0, st, zFull, (long)(nt + 1) * n * (phi->d + 4), errp);
            if (ctrlFull) hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(256), 0, st, ctrlFull, (long)(nt + 1) * n * ra.cdim, errp);
            return 0;
        }
        if (g_prof_on) { (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); ev0 = ev1 = nullptr; }
        if (rc != 1) return rc;
    }
#endif
    rc = pack_weights(pl, phi, ws, st);
    if (rc) return rc;
    const DevPlan* plp = reinterpret_cast<const DevPlan*>(ws + pl.oPlan);
    // one-CU weight-stationary kernel (nocf_mono.inc): two-layer networks of up to 128 hidden units whose shape has an
    // instantiation (singlequad): no weight stream, no inter-workgroup traffic; also with intermediates
    MonoPlan mpl;
    const bool use_mono = (!s_all || env_int("NOCF_MONO_REC", 1) != 0) && env_int("NOCF_MONO", 1) != 0 &&
                          make_mono_plan(pl, pb.nAgents, &mpl) == 0 && workspace_bytes >= mono_ws_bytes(mpl);
    if (use_mono) {
        mpl.pp.cb = phi->cb;
        DevPhi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev};
        hipLaunchKernelGGL(mono_pack_kernel, dim3(64), dim3(256), 0, st, mpl, P, ws);
        const size_t ldsBytes = (size_t)mpl.pp.ldsFloats * 4;
        const void* fk = nullptr;
#define NOCF_MONO_PICK(M_, D_) if (mpl.KBM == M_ && mpl.KBD == D_) fk = s_all ? reinterpret_cast<const void*>(rollout_mono_kernel<M_, D_, true>) : reinterpret_cast<const void*>(rollout_mono_kernel<M_, D_, false>);
        MONO_SHAPES(NOCF_MONO_PICK)
#undef NOCF_MONO_PICK
        e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes); if (e) return (int)e;
        if (env_int("NOCF_DEBUG", 0)) fprintf(stderr, "[nocf] mono kernel: %d hidden / %d input k-blocks, LDS %zu B/workgroup\n", mpl.KBM, mpl.KBD, ldsBytes);
        if (g_prof_on) {
            if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
            (void)hipEventRecord(ev0, st);
        }
        const MonoPlan* mpp = reinterpret_cast<const MonoPlan*>(ws + mpl.pp.oPlan);
        const bool mono_rec = act && s_all && !tapeSc && (phi->m % 16) == 0;   // activation record: the one-CU kernel writes it too
        if (mono_rec) { ra.act = act; ra.actRows = (long)nt * ((stepper == NOCF_RK4) ? 4 : 1) * n; }
        void* args[] = {(void*)&mpp, (void*)&pb, (void*)&ws, (void*)&ra};
        e = hipLaunchKernel(fk, dim3((int)((n + 15) / 16)), dim3(256), args, ldsBytes, st); if (e) return (int)e;
        ra.act = nullptr; ra.actRows = 0;
        if (mono_rec && act_recorded) *act_recorded = 1;
        g_last_kernel = "rollout_mono_kernel";
        e = hipGetLastError();
        if (e) return (int)e;
        if (g_prof_on) { (void)hipEventRecord(ev1, st); g_prof_events.emplace_back(ev0, ev1); }
        if (cost_sums && seg) {                       // one row of 8 sums per segment
            for (int k = 0; k < seg->n; ++k) {
                const long r0 = (long)k * seg->rows, rn = std::min<long>(seg->rows, (long)n - r0);
                if (rn <= 0) break;
                hipLaunchKernelGGL(cost_sum_kernel, dim3(1), dim3(256), 0, st, persample + r0 * 7, rn, cost_sums + 8 * k, errp);
            }
            e = hipGetLastError();
            if (e) return (int)e;
        } else if (cost_sums) {
            hipLaunchKernelGGL(cost_sum_kernel, dim3(1), dim3(256), 0, st, persample, (long)n, cost_sums, errp, mean_args);
            e = hipGetLastError();
            if (e) return (int)e;
        }
        return 0;
    }
    if (seg) return NOCF_E_SHAPE;                     // (segments: no one-CU instantiation for this shape -- the caller launches them one by one)
    {
        const size_t ldsBytes = (size_t)pl.ldsFloats * 4;
        const int grid = (int)((n + pl.T - 1) / pl.T);
        const int block = pl.nwaves * 64;
        if (g_prof_on) {
            if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
            (void)hipEventRecord(ev0, st);
        }
        bool launched = false;
        if (pl.T == 4 && env_int("NOCF_FIXED", 1)) {
            // shape-specialised instantiations (FIXED_SHAPES): taken only when the run-time plan equals the
            // compile-time one bit for bit, so the environment knobs and odd shapes always get the generic kernel
            const void* fk = nullptr;
#define NOCF_TRY_FIXED(D, M, NTH, R, NAG) \
            if (!fk && plan_is<FixedPlan<D, M, NTH, R, NAG, 0>>(pl)) fk = reinterpret_cast<const void*>(rollout_kernel<1, FixedPlan<D, M, NTH, R, NAG, 0>>);
            FIXED_SHAPES(NOCF_TRY_FIXED)
            FIXED_SHAPES_EXTRA(NOCF_TRY_FIXED)
            if (s_all) { FIXED_SHAPES_TRAIN(NOCF_TRY_FIXED) }
#undef NOCF_TRY_FIXED
            if (fk) {
                e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes); if (e) return (int)e;
                void* args[] = {(void*)&plp, (void*)&pb, (void*)&ws, (void*)&ra};
                e = hipLaunchKernel(fk, dim3(grid), dim3(block), args, ldsBytes, st); if (e) return (int)e;
                launched = true;
                g_last_kernel = "rollout_kernel<shape-specialised>";
                if (env_int("NOCF_DEBUG", 0)) fprintf(stderr, "[nocf] shape-specialised rollout kernel\n");
            }
        }
        if (!launched) g_last_kernel = "rollout_kernel<generic>";
        if (!launched) switch (pl.T / 4) {
            case 1: e = set_lds(rollout_kernel<1, DynPlan>, ldsBytes); if (e) return (int)e;
                    hipLaunchKernelGGL((rollout_kernel<1, DynPlan>), dim3(grid), dim3(block), ldsBytes, st, plp, pb, ws, ra); break;
            case 2: e = set_lds(rollout_kernel<2, DynPlan>, ldsBytes); if (e) return (int)e;
                    hipLaunchKernelGGL((rollout_kernel<2, DynPlan>), dim3(grid), dim3(block), ldsBytes, st, plp, pb, ws, ra); break;
            default: e = set_lds(rollout_kernel<4, DynPlan>, ldsBytes); if (e) return (int)e;
                    hipLaunchKernelGGL((rollout_kernel<4, DynPlan>), dim3(grid), dim3(block), ldsBytes, st, plp, pb, ws, ra); break;
        }
    }
    e = hipGetLastError();
    if (e) return (int)e;
    if (g_prof_on) { (void)hipEventRecord(ev1, st); g_prof_events.emplace_back(ev0, ev1); }
    if (cost_sums) {
        hipLaunchKernelGGL(cost_sum_kernel, dim3(1), dim3(256), 0, st, persample, (long)n, cost_sums, errp, mean_args);
        e = hipGetLastError();
        if (e) return (int)e;
    }
    return 0;
}

int nocf_rollout_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                     double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                     float* z_out, float* persample, float* cost_sums, float* zFull, float* ctrlFull,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return rollout_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, zFull, ctrlFull,
                        workspace, workspace_bytes, stream, nullptr);
}

int nocf_rollout_means_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                           double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                           float* z_out, float* persample, float* cost_sums, float* cost_means, float* zFull, float* ctrlFull,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (cost_means && !cost_sums) return NOCF_E_NULL;
    return rollout_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, zFull, ctrlFull,
                        workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cost_means);
}

int nocf_rollout_segments_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                              int32_t nseg, int64_t rows_per_seg, const double* t0s, double t1, const int32_t* nts, const int32_t* slot0s,
                              int32_t stepper, const float* alph, float* z_out, float* persample, float* cost_sums, float* zFull, float* ctrlFull,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!t0s || !nts) return NOCF_E_NULL;
    if (nseg < 1 || nseg > NOCF_MAX_SEG || rows_per_seg < 16 || (rows_per_seg % 16) != 0 || rows_per_seg > 0x7fffffffL ||
        n > (int64_t)nseg * rows_per_seg || n <= (int64_t)(nseg - 1) * rows_per_seg) return NOCF_E_SHAPE;
    SegTab sg;
    memset(&sg, 0, sizeof(sg));
    sg.n = nseg; sg.rows = (int)rows_per_seg;
    int ntmax = 0;
    for (int k = 0; k < nseg; ++k) {
        if (nts[k] < 1) return NOCF_E_SHAPE;
        if (slot0s && slot0s[k] < 0) return NOCF_E_SHAPE;
        sg.t0[k] = t0s[k]; sg.nt[k] = nts[k]; sg.slot0[k] = slot0s ? slot0s[k] : 0;
        ntmax = std::max(ntmax, (int)nts[k]);
    }
    return rollout_impl(phi, prob, x, n, t0s[0], t1, ntmax, stepper, alph, z_out, persample, cost_sums, zFull, ctrlFull,
                        workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr, nullptr, &sg);
}

int nocf_rollout_record_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                            double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                            float* z_out, float* persample, float* cost_sums, float* s_all,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!s_all || !z_out) return NOCF_E_NULL;
    return rollout_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, nullptr, nullptr,
                        workspace, workspace_bytes, stream, s_all);
}

size_t nocf_activation_record_floats(int32_t d, int32_t m, int32_t nTh, int64_t n, int32_t nt, int32_t stepper) {
#ifdef NOCF_JIT_ONLY
    (void)d; (void)m; (void)nTh; (void)n; (void)nt; (void)stepper;
    return 0;
#else
    size_t dummy = 0;
    if (nTh != 2 || n < 1 || nt < 1 || (stepper != NOCF_RK4 && stepper != NOCF_RK1)) return 0;
    // shapes with a recording kernel: the split-role kernel's (m = 512) and the one-CU kernel's (m <= 128 in whole 16-blocks, d+1 <= 16;
    // whether that kernel is taken also depends on the problem: `recorded` of the record call says so)
    const bool duo = (m == 512 || m == 256) && env_int("NOCF_DUO", 1) != 0 &&
                     duo_workspace_bytes(d, m, nTh, d + 1 < 10 ? d + 1 : 10, 1, n, &dummy) == 0;   // (zero-padded widths: no record, its rows have the real width)
    const bool mono = env_int("NOCF_MONO", 1) != 0 && env_int("NOCF_MONO_REC", 1) != 0 && m <= 128 && (m % 16) == 0 && d + 1 <= 32 && m > 32;
    if (!duo && !mono) return 0;
    return (size_t)nt * ((stepper == NOCF_RK4) ? 4 : 1) * (size_t)n * (size_t)(4 * m + d + 1);
#endif
}

int nocf_rollout_record_act_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                                double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                                float* z_out, float* persample, float* cost_sums, float* s_all, float* act_rec, int32_t* recorded,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!s_all || !z_out) return NOCF_E_NULL;
    return rollout_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, nullptr, nullptr,
                        workspace, workspace_bytes, stream, s_all, act_rec, recorded);
}

static int rollout_bwd_impl(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                            const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                            float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                            float* PHIb, float* lam0, const float* act_rec, void* workspace, size_t workspace_bytes, void* stream,
                            int value_only = 0, const float* gbar_in = nullptr, float* sbar_out = nullptr);

// ---- training tape + split-role adjoint (nocf_duo_bwd.inc)
static void tape_offsets(int32_t d, int32_t m, int64_t n, int32_t nt, int32_t stepper, size_t* oU1, size_t* oSc, size_t* total) {
    const size_t R = ((size_t)nt * ((stepper == NOCF_RK4) ? 4 : 1) + 1) * (size_t)n;
    const size_t gpad = (R * (size_t)(d + 1) + 3) / 4 * 4;
    *oU1 = 4 * R * (size_t)m + gpad;
    *oSc = *oU1 + (size_t)n * (size_t)m;
    *total = *oSc + 4 * R;
}

size_t nocf_tape_floats(int32_t d, int32_t m, int32_t nTh, int64_t n, int32_t nt, int32_t stepper) {
#ifdef NOCF_JIT_ONLY
    (void)d; (void)m; (void)nTh; (void)n; (void)nt; (void)stepper;
    return 0;
#else
    size_t dummy = 0, a, b, tot;
    if (nTh != 2 || n < 1 || nt < 1 || (stepper != NOCF_RK4 && stepper != NOCF_RK1)) return 0;
    if (env_int("NOCF_DUO", 1) == 0 || env_int("NOCF_DUO_BWD", 1) == 0) return 0;
    if (m != 512 || duo_workspace_bytes(d, m, nTh, d + 1 < 10 ? d + 1 : 10, 1, n, &dummy) != 0) return 0;
    tape_offsets(d, m, n, nt, stepper, &a, &b, &tot);
    return tot;
#endif
}

int nocf_rollout_tape_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                          double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                          float* z_out, float* persample, float* cost_sums, float* s_all, float* tape, int32_t* recorded,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!s_all || !z_out || !phi) return NOCF_E_NULL;
    if (!tape)
        return rollout_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, nullptr, nullptr,
                            workspace, workspace_bytes, stream, s_all, nullptr, recorded);
    size_t oU1, oSc, tot;
    tape_offsets(phi->d, phi->m, n, nt, stepper, &oU1, &oSc, &tot);
    return rollout_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, nullptr, nullptr,
                        workspace, workspace_bytes, stream, s_all, tape, recorded, tape + oU1, tape + oSc);
}

int nocf_poison_if_failed_f32(float* buf, int64_t count, void* stream) {
    if (!buf) return NOCF_E_NULL;
    if (!g_last_errp || count < 1) return 0;
    hipLaunchKernelGGL(poison_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, buf, (long)count, g_last_errp);
    return (int)hipGetLastError();
}

size_t nocf_dw_scratch_floats(void) {
#ifdef NOCF_JIT_ONLY
    return 0;
#else
    return duo_dw_scratch_floats();
#endif
}

size_t nocf_bwd_colsum_floats(int64_t n) {
#ifdef NOCF_JIT_ONLY
    (void)n;
    return 0;
#else
    return n < 1 ? 0 : duo_bwd_colsum_floats((long)n);
#endif
}

static int bwd_tape_impl(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper,
                         const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                         const float* tape, float* Y, float* Ab, float* Wb, float* Qb, float* Ob, float* Gb, float* lam0,
                         float* dK1, float* dK0, float* dw_scratch, size_t dw_scratch_floats, int32_t* dw_done,
                         float* colsum, size_t colsum_floats, void* workspace, size_t workspace_bytes, void* stream) {
    if (dw_done) *dw_done = 0;
#ifdef NOCF_JIT_ONLY
    (void)dK1; (void)dK0; (void)dw_scratch; (void)dw_scratch_floats; (void)colsum; (void)colsum_floats;
    (void)phi; (void)prob; (void)n; (void)nt; (void)stepper; (void)alph; (void)inv_n; (void)s_all; (void)z_final; (void)hs; (void)tape;
    (void)Y; (void)Ab; (void)Wb; (void)Qb; (void)Ob; (void)Gb; (void)lam0; (void)workspace; (void)workspace_bytes; (void)stream;
    return NOCF_E_SHAPE;
#else
    int rc = check_phi(phi);
    if (rc) return rc;
    if (!alph || !s_all || !z_final || !hs || !tape || !Y || !Ab || (!Wb && !colsum) || !Qb || !Ob || !Gb || !workspace) return NOCF_E_NULL;
    if (n < 1 || nt < 1) return NOCF_E_SHAPE;
    if (colsum && colsum_floats < nocf_bwd_colsum_floats(n)) return NOCF_E_WORKSPACE;
    if (stepper != NOCF_RK4 && stepper != NOCF_RK1) return NOCF_E_STEPPER;
    DevProb pb;
    rc = fill_prob(prob, phi->d, &pb);
    if (rc) return rc;
    size_t oU1, oSc, tot;
    tape_offsets(phi->d, phi->m, n, nt, stepper, &oU1, &oSc, &tot);
    DuoBwdHost h;
    h.csum = colsum; h.csum_floats = colsum_floats;
    h.s_all = s_all; h.z_final = z_final; h.hs = hs; h.tape = tape; h.tapeU1 = tape + oU1; h.tapeSc = tape + oSc;
    h.n = n; h.nt = nt; h.stepper = stepper;
    h.a0 = alph[0]; h.a3 = alph[3]; h.a4 = alph[4]; h.a5 = alph[5]; h.inv_n = (float)inv_n;
    h.Y = Y; h.Ab = Ab; h.Wb = Wb; h.Qb = Qb; h.Ob = Ob; h.Gb = Gb; h.lam0 = lam0;
    h.dK1 = dK1; h.dK0 = dK0; h.dw_scratch = dw_scratch; h.dw_scratch_floats = dw_scratch_floats; h.dw_done = dw_done;
    h.stamps = g_stamp_buf;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_prof_on) { if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown; }
    const unsigned* errp = nullptr;
    g_last_errp = nullptr;
    rc = duo_bwd_launch(phi, pb, h, (float*)workspace, workspace_bytes, (hipStream_t)stream, &errp, env_int("NOCF_DEBUG", 0), ev0, ev1);
    if (rc == 0) {
        g_last_errp = errp;
        g_last_kernel = "rollout_duo_bwd_kernel";
        if (g_prof_on) g_prof_events.emplace_back(ev0, ev1);
        return 0;
    }
    if (g_prof_on) { (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); }
    return rc == 1 ? NOCF_E_SHAPE : rc;
#endif
}

int nocf_rollout_bwd_tape_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper,
                              const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                              const float* tape, float* Y, float* Ab, float* Wb, float* Qb, float* Ob, float* Gb, float* lam0,
                              float* dK1, float* dK0, float* dw_scratch, size_t dw_scratch_floats, int32_t* dw_done,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return bwd_tape_impl(phi, prob, n, nt, stepper, alph, inv_n, s_all, z_final, hs, tape, Y, Ab, Wb, Qb, Ob, Gb, lam0,
                         dK1, dK0, dw_scratch, dw_scratch_floats, dw_done, nullptr, 0, workspace, workspace_bytes, stream);
}

int nocf_rollout_bwd_tape_sums_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper,
                                   const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                                   const float* tape, float* Y, float* Ab, float* Qb, float* Ob, float* Gb, float* lam0,
                                   float* dK1, float* dK0, float* dw_scratch, size_t dw_scratch_floats, int32_t* dw_done,
                                   float* colsum, size_t colsum_floats, void* workspace, size_t workspace_bytes, void* stream) {
    if (!colsum) return NOCF_E_NULL;
    return bwd_tape_impl(phi, prob, n, nt, stepper, alph, inv_n, s_all, z_final, hs, tape, Y, Ab, nullptr, Qb, Ob, Gb, lam0,
                         dK1, dK0, dw_scratch, dw_scratch_floats, dw_done, colsum, colsum_floats, workspace, workspace_bytes, stream);
}

int nocf_rollout_bwd_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                         const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                         float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                         float* PHIb, float* lam0, void* workspace, size_t workspace_bytes, void* stream) {
    return rollout_bwd_impl(phi, prob, n, nt, stepper, t1, alph, inv_n, s_all, z_final, hs, Y, Ob, V, Ab, Qb, U0, Wb, Gb, Sx, PHIb, lam0,
                            nullptr, workspace, workspace_bytes, stream);
}

int nocf_rollout_bwd_act_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                             const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                             float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                             float* PHIb, float* lam0, const float* act_rec, void* workspace, size_t workspace_bytes, void* stream) {
    return rollout_bwd_impl(phi, prob, n, nt, stepper, t1, alph, inv_n, s_all, z_final, hs, Y, Ob, V, Ab, Qb, U0, Wb, Gb, Sx, PHIb, lam0,
                            act_rec, workspace, workspace_bytes, stream);
}

int nocf_phi_grad_bwd_f32(const NocfPhi* phi, const float* s, int64_t n, const float* gbar, float* sbar,
                          float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!gbar || !sbar) return NOCF_E_NULL;
    return rollout_bwd_impl(phi, nullptr, n, 0, NOCF_RK4, 0.0, nullptr, 0.0, s, nullptr, nullptr, Y, Ob, V, Ab, Qb, U0, Wb, Gb, Sx, nullptr, nullptr,
                            nullptr, workspace, workspace_bytes, stream, 2, gbar, sbar);
}

int nocf_phi_value_bwd_f32(const NocfPhi* phi, const float* s, int64_t n, float* gout,
                           float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!gout) return NOCF_E_NULL;
    return rollout_bwd_impl(phi, nullptr, n, 0, NOCF_RK4, 0.0, nullptr, 0.0, s, nullptr, nullptr, Y, Ob, V, Ab, Qb, U0, Wb, Gb, Sx, gout, nullptr,
                            nullptr, workspace, workspace_bytes, stream, 1);
}

static int rollout_bwd_impl(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                            const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                            float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                            float* PHIb, float* lam0, const float* act_rec, void* workspace, size_t workspace_bytes, void* stream,
                            int value_only, const float* gbar_in, float* sbar_out) {
    int rc = check_phi(phi);
    if (rc) return rc;
    if (value_only) { if (!s_all || nt != 0 || (value_only == 2 && (!gbar_in || !sbar_out))) return NOCF_E_NULL; }
    else if (!alph || !z_final || !hs) return NOCF_E_NULL;
    if (!s_all || !Y || !Ob || !V || !Ab || !Qb || !U0 || !Wb || !Gb || !Sx || (!PHIb && value_only != 2) || !workspace)
        return NOCF_E_NULL;
    if (n < 1 || (nt < 1 && !value_only)) return NOCF_E_SHAPE;
    if (stepper != NOCF_RK4 && stepper != NOCF_RK1) return NOCF_E_STEPPER;
    DevProb pb;
    if (value_only) {                                          // (no physics on this path: Phi alone)
        memset(&pb, 0, sizeof(pb));
        pb.kind = NOCF_PROB_CROSS2D; pb.obstacle = NOCF_OBS_NONE; pb.nAgents = 1; pb.agentDim = 2; pb.r = 1.0;
    } else {
        rc = fill_prob(prob, phi->d, &pb);
        if (rc) return rc;
    }
    DevPlan pl;
    rc = make_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, &pl, 1);
    if (rc) return rc;
    if (pl.T != 4) return NOCF_E_SHAPE;
    pl.cb = phi->cb;
    if (workspace_bytes < plan_ws_bytes(pl)) return NOCF_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    rc = pack_weights(pl, phi, ws, st);
    if (rc) return rc;
    const DevPlan* plp = reinterpret_cast<const DevPlan*>(ws + pl.oPlan);
    BwdArgs ba;
    ba.sAll = s_all; ba.zT = z_final; ba.hs = hs; ba.n = n; ba.nt = nt; ba.nstage = (stepper == NOCF_RK4) ? 4 : 1;
    ba.t1 = (float)t1;
    if (alph) { ba.a0 = alph[0]; ba.a3 = alph[3]; ba.a4 = alph[4]; ba.a5 = alph[5]; } else { ba.a0 = ba.a3 = ba.a4 = ba.a5 = 0.f; }
    ba.inv_n = (float)inv_n; ba.value_only = value_only; ba.gbar_in = gbar_in; ba.sbar_out = sbar_out;
    ba.Y = Y; ba.Ob = Ob; ba.V = V; ba.Ab = Ab; ba.Qb = Qb; ba.U0 = U0; ba.Wb = Wb; ba.Gb = Gb; ba.Sx = Sx;
    ba.PHIb = PHIb; ba.lam0 = lam0;
    ba.act = (act_rec && phi->nTh == 2) ? act_rec : nullptr; ba.actRows = (long)nt * ba.nstage * n;
    ba.lstride = ((long)nt * ba.nstage + 2) * n * phi->m;
    ba.gpart = nullptr; ba.gstride = 0;
    const size_t ldsBytes = (size_t)pl.ldsFloats * 4;
    const void* fk = nullptr;
    if (env_int("NOCF_FIXED", 1)) {
#define NOCF_TRY_FIXED(D, M, NTH, R, NAG) \
        if (!fk && plan_is<FixedPlan<D, M, NTH, R, NAG, 1>>(pl)) fk = reinterpret_cast<const void*>(rollout_bwd_kernel<1, FixedPlan<D, M, NTH, R, NAG, 1>>);
        FIXED_SHAPES(NOCF_TRY_FIXED)
        FIXED_SHAPES_EXTRA(NOCF_TRY_FIXED)
        FIXED_SHAPES_TRAIN(NOCF_TRY_FIXED)
#undef NOCF_TRY_FIXED
    }
    if (!fk) fk = reinterpret_cast<const void*>(rollout_bwd_kernel<1, DynPlan>);
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e) return (int)e;
    {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (g_prof_on) {
            if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
            (void)hipEventRecord(ev0, st);
        }
        void* args[] = {(void*)&plp, (void*)&pb, (void*)&ws, (void*)&ba};
        e = hipLaunchKernel(fk, dim3((int)((n + 3) / 4)), dim3(pl.nwaves * 64), args, ldsBytes, st);
        if (e) return (int)e;
        if (g_prof_on) { (void)hipEventRecord(ev1, st); g_prof_events.emplace_back(ev0, ev1); }
        g_last_kernel = "rollout_bwd_kernel";
        g_last_errp = nullptr;
    }
    return (int)hipGetLastError();
}

// ---- the adjoint on the one-CU weight-stationary layout (nocf_mono_bwd.inc): medium two-layer networks, every problem class
int64_t nocf_mid_grad_rows(int32_t d, int32_t m, int32_t nTh, int32_t r, int32_t n_agents, int64_t n) {
#ifdef NOCF_JIT_ONLY
    (void)d; (void)m; (void)nTh; (void)r; (void)n_agents; (void)n;
    return 0;
#else
    if (env_int("NOCF_MONO", 1) == 0 || env_int("NOCF_MONO_BWD", 1) == 0 || n < 1 || m <= 32) return 0;
    DevPlan pl;
    if (make_plan(d, m, nTh, r, n_agents, &pl, 0)) return 0;
    MonoPlan mpl;
    if (make_mono_plan(pl, n_agents, &mpl, true)) return 0;
    if (!((mpl.KBM == 8 || mpl.KBM == 6 || mpl.KBM == 4) && (mpl.KBD == 1 || mpl.KBD == 2))) return 0;
    return std::min<int64_t>((n + 15) / 16, 1024);           // workgroups = partial vectors: beyond 1024 tiles a workgroup takes several
#endif
}

int nocf_rollout_bwd_mid_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                             const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                             const float* act_rec, float* gpart, int64_t gpart_rows, float* lam0, void* workspace, size_t workspace_bytes,
                             void* stream) {
#ifdef NOCF_JIT_ONLY
    (void)phi; (void)prob; (void)n; (void)nt; (void)stepper; (void)t1; (void)alph; (void)inv_n; (void)s_all; (void)z_final; (void)hs;
    (void)act_rec; (void)gpart; (void)gpart_rows; (void)lam0; (void)workspace; (void)workspace_bytes; (void)stream;
    return NOCF_E_SHAPE;
#else
    int rc = check_phi(phi);
    if (rc) return rc;
    if (!alph || !s_all || !z_final || !hs || !gpart || !workspace) return NOCF_E_NULL;
    if (n < 1 || nt < 1) return NOCF_E_SHAPE;
    if (stepper != NOCF_RK4 && stepper != NOCF_RK1) return NOCF_E_STEPPER;
    DevProb pb;
    rc = fill_prob(prob, phi->d, &pb);
    if (rc) return rc;
    const int64_t rows = nocf_mid_grad_rows(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, n);
    if (rows == 0) return NOCF_E_SHAPE;
    if (gpart_rows < rows) return NOCF_E_WORKSPACE;
    DevPlan pl;
    rc = make_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, &pl, 0);
    if (rc) return rc;
    pl.cb = phi->cb;
    MonoPlan mpl;
    rc = make_mono_plan(pl, pb.nAgents, &mpl, true);
    if (rc) return rc;
    if (workspace_bytes < mono_ws_bytes(mpl)) return NOCF_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    rc = pack_weights(mpl.pp, phi, ws, st);                    // the padded vectors and the copy of A sit in front of the plan record
    if (rc) return rc;
    DevPhi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev};
    hipLaunchKernelGGL(mono_pack_kernel, dim3(64), dim3(256), 0, st, mpl, P, ws);
    BwdArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.sAll = s_all; ba.zT = z_final; ba.hs = hs; ba.n = n; ba.nt = nt; ba.nstage = (stepper == NOCF_RK4) ? 4 : 1;
    ba.t1 = (float)t1;
    ba.a0 = alph[0]; ba.a3 = alph[3]; ba.a4 = alph[4]; ba.a5 = alph[5]; ba.inv_n = (float)inv_n;
    ba.lam0 = lam0;
    ba.gpart = gpart; ba.gstride = nocf_small_grad_floats(phi->d, phi->m);
    ba.act = (act_rec && (phi->m % 16) == 0) ? act_rec : nullptr; ba.actRows = (long)nt * ba.nstage * n;
    const size_t ldsBytes = (size_t)mpl.pp.ldsFloats * 4;
    const void* fk = nullptr;
#define NOCF_MBW_PICK(M_, D_) if (mpl.KBM == M_ && mpl.KBD == D_) fk = reinterpret_cast<const void*>(rollout_mono_bwd_kernel<M_, D_>);
    NOCF_MBW_PICK(8, 1) NOCF_MBW_PICK(6, 1) NOCF_MBW_PICK(4, 1) NOCF_MBW_PICK(8, 2) NOCF_MBW_PICK(6, 2) NOCF_MBW_PICK(4, 2)
#undef NOCF_MBW_PICK
    if (!fk) return NOCF_E_SHAPE;
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e) return (int)e;
    if (env_int("NOCF_DEBUG", 0)) fprintf(stderr, "[nocf] mono adjoint kernel: %d hidden k-blocks, LDS %zu B/workgroup\n", mpl.KBM, ldsBytes);
    const MonoPlan* mpp = reinterpret_cast<const MonoPlan*>(ws + mpl.pp.oPlan);
    const float* wsc = ws;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_prof_on) {
        if (hipEventCreate(&ev0) || hipEventCreate(&ev1)) return (int)hipErrorUnknown;
        (void)hipEventRecord(ev0, st);
    }
    void* args[] = {(void*)&mpp, (void*)&pb, (void*)&wsc, (void*)&ba};
    e = hipLaunchKernel(fk, dim3((unsigned)rows), dim3(256), args, ldsBytes, st);
    if (e) return (int)e;
    if (g_prof_on) { (void)hipEventRecord(ev1, st); g_prof_events.emplace_back(ev0, ev1); }
    g_last_kernel = "rollout_mono_bwd_kernel";
    g_last_errp = nullptr;
    return (int)hipGetLastError();
#endif
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
    const DevPlan* plp = reinterpret_cast<const DevPlan*>(ws + pl.oPlan);
    const size_t ldsBytes = (size_t)pl.ldsFloats * 4;
    const int grid = (int)((n + pl.T - 1) / pl.T);
    const int block = pl.nwaves * 64;
    hipError_t e;
    switch (pl.T / 4) {
        case 1: e = set_lds(phi_kernel<1>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(phi_kernel<1>, dim3(grid), dim3(block), ldsBytes, st, plp, ws, s, (long)n, grad, value); break;
        case 2: e = set_lds(phi_kernel<2>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(phi_kernel<2>, dim3(grid), dim3(block), ldsBytes, st, plp, ws, s, (long)n, grad, value); break;
        default: e = set_lds(phi_kernel<4>, ldsBytes); if (e) return (int)e;
                hipLaunchKernelGGL(phi_kernel<4>, dim3(grid), dim3(block), ldsBytes, st, plp, ws, s, (long)n, grad, value); break;
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

// ------------------------------------------------------------------------------------------
// double precision (nocf_f64.inc)
// ------------------------------------------------------------------------------------------
size_t nocf_workspace_bytes_f64(int32_t d, int32_t m, int32_t nTh) {
    if (d < 1 || m < 1 || nTh < 2) return 0;
    return f64_ws_doubles(d, m, nTh) * sizeof(double);
}

static int rollout_f64_impl(const NocfPhi64* phi, const NocfProb64* prob, const double* x, int64_t n,
                            double t0, double t1, int32_t nt, int32_t stepper, const double* alph,
                            double* z_out, double* persample, double* cost_sums, double* zFull, double* ctrlFull, double* s_all,
                            void* workspace, size_t workspace_bytes, void* stream);
int nocf_rollout_f64(const NocfPhi64* phi, const NocfProb64* prob, const double* x, int64_t n,
                     double t0, double t1, int32_t nt, int32_t stepper, const double* alph,
                     double* z_out, double* persample, double* cost_sums, double* zFull, double* ctrlFull,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return rollout_f64_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, zFull, ctrlFull, nullptr, workspace, workspace_bytes, stream);
}
int nocf_rollout_record_f64(const NocfPhi64* phi, const NocfProb64* prob, const double* x, int64_t n,
                            double t0, double t1, int32_t nt, int32_t stepper, const double* alph,
                            double* z_out, double* persample, double* cost_sums, double* s_all,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!s_all || !z_out) return NOCF_E_NULL;
    return rollout_f64_impl(phi, prob, x, n, t0, t1, nt, stepper, alph, z_out, persample, cost_sums, nullptr, nullptr, s_all, workspace, workspace_bytes, stream);
}
static int rollout_f64_impl(const NocfPhi64* phi, const NocfProb64* prob, const double* x, int64_t n,
                            double t0, double t1, int32_t nt, int32_t stepper, const double* alph,
                            double* z_out, double* persample, double* cost_sums, double* zFull, double* ctrlFull, double* s_all,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!phi || !prob) return NOCF_E_NULL;
    if (!phi->K0 || !phi->b0 || !phi->K || !phi->b || !phi->w || !phi->A || !phi->cw || !phi->cb_dev) return NOCF_E_NULL;
    if (!x || !alph || !workspace || !prob->xtarget) return NOCF_E_NULL;
    if (n < 1 || nt < 1 || phi->d < 1 || phi->m < 1 || phi->nTh < 2 || phi->r < 1 || phi->r > 16) return NOCF_E_SHAPE;
    if (stepper != NOCF_RK4 && stepper != NOCF_RK1) return NOCF_E_STEPPER;
    if ((zFull != nullptr) != (ctrlFull != nullptr)) return NOCF_E_NULL;
    if (cost_sums && !persample) return NOCF_E_NULL;
    NocfProb p32;                                    // the same checks as the fp32 entry (kind / obstacle / agent count)
    p32.kind = prob->kind; p32.obstacle = prob->obstacle; p32.n_agents = prob->n_agents; p32.training = prob->training;
    p32.r = prob->r; p32.alph_Q = prob->alph_Q; p32.alph_W = prob->alph_W; p32.mass = prob->mass; p32.grav = prob->grav; p32.xtarget = nullptr;
    DevProb pb32;
    int rc = fill_prob(&p32, phi->d, &pb32);
    if (rc) return rc;
    F64Prob pb{pb32.kind, pb32.obstacle, pb32.nAgents, pb32.training, pb32.agentDim, prob->r, prob->alph_Q, prob->alph_W, prob->mass, prob->grav, prob->xtarget};
    if (workspace_bytes < nocf_workspace_bytes_f64(phi->d, phi->m, phi->nTh)) return NOCF_E_WORKSPACE;
    // samples per workgroup: 4 when the batch still fills the chip that way, fewer for small batches or tight LDS
    F64Plan pl;
    int T = 0;
    const int pref[3] = {n >= 1024 ? 4 : (n >= 512 ? 2 : 1), 2, 1};
    for (int q = 0; q < 3 && !T; ++q) if ((q == 0 || pref[q] < pref[0]) && make_f64_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, pref[q], &pl) == 0) T = pref[q];
    if (!T) return NOCF_E_LDS;
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
    F64Phi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev, phi->d, phi->m, phi->nTh, phi->r};
    hipLaunchKernelGGL(f64_pack_kernel, dim3(512), dim3(256), 0, st, pl, P, ws);
    F64Args ra;
    ra.x = x; ra.n = n; ra.t0 = t0; ra.t1 = t1; ra.h = (t1 - t0) / nt; ra.nt = nt; ra.stepper = stepper; ra.a0 = alph[0];
    ra.z_out = z_out; ra.persample = persample; ra.zFull = zFull; ra.ctrlFull = ctrlFull; ra.sAll = s_all;
    ra.cdim = nocf_ctrl_dim(&p32, phi->d);
#ifdef NOCF_STAMPS
    { const int dbg = env_int("NOCF_F64_DBG", 0); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f64_dbg), &dbg, sizeof(dbg)); }
#endif
    const size_t ldsBytes = (size_t)pl.ldsDoubles * 8;
    const bool wide = phi->m > 256;
    const void* fk = wide ? (T == 4 ? reinterpret_cast<const void*>(rollout_f64_kernel<4, true>)
                             : T == 2 ? reinterpret_cast<const void*>(rollout_f64_kernel<2, true>) : reinterpret_cast<const void*>(rollout_f64_kernel<1, true>))
                          : (T == 4 ? reinterpret_cast<const void*>(rollout_f64_kernel<4, false>)
                             : T == 2 ? reinterpret_cast<const void*>(rollout_f64_kernel<2, false>) : reinterpret_cast<const void*>(rollout_f64_kernel<1, false>));
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e) return (int)e;
    if (env_int("NOCF_DEBUG", 0)) fprintf(stderr, "[nocf] f64 kernel: %d sample(s) per workgroup, LDS %zu B\n", T, ldsBytes);
    const double* wsc = ws;
    void* args[] = {(void*)&pl, (void*)&P, (void*)&pb, (void*)&wsc, (void*)&ra};
    e = hipLaunchKernel(fk, dim3((unsigned)((n + T - 1) / T)), dim3(256), args, ldsBytes, st);
    if (e) return (int)e;
    g_last_kernel = "rollout_f64_kernel";
    if (cost_sums) hipLaunchKernelGGL(f64_cost_sum_kernel, dim3(1), dim3(256), 0, st, persample, (long)n, cost_sums);
    e = hipGetLastError();
    return (int)e;
}

// the adjoint in double precision (nocf_f64_bwd.inc): rows streamed in the layout of nocf_rollout_bwd_f32
int nocf_rollout_bwd_f64(const NocfPhi64* phi, const NocfProb64* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                         const double* alph, double inv_n, const double* s_all, const double* z_final, const double* hs,
                         double* Y, double* Ob, double* V, double* Ab, double* Qb, double* U0, double* Wb, double* Gb, double* Sx,
                         double* PHIb, double* lam0, void* workspace, size_t workspace_bytes, void* stream) {
    if (!phi || !prob) return NOCF_E_NULL;
    if (!phi->K0 || !phi->b0 || !phi->K || !phi->b || !phi->w || !phi->A || !phi->cw || !phi->cb_dev) return NOCF_E_NULL;
    if (!alph || !s_all || !z_final || !hs || !Y || !Ob || !V || !Ab || !Qb || !U0 || !Wb || !Gb || !Sx || !PHIb || !workspace || !prob->xtarget) return NOCF_E_NULL;
    if (n < 1 || nt < 1 || phi->d < 1 || phi->m < 1 || phi->nTh < 2 || phi->r < 1 || phi->r > 16) return NOCF_E_SHAPE;
    if (stepper != NOCF_RK4 && stepper != NOCF_RK1) return NOCF_E_STEPPER;
    NocfProb p32;
    p32.kind = prob->kind; p32.obstacle = prob->obstacle; p32.n_agents = prob->n_agents; p32.training = prob->training;
    p32.r = prob->r; p32.alph_Q = prob->alph_Q; p32.alph_W = prob->alph_W; p32.mass = prob->mass; p32.grav = prob->grav; p32.xtarget = nullptr;
    DevProb pb32;
    int rc = fill_prob(&p32, phi->d, &pb32);
    if (rc) return rc;
    F64Prob pb{pb32.kind, pb32.obstacle, pb32.nAgents, pb32.training, pb32.agentDim, prob->r, prob->alph_Q, prob->alph_W, prob->mass, prob->grav, prob->xtarget};
    if (workspace_bytes < nocf_workspace_bytes_f64(phi->d, phi->m, phi->nTh)) return NOCF_E_WORKSPACE;
    const bool wide = phi->m > 256;                            // (the register-tiled products; they run at 2 or 1 samples per workgroup here)
    F64BwdPlan bp;
    int T = 0;
    const int tpref = env_int("NOCF_F64_BWD_T", 0);             // (diagnostic: force the samples per workgroup)
    for (int cand : {4, 2, 1}) {
        if (wide && cand == 4) continue;
        if (tpref && cand > tpref) continue;
        if (make_f64_bwd_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, cand, &bp) == 0) { T = cand; break; }
    }
    if (!T) return NOCF_E_LDS;
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
    F64Phi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev, phi->d, phi->m, phi->nTh, phi->r};
    {   // the transposed images K0T / KT of the evaluation kernel's workspace (the packed MFMA images are not used here)
        F64Plan plf;
        if (make_f64_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, 1, &plf)) return NOCF_E_LDS;
        hipLaunchKernelGGL(f64_pack_kernel, dim3(512), dim3(256), 0, st, plf, P, ws);
    }
    F64BwdArgs ba;
    ba.sAll = s_all; ba.zT = z_final; ba.hs = hs; ba.n = n; ba.nt = nt; ba.nstage = (stepper == NOCF_RK4) ? 4 : 1;
    ba.t1 = t1; ba.a0 = alph[0]; ba.a3 = alph[3]; ba.a4 = alph[4]; ba.a5 = alph[5]; ba.inv_n = inv_n;
    ba.Y = Y; ba.Ob = Ob; ba.Wb = Wb; ba.V = V; ba.Ab = Ab; ba.Qb = Qb; ba.U0 = U0; ba.lstride = ((long)nt * ba.nstage + 2) * n * phi->m;
    ba.Gb = Gb; ba.Sx = Sx; ba.PHIb = PHIb; ba.lam0 = lam0;
    const size_t ldsBytes = (size_t)bp.ldsDoubles * 8;
    const void* fk = wide ? (T == 2 ? reinterpret_cast<const void*>(rollout_bwd_f64_kernel<2, true>) : reinterpret_cast<const void*>(rollout_bwd_f64_kernel<1, true>))
                          : (T == 4 ? reinterpret_cast<const void*>(rollout_bwd_f64_narrow_kernel<4>)
                             : T == 2 ? reinterpret_cast<const void*>(rollout_bwd_f64_narrow_kernel<2>) : reinterpret_cast<const void*>(rollout_bwd_f64_narrow_kernel<1>));
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e) return (int)e;
    if (env_int("NOCF_DEBUG", 0)) fprintf(stderr, "[nocf] f64 adjoint kernel: %d sample(s) per workgroup, LDS %zu B\n", T, ldsBytes);
    const double* wsc = ws;
    void* args[] = {(void*)&bp, (void*)&P, (void*)&pb, (void*)&wsc, (void*)&ba};
    e = hipLaunchKernel(fk, dim3((unsigned)((n + T - 1) / T)), dim3(256), args, ldsBytes, st);
    if (e) return (int)e;
    g_last_kernel = "rollout_bwd_f64_kernel";
    g_last_errp = nullptr;
    return (int)hipGetLastError();
}

int nocf_phi_f64(const NocfPhi64* phi, const double* s, int64_t n, double* value, double* grad,
                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!phi || !s || !workspace || (!value && !grad)) return NOCF_E_NULL;
    if (!phi->K0 || !phi->b0 || !phi->K || !phi->b || !phi->w || !phi->A || !phi->cw || !phi->cb_dev) return NOCF_E_NULL;
    if (n < 1 || phi->d < 1 || phi->m < 1 || phi->nTh < 2 || phi->r < 1 || phi->r > 16) return NOCF_E_SHAPE;
    if (workspace_bytes < nocf_workspace_bytes_f64(phi->d, phi->m, phi->nTh)) return NOCF_E_WORKSPACE;
    F64Plan pl;
    int T = 0;
    const int pref[3] = {n >= 1024 ? 4 : (n >= 512 ? 2 : 1), 2, 1};
    for (int q = 0; q < 3 && !T; ++q) if ((q == 0 || pref[q] < pref[0]) && make_f64_plan(phi->d, phi->m, phi->nTh, phi->r, 1, pref[q], &pl) == 0) T = pref[q];
    if (!T) return NOCF_E_LDS;
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
    F64Phi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev, phi->d, phi->m, phi->nTh, phi->r};
    hipLaunchKernelGGL(f64_pack_kernel, dim3(512), dim3(256), 0, st, pl, P, ws);
    const size_t ldsBytes = (size_t)pl.ldsDoubles * 8;
    const bool wide = phi->m > 256;
    const void* fk = wide ? (T == 4 ? reinterpret_cast<const void*>(phi_f64_kernel<4, true>)
                             : T == 2 ? reinterpret_cast<const void*>(phi_f64_kernel<2, true>) : reinterpret_cast<const void*>(phi_f64_kernel<1, true>))
                          : (T == 4 ? reinterpret_cast<const void*>(phi_f64_kernel<4, false>)
                             : T == 2 ? reinterpret_cast<const void*>(phi_f64_kernel<2, false>) : reinterpret_cast<const void*>(phi_f64_kernel<1, false>));
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e) return (int)e;
    const double* wsc = ws;
    long nn = (long)n;
    void* args[] = {(void*)&pl, (void*)&P, (void*)&wsc, (void*)&s, (void*)&nn, (void*)&value, (void*)&grad};
    e = hipLaunchKernel(fk, dim3((unsigned)((n + T - 1) / T)), dim3(256), args, ldsBytes, st);
    if (e) return (int)e;
    return (int)hipGetLastError();
}

int nocf_prob_eval_f64(const NocfProb64* prob, int32_t d, const double* x, const double* p, int64_t n,
                       double* lhqw, double* gradpH, double* ctrls, void* stream) {
    if (!prob || !x || !p || !prob->xtarget || (!lhqw && !gradpH && !ctrls)) return NOCF_E_NULL;
    if (n < 1 || d < 1) return NOCF_E_SHAPE;
    NocfProb p32;
    p32.kind = prob->kind; p32.obstacle = prob->obstacle; p32.n_agents = prob->n_agents; p32.training = prob->training;
    p32.r = prob->r; p32.alph_Q = prob->alph_Q; p32.alph_W = prob->alph_W; p32.mass = prob->mass; p32.grav = prob->grav; p32.xtarget = nullptr;
    DevProb pb32;
    int rc = fill_prob(&p32, d, &pb32);
    if (rc) return rc;
    F64Prob pb{pb32.kind, pb32.obstacle, pb32.nAgents, pb32.training, pb32.agentDim, prob->r, prob->alph_Q, prob->alph_W, prob->mass, prob->grav, prob->xtarget};
    F64Plan pl;
    if (make_f64_plan(d, 1, 2, 1, pb.nAgents, 4, &pl) != 0) return NOCF_E_LDS;
    const size_t ldsBytes = (size_t)pl.ldsDoubles * 8;
    const void* fk = reinterpret_cast<const void*>(prob_f64_kernel<4>);
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    if (e) return (int)e;
    long nn = (long)n;
    int cdim = nocf_ctrl_dim(&p32, d);
    void* args[] = {(void*)&pl, (void*)&pb, (void*)&x, (void*)&p, (void*)&nn, (void*)&lhqw, (void*)&gradpH, (void*)&ctrls, (void*)&cdim};
    e = hipLaunchKernel(fk, dim3((unsigned)((n + 3) / 4)), dim3(256), args, ldsBytes, (hipStream_t)stream);
    if (e) return (int)e;
    return (int)hipGetLastError();
}

}  // extern "C"
