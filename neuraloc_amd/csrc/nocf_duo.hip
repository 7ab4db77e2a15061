// nocf_duo.hip -- weight-STATIONARY rollout for wide networks (m = 512 or 256, nTh = 2, point-agent problems), rounds 3-5:
// two ROLE workgroups per CU, two instruction streams per SIMD.
//
// Why.  Round 2's slab kernel kept a member's slice of K1 in registers in BOTH orientations (2 x 128 per lane), which
// forces one wave per SIMD: a single instruction stream that has to do the gathers, the epilogues and the owners' work
// serially around its MFMAs (0.43 of the fp32 matrix peak; the eight L2 gathers of an evaluation alone were 30 % of it).
// Here every (group, member) is TWO workgroups of 4 waves, 256 registers per wave, two workgroups per CU:
//     role A (member c):  W2 = K1[H_c,:] resident (128 AccVGPRs)      role B (member c):  W3 = K1[:,H_c]^T resident
//       own  sum the 8 partial gradients of the own samples, RK update,    xq  obstacle + pair sums of the own samples (x only)
//            publish the next stage state S                             P3  a[H_c] = w + hN W3 v        (B = V fragments)
//       P1   o[H_c] = K0[H_c,:] s + b0   (A = 40 VGPRs, B = S fragments)     y = tanh(o) . a   (tanh(o) arrives as TH)
//            publish u0 = sigma(o) as U, tanh(o) as TH                  P4  partial g = K0[H_c,:]^T y   (A = LDS image)
//       P2   q[H_c] = W2 u0 + b1         (B = U fragments)                   publish the partial as G
//            publish v = tanh(q) . w as V
//       cst  (owner waves, behind P2) cost integrals of the evaluation just answered; z = A s, A^T z + c for the next owner's step
// so that one wave's L2 waits, barriers and LDS latencies run under the other wave's work.  (Only waits: an fp32 MFMA occupies the
// SIMD's vector ALU -- a co-resident wave makes no VALU progress while the other streams MFMAs, tools/micro/ -- so every VALU
// instruction of either role costs MFMA time, and work that is not on the dependency cycle S -> U -> V -> G -> S is placed where
// its wave would otherwise wait: the cost pass in role B's wait for V, `cst` behind P2 while v travels.)
// A group = 8 members x 2 roles = 16 workgroups works on NT <= 4 tiles of 16 samples; hidden units H_c = [64c, 64c+64); wave w of
// a workgroup owns the 16 features 64c + 16w ...; sample 2c + j of a tile is owned by member c (wave (2t + j) & 3 of its role A).
//
// Matrix work: v_mfma_f32_16x16x4_f32, weights as the A operand (M = 16 features), samples as N = 16; a lane's 4 result
// registers are 4 consecutive features of one sample = the B fragment of the next GEMM.  Every exchange buffer is a
// list of 1 KiB fragments [64 lanes][4 floats].  A workgroup GATHERS a tile's fragments from L2 into an LDS staging buffer (each
// wave its share, all requests of a poll in flight together, sc1 loads: never through the CU's L1), one barrier, and the four waves
// multiply from LDS (ring of 4 ds_read_b128, 3 k-blocks ahead; 33.8 cycles per MFMA).  Stores and LDS reads that only have to be
// ISSUED before a product ends (slot resets, the epilogue's bias vectors, the tanh(o) request) sit inside the MFMA stream (du_gemm_lds's
// mid hook): they issue in the MFMAs' shadow instead of on the path.
//
// Exchange protocol: DATA-TAGGED, no flags.  Every buffer exists twice (parity of the evaluation counter e; the cost scalars QW three
// times) and starts as all-ones words (0xFFFFFFFF: a NaN pattern no result has).  A producer stores a fragment into the parity-e
// buffer; a consumer loads it and re-loads while any word of its 16 bytes is still the sentinel (bounded: on a timeout an error word
// is set, every wave stops waiting, the host raises).  Slots are RESET (sentinel stores) by their producer at a point chosen so that
//   (H2) every reader of the old contents has finished: the reset of X(e-1) is issued only after the producer has SEEN a
//        fragment of evaluation e whose existence implies it (the evaluation is an all-to-all dependency cycle
//        S -> U -> V -> G -> S), and
//   (H1) no reader can see the old contents again: between the reset and the reader's first poll of that slot (one
//        evaluation later) the reader has consumed -- directly or through another workgroup -- a payload that the same producer
//        wave stored AFTER the reset and after an intervening `s_waitcnt vmcnt(0)` (vmcnt retires loads and stores in issue order).
//   Placement: U, TH and V of evaluation e-1 are reset inside P2 of evaluation e (S(e) is staged: every owner has consumed G(e-1),
//   so every role B has finished P4(e-1); P2's vmcnt(0) in front of the V store lies between); G(e-1) inside P3 of evaluation e (V(e)
//   is complete: every owner published S(e) after reading G(e-1); the tanh(o) wait lies between the reset and the G payload); S(e-1)
//   with the payload S(e) (same wave, other parity; its readers consume U / V / G first); the cost scalars QW(e-2) inside P3 of
//   evaluation e: the owner integrates the costs of evaluation e-1 behind P2 of evaluation e -- possibly after V(e) is complete --
//   so a slot is only reset once S(e) proves that its reader is done, and there are three of them (e mod 3).
// A group whose 16 workgroups report the same XCC id keeps payloads in that XCD's L2 (plain stores); any other group
// writes through (sc1).  Placement-independent; workgroups of a group share blockIdx % 8, and the CU census at kernel start makes
// the two workgroups of a CU role A and role B of the SAME member (speed only).
//
// Round 5: the kernel is a template over the group geometry and the width (DuoCfg<G, KBM> below): 8 members of 64 hidden units as described
// here (the default, and the adjoint's), 16 members of 32 with the contraction of P1 / P2 / P3 split over wave pairs for batches of up to 16
// tiles (the 4- and 8-GPU shards of n_train = 1024: it uses the CUs the default form leaves idle), and 4 members of 64 for 256-wide networks.
// Waiting waves sleep through most of a long, periodic wait instead of polling (DCtx: predictive waiting, one-tile groups).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "nocf_duo.h"

#define DU_G 8                     // members of a group: the default form (and the adjoint's): 64 hidden units per member
#define DU_GMAX 16                 // ... and the fine form of the forward (round 5): 16 members of 32 hidden units, see DuoCfg
#define DU_KBM 32                  // 16-wide k-blocks of the hidden width (m = 512)
#define DU_KBD 10                  // 16-wide k-blocks of d+1 (<= 160)
#define DU_DP 160                  // padded d+1
#define DU_R 8                     // B fragments in flight per wave (1 KiB each)
#define DU_NTMAX 4                 // tiles of 16 samples per group (LDS carve of role A)
#define DU_SENT 0xFFFFFFFFu
enum { DUK_S = 1, DUK_U = 2, DUK_T = 3, DUK_V = 4, DUK_G = 5, DUK_Q = 6, DUK_P = 7, DUK_XCC = 8 };

// LDS carve (float offsets).  Role A: A, c, the member's slices of b0 / b1 / w, the staged S and U tiles, per-sample owner state.
#define DA_A 0                                     // [16][160], rows >= r and columns >= d+1 are 0
#define DA_CW (DA_A + 16 * DU_DP)                  // [160]
#define DA_VEC (DA_CW + DU_DP)                     // b0 | b1 | w, 64 each
#define DA_SF (DA_VEC + 192)                       // staged stage-state tile: [10 kb][64 lanes][4]
#define DA_UF (DA_SF + DU_KBD * 256)               // staged u0 tile: [32 kb][64][4]
#define DA_T (DA_UF + DU_KBM * 256)                // per own sample (2 per tile): the block below
#define DS_XS 0                                    // [160] stage state (entry d = time)
#define DS_Z0 160                                  // [160] state at the start of the step
#define DS_ZA 320                                  // [160] RK accumulator
#define DS_AZC 480                                 // [160] A^T (A s) + c of the current stage state
#define DS_ZQ 640                                  // [16]  z = A s
#define DS_CZ 656                                  // [8]   cost integrals L, HJt, Q, W: value, RK accumulator
#define DS_PHX 664                                 // [2]   c.s and 1/2 |A s|^2 (final time)
#define DS_OC 668                                  // [4]   the owner's step -> its cost part: sum p^2, dPhi/dt, the x-only terms q, w
#define DS_STRIDE 672
// Role B: P4's A-operand image, w slice, the staged v tile, the y fragments (also: the positions of the cost pass), cost partials
#define DB_K4 0                                    // [10 mt][4 kb][64][4]
#define DB_VEC (DB_K4 + DU_KBD * 4 * 256)          // w, 64
#define DB_VF (DB_VEC + 64)                        // staged v tile: [32 kb][64][4]
#define DB_YF (DB_VF + DU_KBM * 256)               // [4 waves][64][4]
#define DB_XA DB_VF                                // [2 samples][2 N <= 128 entries][4]: the own samples' positions of the cost pass.  Aliases the head of the
                                                   // staged v tile: written at a tile's entry (every wave is past the previous tile's y barrier, i.e. past its
                                                   // P3 reads of VF), read before the barrier in front of this tile's V gather
#define DB_XP (DB_YF + 1024)                       // [4 waves][2]
#define DB_END (DB_XP + 16)

// The group geometry (round 5).  G members share the 512 hidden units: HPM = 512 / G per member = MTM feature tiles of 16.  A workgroup has
// four waves: with G = 8 every wave owns one feature tile and the whole contraction range (the form of rounds 3-4); with G = 16 a member
// has two feature tiles and the waves (mt, kh) = (wave & 1, wave >> 1) split the CONTRACTION of P1 / P2 / P3 in two halves whose partial
// sums meet in LDS (fixed order: lower half + upper half) -- half the resident slice (64 AccVGPRs), half the MFMAs per product and wave,
// twice the workgroups per tile.  Where a batch has fewer tiles than the chip has CU pairs (n <= 256 rows: the 4- and 8-GPU shards of
// n_train = 1024) this is what uses the idle half of the chip; with several tiles per group it shortens every link of the dependency
// chain that the two roles' tiles interleave on.  An own sample belongs to one member: SPM = 16 / G per member and tile.
// Width (round 5): KBM = m / 16.  m = 512 (KBM = 32) runs with 8 or 16 members; m = 256 (KBM = 16) with FOUR members of 64 hidden units --
// the default form's member (one feature tile and the whole contraction per wave, 64 AccVGPRs of resident weights instead of 128), 8 workgroups
// per tile, four own samples per member and tile (every wave owns one).  Evaluation, intermediates and the recording forward with the activation
// record; the adjoint of such networks is the per-tile one (it loads the record).
template <int G_, int KBM_ = DU_KBM> struct DuoCfg {
    static constexpr int G = G_;
    static constexpr int KBM = KBM_;                // 16-wide k-blocks of the hidden width
    static constexpr int M = 16 * KBM_;             // hidden width
    static constexpr int HPM = M / G_;              // hidden units per member
    static constexpr int MTM = HPM / 16;            // feature tiles per member
    static constexpr int KS = 4 / MTM;              // waves that share a feature tile (each takes 1 / KS of the contraction)
    static constexpr int SPM = 16 / G_;             // own samples per member and tile
    static constexpr int KBW = KBM_ / KS;           // k-blocks of the hidden width per wave (P2, P3)
    static constexpr int KB1 = DU_KBD / KS;         // k-blocks of d + 1 per wave (P1)
    static constexpr int WPG = 2 * G_;              // workgroups per group
    static_assert((KBM_ == DU_KBM && (G_ == 8 || G_ == 16)) || (KBM_ == 16 && G_ == 4), "members per group / width");
};

struct DuoPlan {
    int d, D1, r, nAg, NT, ngroups, spin_max, fast, G, KBM, mReal;     // mReal: the network's hidden width; 16 KBM: the width the kernel runs (zero-padded)
    float hN, cb;
    int mapmode, ldsFloats;
    int dbg, dw;                   // dw: the adjoint with the two weight-gradient roles (32 workgroups per group)
    long oW2, oW3, oK1, oK4;        // float4 offsets of the images in the workspace
    long oA, oVec, oCW;             // float offsets: A [16][160], b0 | b1 | w [3][512], c.weight [160]
    long oPlan, oErr, oXcc, oCen, oX, xStride;
};
static_assert(sizeof(DuoPlan) % 4 == 0 && sizeof(DuoPlan) / 4 <= 256, "plan copy is done by one 256-thread block");

// images (as in round 2's slab kernel):
//   W2[c][w][kb][lane] = K1[64c+16w + lane%16][16kb + 4(lane/16) + q]          (P2: out i, contraction j)
//   W3[c][w][kb][lane] = K1[16kb + 4(lane/16) + q][64c+16w + lane%16]          (P3: out j, contraction i)
//   K1[c][w][kb][lane] = K0[64c+16w + lane%16][16kb + 4(lane/16) + q]          (P1: out i, contraction dim)
//   K4[c][mt][kb][lane] = K0[64c + 16kb + 4(lane/16) + q][16mt + lane%16]      (P4: out dim, contraction i in H_c)
// ... and (one launch instead of four) clears the error words (first chunk of a call), the XCC table and the CU census, and fills the
// exchange area with the sentinel
__global__ void duo_pack_kernel(DuoPlan dp, DevPhi P, float* __restrict__ ws, int clear_err) {
    float4* ws4 = reinterpret_cast<float4*>(ws);
    {
        const long stride_ = (long)gridDim.x * blockDim.x, gid_ = (long)blockIdx.x * blockDim.x + threadIdx.x;
        unsigned* wu = reinterpret_cast<unsigned*>(ws);
        if (clear_err) for (long i = gid_; i < 64; i += stride_) wu[dp.oErr + i] = 0u;
        for (long i = gid_; i < 32 * 16 + 8 * 520 + 32 * 32; i += stride_) wu[dp.oXcc + i] = 0u;      // XCC table, CU census, progress counters
        uint4* x4 = reinterpret_cast<uint4*>(wu + dp.oX);                       // (oX and xStride are multiples of 64 words)
        const long nx4 = (long)dp.ngroups * dp.xStride / 4;
        const uint4 sen = make_uint4(DU_SENT, DU_SENT, DU_SENT, DU_SENT);
        for (long i = gid_; i < nx4; i += stride_) x4[i] = sen;
    }
    // (a network of 129 ... 511 hidden units runs zero-padded to 256 / 512: a padded unit has K0 row, K1 row and column, b0, b1 and w all zero, so
    // tanh(o) = 0 and v = 0 for it and it adds exact zeros to every sum it enters -- sigma(0) = log 2 is multiplied by a zero column of K1)
    const int m = 16 * dp.KBM, mr = dp.mReal, D1 = dp.D1;
    // (the geometry of DuoCfg at run time: one pack kernel for all forms; every image has the same size in both forms of a width)
    const int G = dp.G, HPM = m / G, MTM = HPM / 16, KS = 4 / MTM, KBW = dp.KBM / KS, KB1 = DU_KBD / KS;
    const long nW = (long)G * 4 * KBW * 64, nK1 = (long)G * 4 * KB1 * 64, nK4 = (long)G * DU_KBD * MTM * 64;
    const long total = 2 * nW + nK1 + nK4;
    const long stride = (long)gridDim.x * blockDim.x;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long idx = gid; idx < total; idx += stride) {
        float v[4];
        long dst;
        if (idx < 2 * nW) {
            const bool third = idx >= nW;
            long q = third ? idx - nW : idx;
            const int lane = q & 63; q >>= 6;
            const int kb = q % KBW; q /= KBW;                // q = c*4 + w
            const int c = (int)q >> 2, w = (int)q & 3;
            const int o = HPM * c + 16 * (w % MTM) + (lane & 15);
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * ((w / MTM) * KBW + kb) + 4 * (lane >> 4) + e;
                v[e] = (k < mr && o < mr) ? (third ? P.K[(long)k * mr + o] : P.K[(long)o * mr + k]) : 0.f;
            }
            dst = (third ? dp.oW3 : dp.oW2) + (third ? idx - nW : idx);
        } else if (idx < 2 * nW + nK1) {
            long q = idx - 2 * nW;
            const int lane = q & 63; q >>= 6;
            const int kb = q % KB1; q /= KB1;
            const int c = (int)q >> 2, w = (int)q & 3;
            const int o = HPM * c + 16 * (w % MTM) + (lane & 15);
            for (int e = 0; e < 4; ++e) {
                const int k = 16 * ((w / MTM) * KB1 + kb) + 4 * (lane >> 4) + e;
                v[e] = (k < D1 && o < mr) ? P.K0[(long)o * D1 + k] : 0.f;
            }
            dst = dp.oK1 + (idx - 2 * nW);
        } else {
            long q = idx - 2 * nW - nK1;
            const int lane = q & 63; q >>= 6;
            const int kb = q % MTM; q /= MTM;
            const int mt = q % DU_KBD; const int cmem = q / DU_KBD;
            const int dim = 16 * mt + (lane & 15);
            for (int e = 0; e < 4; ++e) {
                const int i = HPM * cmem + 16 * kb + 4 * (lane >> 4) + e;
                v[e] = (dim < D1 && i < mr) ? P.K0[(long)i * D1 + dim] : 0.f;
            }
            dst = dp.oK4 + (idx - 2 * nW - nK1);
        }
        ws4[dst] = make_float4(v[0], v[1], v[2], v[3]);
    }
    for (long i = gid; i < 16 * DU_DP; i += stride) {
        const int row = (int)(i / DU_DP), col = (int)(i % DU_DP);
        ws[dp.oA + i] = (row < dp.r && col < D1) ? P.A[(long)row * D1 + col] : 0.f;
    }
    for (long i = gid; i < 3 * m; i += stride) {
        const int which = (int)(i / m), col = (int)(i % m);
        ws[dp.oVec + i] = col < mr ? (which == 0 ? P.b0[col] : (which == 1 ? P.b[col] : P.w[col])) : 0.f;
    }
    for (long i = gid; i < DU_DP; i += stride) ws[dp.oCW + i] = (i < D1) ? P.cw[i] : 0.f;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DuoPlan) / 4) {
        const unsigned* src = reinterpret_cast<const unsigned*>(&dp);
        unsigned v = src[threadIdx.x];
        if (P.cbp && threadIdx.x == offsetof(DuoPlan, cb) / 4) v = __float_as_uint(*P.cbp);
        reinterpret_cast<unsigned*>(ws + dp.oPlan)[threadIdx.x] = v;
    }
}

// ---- the written-out MFMAs.  The resident weight slice lives in AccVGPRs and is named as SrcA directly (left to itself
// the compiler copies each operand down with v_accvgpr_read_b32 first: 46..53 cycles per MFMA instead of 32).  What an asm
// statement does NOT get from the compiler (cdna_hip_programming.md 5.7): wait states and ordering against register-only
// instructions.  So every string opens with the 2 states a VGPR operand written by a VALU instruction just before needs
// (free beside a 32-cycle MFMA), an accumulation chain starts with SrcC = 0 (no zero-initialised registers to race with),
// and DU_FENCE names the accumulators as "+v" operands: their first VALU reader cannot be scheduled above its wait states.
// (Round 2's operand-less fence is what made that kernel's results depend on the register allocation: see nocf_slab.inc.)
// tools/mfma_hazard_check.py checks the shipped ISA for both hazards.
__device__ __forceinline__ void mfma_a0(f32x4& acc, float w_acc, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc) : "a"(w_acc), "v"(b));
}
__device__ __forceinline__ void mfma_a(f32x4& acc, float w_acc, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w_acc), "v"(b));
}
__device__ __forceinline__ void mfma_v0(f32x4& acc, float w, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, float w, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
}
#define DU_FENCE2(a, b) asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a), "+v"(b))
#define DU_FENCE4(a, b, c, d) asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
// Store-data hazard (found in round 4, tools/store_hazard_check.py): a 128-bit buffer store reads its data registers over several cycles;
// hipcc's hazard recognizer assumes that a store with an SGPR soffset needs no wait state before a VALU instruction overwrites them, and
// schedules such a write into the very next slot -- on gfx950 the last lanes of every 16 then store the NEW value (the adjoint's obar
// stream came out with the y stream's fourth component in lanes 12-15 of every row).  The guard keeps the data live for two more states.
#define DU_STORE_GUARD(u) asm volatile("s_nop 1" :: "v"(u))
// Cache policy of the training traffic (aux of the raw buffer builtins: 2 = nt): the activation record (2.7 GB per recording forward at n = 1024), the
// adjoint's row streams (3.4 GB) and its tape loads (2.7 GB, each read once) are non-temporal -- they pass through the L2 the exchange lives in
// instead of displacing it.  Measured (round 5, -D...=0 against =2): recording forward 5.66 -> 5.52 ms, adjoint 6.50 -> 6.40 ms.
#ifndef DU_XST_AUX
#define DU_XST_AUX 0               // the exchange's stores between groups of one XCD (experiments: 2 = nt, 1 = sc0)
#endif
#ifndef DU_STREAM_AUX
#define DU_STREAM_AUX 2            // the adjoint's row-stream stores
#endif
#ifndef DU_TAPE_LD_AUX
#define DU_TAPE_LD_AUX 2           // the adjoint's tape loads
#endif
#ifndef DU_REC_AUX
#define DU_REC_AUX 2               // the activation record's stores
#endif
#ifndef DU_SLOWNAP
#define DU_SLOWNAP 8               // 64-clock quanta between failed polls when a group has several tiles (du_spin)
#endif
#define DU_PIN(v) asm volatile("" : "+s"(v))

struct DCtx {
    __amdgpu_buffer_rsrc_t xrs;    // this group's exchange area
    unsigned* err;
    int fast, spin_max;
    bool slow;                     // several tiles per group: longer naps between failed polls (see du_spin)
    bool dead;                     // this wave has seen the error word set (or timed out itself): it no longer waits
    // Predictive waiting (round 5).  A wave that polls is not free: its requests fill the CU's vector-memory queue and its sentinel tests are
    // VALU instructions that a co-resident wave's MFMA stream has to make room for (measured with the two roles of a member on one CU:
    // role B's P4 takes 5.6 k cycles instead of 2.3 k while the role-A waves of its CU poll for S and G -- its three stores queue behind their
    // requests) -- and the long waits are PERIODIC: role B waits for v through all of role A's step, P1 and P2, an owner for the partial gradients
    // through P3 and P4, every evaluation alike.  So a wave measures how long a payload took to become valid COUNTED FROM ITS OWN LAST OUTPUT
    // for that tile (role A: the end of P2, role B: the end of P4 -- the `anchor`), keeps the last two such intervals per exchange kind and tile
    // in a wave-private LDS slot, and sleeps until anchor + 3/4 of the shorter one - `lead` (~1 000 clocks) before its first poll.  Sleeping costs
    // no issue slot and no request; the poll loop behind it is unchanged, so a wrong prediction is only ever a nap that ends early or a little
    // late, never a missed payload.  What this does NOT use, with reasons measured this round: the time since the wave ENTERED the wait (the entry
    // jitters by thousands of clocks with the co-resident role's work: the owner waves overslept, n = 512 went from 3.36 to 3.56 ms), and the
    // absolute period of the payload's arrival (a late riser delays everybody behind it on the dependency cycle, which lengthens the very period
    // the next prediction is made from; four waits per cycle feed that loop and it ran away: evaluations of 300 k cycles).  The interval from the
    // wave's own output to the payload contains no waiting of this wave, so an overslept nap inflates one measurement and the next one is
    // true again (and the minimum of the last two is what is used).
    int udelay;                    // naps (64 clocks each) in front of the FIRST poll for U: see the U gather
    int pw;                        // LDS float index of this wave's words: [3 kinds][4 tiles][2] intervals, then [4 tiles] anchors; -1: no prediction
    unsigned qs, lead;             // sleep quanta (64 clocks) per clock tick, 16.16 fixed point; the lead in ticks
    int pwsh;                      // the nap ends at anchor + d - (d >> pwsh) - lead
};
enum { DPW_S = 0, DPW_G = 1, DPW_V = 2, DPW_ANCHOR = 24, DPW_WORDS = 32 };
__device__ __forceinline__ unsigned du_clock() { return (unsigned)__builtin_amdgcn_s_memtime(); }
// (pwbase: LDS float index of the workgroup's 4 x DPW_WORDS words, or -1; every wave clears its own words)
__device__ __forceinline__ void du_calibrate(DCtx& g, int pwbase, int wave, int lane) {
    g.pw = -1; g.qs = 0u; g.lead = 0u;
    if (pwbase < 0) return;
    g.pw = pwbase + wave * DPW_WORDS;
    if (lane < DPW_WORDS) lds[g.pw + lane] = 0.f;
    const unsigned t0 = du_clock();
    __builtin_amdgcn_s_sleep(127);
    const unsigned dt = du_clock() - t0;                       // ticks per 127 x 64 clocks
    g.qs = dt ? (127u << 16) / dt : 0u;
    g.lead = dt >> 3;                                          // ~1 000 clocks
    if (!g.qs || !g.lead) g.pw = -1;
}
// this wave has just published its output for tile t (or passed the point that stands for it): the intervals of its next waits count from here
__device__ __forceinline__ void du_anchor(DCtx& g, int t, int lane) {
    if (g.pw < 0) return;
    const unsigned now = du_clock();
    if (lane == 0) lds[g.pw + DPW_ANCHOR + t] = __uint_as_float(now);
}
struct DWait { unsigned anchor, d1; int at; };
__device__ __forceinline__ DWait du_wait_begin(DCtx& g, int kind, int t) {
    DWait w; w.anchor = 0u; w.d1 = 0u; w.at = -1;
    if (g.pw < 0 || kind < 0) return w;
    w.at = g.pw + 2 * (kind * 4 + t);
    w.anchor = __builtin_amdgcn_readfirstlane(__float_as_uint(lds[g.pw + DPW_ANCHOR + t]));
    w.d1 = __builtin_amdgcn_readfirstlane(__float_as_uint(lds[w.at]));
    const unsigned d2 = __builtin_amdgcn_readfirstlane(__float_as_uint(lds[w.at + 1]));
    const unsigned d = w.d1 < d2 ? w.d1 : d2;
    if (w.anchor != 0u && d >= 4u * g.lead) {                  // (shorter intervals are hops: not predictable to a poll's length)
        const int ahead = (int)(w.anchor + d - (d >> g.pwsh) - g.lead - du_clock());   // ticks until the nap should end
        if (ahead > 0 && (unsigned)ahead < d) {
            const unsigned q = (unsigned)(((unsigned long long)(unsigned)ahead * g.qs) >> 16);
            for (unsigned i = 0; i < (q >> 2); ++i) __builtin_amdgcn_s_sleep(4);
        }
    }
    return w;
}
__device__ __forceinline__ void du_wait_end(DCtx& g, const DWait& w, int lane) {
    if (w.at < 0) return;
    const unsigned now = du_clock();
    if (lane == 0) { lds[w.at] = __uint_as_float(w.anchor ? now - w.anchor : 0u); lds[w.at + 1] = __uint_as_float(w.d1); }
}

__device__ __forceinline__ u32x4 du_ld(const DCtx& g, int vbyte, int sbyte) {
    return __builtin_amdgcn_raw_buffer_load_b128(g.xrs, vbyte, sbyte, 16 /*sc1: not through the CU's L1*/);
}
__device__ __forceinline__ void du_st(const DCtx& g, int vbyte, int sbyte, f32x4 v) {
    u32x4 u;
    u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]); u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
    // (round 6: the guard sits in the write-through branch only -- the branch where tools/store_hazard_check.py finds overwritten store data
    // without it, 24 places: the U store followed by the tanh's v_rcp.  A guard behind EVERY store pinned the stores between the written-out MFMAs
    // and cost 0.5 %; build() runs the checker over the ISA of every build and refuses a library with a finding)
    if (g.fast) __builtin_amdgcn_raw_buffer_store_b128(u, g.xrs, vbyte, sbyte, DU_XST_AUX /*stays in the XCD's L2*/);
    else {
        // (two wait states directly behind the store, fenced against the scheduler on both sides: an asm that merely NAMES the data as an input
        // lets the compiler rebuild a constant payload -- the sentinel -- in registers that overlap the store's, in between)
        __builtin_amdgcn_raw_buffer_store_b128(u, g.xrs, vbyte, sbyte, 16 /*sc1: write-through*/);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void du_st_sent(const DCtx& g, int vbyte, int sbyte) {
    const f32x4 sv = {__uint_as_float(DU_SENT), __uint_as_float(DU_SENT), __uint_as_float(DU_SENT), __uint_as_float(DU_SENT)};
    du_st(g, vbyte, sbyte, sv);
}
__device__ __forceinline__ bool du_bad(const u32x4& v) {
    const unsigned a = v.x > v.y ? v.x : v.y, b = v.z > v.w ? v.z : v.w;
    return (a > b ? a : b) == DU_SENT;
}
__device__ __forceinline__ f32x4 du_f(const u32x4& v) {
    return (f32x4){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
// one more failed poll: back off; true when the caller should stop waiting (timeout, or somebody else already gave up)
__device__ __forceinline__ bool du_spin(DCtx& g, int& spins, unsigned what) {
    if (g.dead) return true;
    // (with one tile per group an evaluation is one dependent chain and a short nap finds the data sooner; with two the other tile's
    // work covers the wait and fewer polls leave more of L2 to it: 5.03 -> 4.97 ms at n = 1024, 2.85 -> 2.83 at n = 256)
    if (g.slow) __builtin_amdgcn_s_sleep(DU_SLOWNAP); else __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 63) == 0) {
        const unsigned ev = __builtin_amdgcn_readfirstlane(__hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (ev != 0u) { g.dead = true; return true; }
    }
    if (spins > g.spin_max) { atomicExch(g.err, 0x3000u + what); g.dead = true; return true; }
    return false;
}

// My share of an exchange buffer of NF fragments -> LDS: wave w takes the fragments f = w, w+4, ... (NFW = ceil(NF / 4) of them,
// all requested together: one L2 round trip), repeats the requests while any of them shows a sentinel word (the data IS the
// signal), then writes them to the staging buffer at LDS float4 index l4.  The four waves of a workgroup share the staged tile:
// a fragment crosses L2 -> CU once per workgroup (streamed by every wave for itself -- the first form of this kernel -- the
// 32 waves of a group pulled 4 MB per GEMM and XCD through L2 and every stream ran at the L2 latency: 190 cycles per k-block).
template <int NF>
__device__ __forceinline__ void du_gather(DCtx& g, int wave, int lane, int sbyte, int l4, unsigned what, int pkind = -1, int ptile = 0) {
    constexpr int NFW = (NF + 3) / 4;
    u32x4 v[NFW];
    int spins = 0;
    const DWait tw = du_wait_begin(g, pkind, ptile);
    while (true) {
#pragma unroll
        for (int u = 0; u < NFW; ++u) if (NF % 4 == 0 || wave + 4 * u < NF) v[u] = du_ld(g, lane * 16, sbyte + (wave + 4 * u) * 1024);
        bool bad = false;
#pragma unroll
        for (int u = 0; u < NFW; ++u) if (NF % 4 == 0 || wave + 4 * u < NF) bad |= du_bad(v[u]);
        if (!__any(bad)) break;
        if (du_spin(g, spins, what)) break;
    }
    du_wait_end(g, tw, lane);
    float4* L4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int u = 0; u < NFW; ++u)
        if (NF % 4 == 0 || wave + 4 * u < NF) { const f32x4 f = du_f(v[u]); L4[l4 + (wave + 4 * u) * 64 + lane] = make_float4(f[0], f[1], f[2], f[3]); }
}

// The same with the NF fragments split over TWO waves (which = 0 / 1 takes the even / odd ones).
template <int NF>
__device__ __forceinline__ void du_gather2(DCtx& g, int which, int lane, int sbyte, int l4, unsigned what, int pkind = -1, int ptile = 0) {
    constexpr int NFW = (NF + 1) / 2;
    u32x4 v[NFW];
    int spins = 0;
    const DWait tw = du_wait_begin(g, pkind, ptile);
    while (true) {
#pragma unroll
        for (int u = 0; u < NFW; ++u) if (NF % 2 == 0 || which + 2 * u < NF) v[u] = du_ld(g, lane * 16, sbyte + (which + 2 * u) * 1024);
        bool bad = false;
#pragma unroll
        for (int u = 0; u < NFW; ++u) if (NF % 2 == 0 || which + 2 * u < NF) bad |= du_bad(v[u]);
        if (!__any(bad)) break;
        if (du_spin(g, spins, what)) break;
    }
    du_wait_end(g, tw, lane);
    float4* L4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int u = 0; u < NFW; ++u)
        if (NF % 2 == 0 || which + 2 * u < NF) { const f32x4 f = du_f(v[u]); L4[l4 + (which + 2 * u) * 64 + lane] = make_float4(f[0], f[1], f[2], f[3]); }
}

// acc = sum over NKB k-blocks of W[kb] (A operand: AccVGPRs if ACC, else VGPRs) x the staged fragments at LDS float4 index b4 + kb*64 (B operand).
// Two accumulation chains (a dependent v_mfma_f32_16x16x4_f32 may issue 40 cycles behind its producer; the chains alternate
// at 32), B fragments read 3 k-blocks (12 MFMAs = 384 cycles) ahead of their use.
struct DuNoMid { __device__ __forceinline__ void operator()(int) const {} };
// mid(kb): called behind every k-block's MFMAs (kb is a compile-time constant after unrolling) -- stores / loads that only have to be
// ISSUED before the product ends (slot resets behind the second k-block, the epilogue's LDS operands two k-blocks before the end) go
// there: they issue in the shadow of the MFMAs instead of in front of the first one or behind the last
template <int NKB, bool ACC, typename Mid = DuNoMid>
__device__ __forceinline__ f32x4 du_gemm_lds(const f32x4 (&W)[NKB], int b4, Mid mid = Mid()) {
    const float4* L4 = reinterpret_cast<const float4*>(lds);
    float4 ring[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) ring[i] = L4[b4 + (i < NKB ? i : 0) * 64];
    f32x4 a0, a1;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 3 < NKB) ring[(kb + 3) & 3] = L4[b4 + (kb + 3) * 64];
        const float4 b = ring[kb & 3];
        if (ACC) {
            if (kb == 0) { mfma_a0(a0, W[0][0], b.x); mfma_a0(a1, W[0][1], b.y); }
            else { mfma_a(a0, W[kb][0], b.x); mfma_a(a1, W[kb][1], b.y); }
            mfma_a(a0, W[kb][2], b.z); mfma_a(a1, W[kb][3], b.w);
        } else {
            if (kb == 0) { mfma_v0(a0, W[0][0], b.x); mfma_v0(a1, W[0][1], b.y); }
            else { mfma_v(a0, W[kb][0], b.x); mfma_v(a1, W[kb][1], b.y); }
            mfma_v(a0, W[kb][2], b.z); mfma_v(a1, W[kb][3], b.w);
        }
        mid(kb);
    }
    DU_FENCE2(a0, a1);
    return a0 + a1;
}

// (Tried in round 5 and taken out: the fine form's P2 / P3 with the wave's sixteen fragments taken STRAIGHT from L2 into registers -- no staging
// buffer, no barrier, the product starting on the first four while the other twelve arrive.  n = 128: 2.47 ms against 2.35 staged, n = 256 equal:
// the twelve follow their request by a full round trip, which the first 16 MFMAs do not cover, and every fragment crosses L2 -> CU twice.)

// Diagnostic build only (-DNOCF_STAMPS): per-wave shader-clock timeline of ONE evaluation of group 0 / member 0 (both roles):
// tools/duo_timeline.py.  The production library contains no stamp.
#ifdef NOCF_STAMPS
// (100 MHz, one clock for the whole chip: s_memtime differs between workgroups.  The asm is volatile with a memory clobber so that
// a stamp stays where it is written relative to loads, stores and other stamps.)
__device__ __forceinline__ unsigned long long du_now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define DTL(id) do { if (tlp) { const unsigned long long t_ = du_now(); if (lane == 0) tlp[id] = t_; } } while (0)
#elif defined(NOCF_ACC)
#define DTLB(id) DTL(id)                            /* (points that exist in the lap-counter build only: in front of the gathers' barriers) */
// Second diagnostic build (-DNOCF_ACC, tools/duo_acc.py): the same points as LAP counters -- the shader clocks since the wave's previous point are
// added to the point's bucket (32 per wave, in LDS), over the WHOLE rollout, and written out at the end: where every wave's time goes (waits for
// each exchange kind, barriers, products, epilogues), for every wave of the launch.  ~40 clocks per point (MI355X_MICROARCH.md).
__device__ __forceinline__ unsigned long long du_acc_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define DTL(id) do { const unsigned long long n_ = du_acc_now(); \
                     if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(lds) + acc_base_ + ((id) % 40), (unsigned)(n_ - acc_t_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                     acc_t_ = n_; } while (0)
#else
#define DTL(id) do { } while (0)
#endif
#ifndef DTLB
#define DTLB(id) do { } while (0)
#endif

// ---- x-only running-cost terms (Cross2D.py:89-160, SwarmTraj.py:89-162): cyclic pairing, agent a meets (a+j) mod N
__device__ __forceinline__ float du_pair_term(float s2, float thr, float thr2, float nden_l2e) {
    // branch-free: v_sqrt_f32 / v_exp_f32 (1 ulp each); the mask is the reference's `dist < thr`, entries equal to 1 are dropped
    const float dist = __builtin_amdgcn_sqrtf(s2);
    const float e = __builtin_amdgcn_exp2f((dist * dist) * nden_l2e);
    const bool keep = (s2 < thr2) & (dist < thr) & (e != 1.f);
    return keep ? e : 0.f;
}
struct DXPar { float thr, thr2, nden_l2e, den, thr_pair2; int N, J, JJ, Jh; bool obs, wantW; };
__device__ __forceinline__ DXPar du_x_params(const DevProb& pb, int PD) {
    DXPar xp;
    xp.N = pb.nAgents;
    xp.obs = pb.obstacle != NOCF_OBS_NONE && (PD == 2 || pb.alphQ > 0.0);
    xp.wantW = want_W(pb) && xp.N >= 2;
    xp.den = (float)(2.0 * pb.r * pb.r);
    const double fac = pb.training ? (PD == 3 ? 3.2 : 2.2) : 2.0;
    xp.thr = (float)(fac * pb.r);
    xp.thr2 = xp.thr * xp.thr * 1.000002f;
    xp.nden_l2e = -1.4426950408889634f / xp.den;
    xp.thr_pair2 = (float)((pb.training ? 2.2 : 2.0) * pb.r);        // the two-agent form of calcW
    xp.J = (xp.N - 1) >> 1;                                           // partners every agent meets: (a + j) mod N, j = 1..J
    xp.JJ = (xp.wantW && xp.N > 2) ? xp.J + (((xp.N & 1) == 0) ? 1 : 0) : 0;   // even N: also the opposite agent j = N/2, from the lower half only
    xp.Jh = (xp.JJ + 1) >> 1;                                         // the two halves of the partner range: [1, Jh], (Jh, JJ]
    return xp;
}
// One wave = one own sample x one half of the partner range; lane a < N is agent a.  The sample's positions sit in LDS as
// [2 N entries][4 floats] (entry i and i + N are the same agent, the fourth float is padding): the partner (a + j) mod N is entry
// a + j -- no wrap, one ds_read_b128 per partner, consecutive lanes read consecutive entries.  Half 0 adds the obstacle terms.
// Straight-line selects, no per-pair branches; every unordered pair is counted once.  Returns this lane's partial sums.
// (NP: the partner range in NP parts, `half` = 0 .. NP - 1: two waves per own sample in the default form, four in the fine form -- one own
// sample per member --, one with four own samples per member, m = 256)
template <int PD, int NP = 2>
__device__ __forceinline__ void du_x_wave(const DevProb& pb, const DXPar& xp, int x4 /* LDS float4 index of the sample's entries */, int lane, int half,
                                          float& qacc, float& wacc) {
    const int N = xp.N;
    const float4* L4 = reinterpret_cast<const float4*>(lds);
    const bool act = lane < N;
    const int a = act ? lane : 0;
    const float4 me = L4[x4 + a];
    if (half == 0 && xp.obs) {
        const float ob = (PD == 2) ? obstacle_cross2d(pb, me.x, me.y) : obstacle_swarm(pb, me.x, me.y, me.z);
        qacc += act ? ob : 0.f;
    }
    if (!xp.wantW) return;
    if (N == 2) {
        if (half == 0 && lane == 0) {
            const float4 o = L4[x4 + 1];
            float s2 = (me.x - o.x) * (me.x - o.x); s2 += (me.y - o.y) * (me.y - o.y);
            if (PD == 3) s2 += (me.z - o.z) * (me.z - o.z);
            const float dist = sqrtf(s2);
            if (dist < xp.thr_pair2) wacc += expf(-(dist * dist) / xp.den);
        }
        return;
    }
    int jlo = half ? xp.Jh + 1 : 1, jhi = half ? xp.JJ : xp.Jh;                 // partners jlo..jhi (inclusive)
    if (NP == 4) { const int Jq = (xp.JJ + 3) >> 2; jlo = half * Jq + 1; jhi = (half + 1) * Jq < xp.JJ ? (half + 1) * Jq : xp.JJ; }
    if (NP == 1) { jlo = 1; jhi = xp.JJ; }
    const bool actlow = act & (a < (N >> 1));                                   // the opposite agent (even N) counts from the lower half only
    constexpr int XW = 5;                                   // partners per round (their LDS reads are in flight together)
    for (int j = jlo; j <= jhi; j += XW) {
        float4 p[XW];
#pragma unroll
        for (int u = 0; u < XW; ++u) p[u] = L4[x4 + a + j + u];
        float s2[XW];
#pragma unroll
        for (int u = 0; u < XW; ++u) {
            const float e0 = me.x - p[u].x, e1 = me.y - p[u].y;
            float a2 = e0 * e0; a2 += e1 * e1;
            if (PD == 3) { const float e2 = me.z - p[u].z; a2 += e2 * e2; }
            const bool valid = (j + u <= jhi) & ((j + u <= xp.J) ? act : actlow);
            s2[u] = valid ? a2 : 3.0e38f;
        }
        bool near = false;
#pragma unroll
        for (int u = 0; u < XW; ++u) near |= s2[u] < xp.thr2;
        // pairs within the interaction radius are rare: the square roots and exponentials are skipped for the whole wave unless
        // one of its pairs of this round is near -- every skipped term is exactly 0
        if (__any(near)) {
#pragma unroll
            for (int u = 0; u < XW; ++u) wacc += du_pair_term(s2[u], xp.thr, xp.thr2, xp.nden_l2e);
        }
    }
}

// byte offsets of the exchange kinds inside a group's area (functions of NT), all pinned in scalar registers by the kernel
struct DXOff { int S, U, T, V, G, Q, P; };
// (bwd: the adjoint's layout -- the Q area holds the physics term of two own samples per member, 2 x 160 floats, instead of 2 x 2 scalars)
__host__ __device__ inline long duo_x_layout(int NT, DXOff* o, bool bwd = false, int G = DU_G, int KBM = DU_KBM) {
    long x = 0;
    auto take = [&](long nfl) { const long at = x; x += (nfl + 63) / 64 * 64; return at; };
    const long s = take(2L * NT * DU_KBD * 256), u = take(2L * NT * KBM * 256), t = take(2L * NT * KBM * 256), v = take(2L * NT * KBM * 256);
    // (cost scalars: 2 floats per own sample = 32 per tile whatever the form; the adjoint: 160 per own sample)
    const long gg = take(2L * NT * G * DU_KBD * 256), q = take(3L * NT * 32 * (bwd ? 80 : 1)), p = take((long)NT * DU_G * 4 * 16);
    if (o) { o->S = (int)(s * 4); o->U = (int)(u * 4); o->T = (int)(t * 4); o->V = (int)(v * 4); o->G = (int)(gg * 4); o->Q = (int)(q * 4); o->P = (int)(p * 4); }
    return x;                                      // floats per group
}

struct DuoRun { long row0, n_total; };             // rows of this launch inside the caller's batch (chunked launches; sAll indexing)

// REC: the training variant also stores every stage input (RollArgs::sAll); ZF: intermediates (trajectories and controls, one more
// evaluation per step); the plain evaluation variant carries no trace of either
// MODE 1 (ONE): the launch has one tile per group in the default geometry (257 ... 512 rows): the instantiation WITH the owner's flag below.  The flag
// costs that geometry registers it does not have (11 spilled instead of 2-5: n = 1024 +2 %), and only one-tile groups gain from it
template <int PD, bool REC, bool ZF, int GM = DU_G, int KBMT = DU_KBM, int MODE = 0>
__global__ void __launch_bounds__(256, 2) rollout_duo_kernel(const DuoPlan* __restrict__ dpp, DevProb pb, float* ws, RollArgs ra, DuoRun rr) {
    typedef DuoCfg<GM, KBMT> CF;
    constexpr int G = CF::G, MTM = CF::MTM, KS = CF::KS, SPM = CF::SPM, KBW = CF::KBW, KB1 = CF::KB1, WPG = CF::WPG, HPM = CF::HPM;
    constexpr int KBM = CF::KBM, MW = CF::M;
    // LDS of the fine form: the waves' partial sums [4 waves][64 lanes][4] (role A: in front of the per-sample blocks; role B: in the half of
    // the K4 region that its half-sized image leaves free)
    constexpr int DAPX = DA_T, DAPW = DA_T + (KS > 1 ? 1024 : 0), DAFL = DAPW + 28, DAT = DAPW + 4 * DPW_WORDS, DBPX = DB_K4 + DU_KBD * MTM * 256, DBPW = DB_END;
    static_assert(KS == 1 || DBPX + 1024 <= DB_VEC, "role B's partial sums must fit behind the K4 image");
    const DuoPlan& dp = *dpp;
    const int bid = blockIdx.x;
    const int jb = bid >> 3;
    const int tid = threadIdx.x;
    // ---- who am I?  Static map: workgroups with equal blockIdx % 8 (one XCD under round-robin dispatch) form the groups.  Census map
    // (default): the two workgroups that share a CU become role A and role B of the SAME (group, member).  Their MFMA phases then
    // alternate on that CU's matrix pipes by construction, and all members of a group see the same interference, so nobody waits for a
    // straggler whose CU-mate belongs to another group (static map at 2 workgroups per CU: 29 % longer evaluations than at 1 per CU).
    // Every workgroup registers on its CU's counter (key: SE / SH / CU id of HW_REG_HW_ID); the first one per CU draws the CU's rank in
    // its XCD slot; when all workgroups of the slot have arrived and #CUs x 2 = #workgroups (every CU holds exactly two), rank r is
    // (group r / 8 of the slot, member r % 8), the first-arrived workgroup takes role A.  Otherwise the slot keeps the static map.
    // Placement decides speed only: results do not depend on which workgroup computes what.
    int group = (bid & 7) + 8 * (jb / WPG);
    int within = jb % WPG;
    int member = (dp.mapmode & 1) ? (within >> 1) : (within % G);
    int role = (dp.mapmode & 1) ? (within & 1) : (within / G);
    if (dp.mapmode & 2) {
        if (tid == 0) {
            const int xsl = bid & 7;
            const int nwg = xsl < dp.ngroups ? WPG * ((dp.ngroups - xsl + 7) / 8) : 0;      // workgroups of this slot that have work
            unsigned* cen = reinterpret_cast<unsigned*>(ws) + dp.oCen + xsl * 520;      // [0] arrived, [1] CUs seen, [8..263] workgroups per CU, [264..519] rank + 1 per CU
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            const unsigned key = (hw >> 8) & 0xffu;
            int ok = 0, rank = 0, sl = 0;
            if (group < dp.ngroups) {
                sl = (int)atomicAdd(cen + 8 + key, 1u);
                if (sl == 0) { rank = (int)atomicAdd(cen + 1, 1u); __hip_atomic_store(cen + 264 + key, (unsigned)rank + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                // (a key that receives a THIRD workgroup -- the key holds SE / SH / CU, so workgroups of one slot on two XCDs would share
                // keys -- still satisfies #CUs x 2 = #workgroups with counts like 1 and 3: it switches the census map off for the slot)
                if (sl > 1) atomicExch(cen + 2, 1u);
                atomicAdd(cen + 0, 1u);
                int spins = 0;
                while ((int)__hip_atomic_load(cen + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > dp.spin_max) { atomicExch(reinterpret_cast<unsigned*>(ws) + dp.oErr, 0x3000u + DUK_XCC); break; }
                }
                ok = spins <= dp.spin_max && (int)__hip_atomic_load(cen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 2 == nwg &&
                     __hip_atomic_load(cen + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
                if (ok && sl != 0) {
                    spins = 0;
                    unsigned r1 = 0;
                    while ((r1 = __hip_atomic_load(cen + 264 + key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > dp.spin_max) { atomicExch(reinterpret_cast<unsigned*>(ws) + dp.oErr, 0x3000u + DUK_XCC); r1 = 1u; break; }
                    }
                    rank = (int)r1 - 1;
                }
            }
            if (ok) atomicAdd(reinterpret_cast<unsigned*>(ws) + dp.oErr + 1, 1u);       // (observability: NOCF_DEBUG=2 prints how many workgroups took the census map)
            lds[0] = (float)ok; lds[1] = (float)rank; lds[2] = (float)sl;
        }
        __syncthreads();
        if (lds[0] != 0.f) {
            const int rank = (int)lds[1];
            group = (bid & 7) + 8 * (rank / G);
            member = rank % G;
            role = (int)lds[2];
            within = member * 2 + role;
        }
        __syncthreads();
    }
    if (group >= dp.ngroups) return;
    const int lane = tid & 63, slot = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ft = KS > 1 ? wave % MTM : wave, kh = KS > 1 ? wave / MTM : 0;                 // this wave's feature tile of the member, its part of the contraction
    int NT = dp.NT; DU_PIN(NT);                                 // (ONE: still read at run time -- as a compile-time 1 the kernel came out with 31 spilled registers instead of 11)
    // MODE (default geometry, m = 512): 0 = several tiles per group: neither the owner's flag nor predictive waiting is compiled in (both only
    // ever ran with one tile per group; their state -- DCtx, the anchors, the calibration -- cost the multi-tile launches 1.1 %, round 6);
    // 1 = 32 one-tile groups (every CU holds the two roles of one member): flag + prediction; 2 = other one-tile launches: prediction only
    constexpr bool ONE = MODE == 1;
    constexpr bool PRED = MODE != 0 || GM != DU_G;
    // the owner's step in two parts (own_state_split / own_book): where a group has ONE tile -- the fine geometry, and the one-tile launches of the
    // 256-wide geometry (MODE 1 there: n <= 1024) -- measured; with two tiles per group it loses in every geometry
    constexpr bool OSPLIT = GM == DU_GMAX || (GM == 4 && MODE == 1);
    constexpr bool FLAGS = ONE || GM != DU_G;                   // the local owner's "about to publish" flag for the stage-state gatherers (see own_state)
    const int d = dp.d;
    const float hN = dp.hN;
    DXOff xo;
    (void)duo_x_layout(NT, &xo, false, G, KBM);
    int xS = xo.S, xU = xo.U, xT = xo.T, xV = xo.V, xG = xo.G, xQ = xo.Q, xP = xo.P;
    DU_PIN(xS); DU_PIN(xU); DU_PIN(xT); DU_PIN(xV); DU_PIN(xG); DU_PIN(xQ); DU_PIN(xP);
    DCtx g;
    g.xrs = __builtin_amdgcn_make_buffer_rsrc(ws + dp.oX + (long)group * dp.xStride, 0, (int)(dp.xStride * 4), 0x00020000);
    g.err = reinterpret_cast<unsigned*>(ws) + dp.oErr;
    g.fast = 0; g.spin_max = dp.spin_max; g.dead = false; g.slow = NT > 1;
    g.pw = -1; g.qs = 0u; g.lead = 0u;                          // (set per role below: the slots live in that role's LDS carve)
    g.udelay = (dp.dbg >> 8) & 15;
    g.pwsh = 2 + ((dp.dbg >> 12) & 3);
    const float4* ws4 = reinterpret_cast<const float4*>(ws);
    float4* L4 = reinterpret_cast<float4*>(lds);
    const int vb = lane * 16;                                   // this lane's 16 bytes of a fragment
#ifdef NOCF_ACC
    const int acc_base_ = dp.ldsFloats - 256 + wave * 32;       // (the host reserves 256 more floats in this build)
    if (lane < 32) lds[acc_base_ + lane] = 0.f;
    unsigned long long acc_t_ = du_acc_now();
    auto acc_dump = [&]() {
        DTL(31);
        if (ra.stamps && lane < 32) ra.stamps[((((long)group * 16 + member) * 2 + role) * 4 + wave) * 32 + lane] = reinterpret_cast<const unsigned*>(lds)[acc_base_ + lane];
    };
#endif
#ifdef NOCF_STAMPS
    unsigned long long* tlp = nullptr;
    const int e_probe = (ra.nt / 2) * ((ra.stepper == NOCF_RK4) ? 4 : 1) + 2;
#define DTL_EPOCH(e_) tlp = (ra.stamps && group == 0 && member < DU_G && (e_) == e_probe) ? ra.stamps + ((member * 2 + role) * 4 + wave) * 128 : nullptr
#else
#define DTL_EPOCH(e_) do { } while (0)
#endif

    // ---- the resident weight slice: 128 AccVGPRs per lane, loaded once (role A: K1[H_c,:], role B: K1[:,H_c]^T)
    f32x4 W[KBW];
    {
        const long bw = (role ? dp.oW3 : dp.oW2) + (long)(member * 4 + wave) * KBW * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) { const float4 a = ws4[bw + kb * 64]; W[kb] = (f32x4){a.x, a.y, a.z, a.w}; }
    }
    // ---- where do the 16 workgroups of my group run?  (same-XCD groups keep the exchange in that XCD's L2)
    if (dp.fast && wave == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc = (xcc & 0xfu) + 1u;
        unsigned* tab = reinterpret_cast<unsigned*>(ws) + dp.oXcc + (long)group * WPG;
        if (lane == 0) __hip_atomic_store(tab + within, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        unsigned v = 0;
        while (true) {
            v = __hip_atomic_load(tab + (lane & (WPG - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v != 0u)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > dp.spin_max) { if (lane == 0) atomicExch(g.err, 0x3000u + DUK_XCC); break; }
        }
        if (lane == 0) lds[0] = __all(v == xcc) ? 1.f : 0.f;
    }
    __syncthreads();
    g.fast = dp.fast && lds[0] != 0.f;
    __syncthreads();

    const int nstage = (ra.stepper == NOCF_RK4) ? 4 : 1;
    const long rowg = (long)group * 16 * NT;                    // first row (of this launch) of the group
    auto own_row = [&](int t, int j) -> long { return rowg + 16 * t + SPM * member + j; };
    // The activation record's stores (training, REC): every wave writes the 4 features it holds of 16 samples, four times per tile and evaluation.
    // As 64-bit pointer arithmetic per lane that was ~10 vector instructions per store and 20 spilled registers in the recording instantiation;
    // here the evaluation's block of a section is a BUFFER whose base the scalar unit forms (section, evaluation: wave-uniform), and the lane adds a
    // 32-bit offset (row, feature) -- rows beyond the batch get an offset beyond the buffer, which drops the store.
    auto rec_block = [&](int section, long eidx) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(ra.act + (long)section * ra.actRows * MW + (eidx * rr.n_total + rr.row0) * MW, 0,
                                                 (int)((rr.n_total - rr.row0) * MW * 4), 0x00020000);
    };
    auto rec_off = [&](int t) -> int {                           // this lane's byte offset in such a block: sample lane & 15 of tile t, features HPM member + 16 ft + 4 slot
        const long rw = rowg + 16 * t + (lane & 15);
        return rw < ra.n ? (int)((rw * MW + HPM * member + 16 * ft + 4 * slot) * 4) : -1;
    };
    auto rec_row = [&](float* row, int pi_, float a, float b, float c, float d_) {      // entries pi_ .. pi_ + 3 of a row of d + 1 floats
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, (d + 1) * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), rs, 4 * pi_, 0, DU_REC_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b), rs, 4 * pi_ + 4, 0, DU_REC_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c), rs, 4 * pi_ + 8, 0, DU_REC_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d_), rs, 4 * pi_ + 12, 0, DU_REC_AUX);
    };
    auto rec_store = [&](__amdgpu_buffer_rsrc_t rs, int off, float a, float b, float c, float d_) {
        u32x4 u;
        u.x = __float_as_uint(a); u.y = __float_as_uint(b); u.z = __float_as_uint(c); u.w = __float_as_uint(d_);
        __builtin_amdgcn_raw_buffer_store_b128(u, rs, off, 0, DU_REC_AUX);
#ifndef DU_R_NOGUARD
        DU_STORE_GUARD(u);
#endif
    };

    if (role == 0) {
        // =====================================================================================================
        // role A: own (partial gradients -> RK update -> next stage state), P1, P2
        // =====================================================================================================
        // P1's A operand: this wave's 16 rows of K0, 40 VGPRs (the AccVGPR half of the 256-register budget is the K1 slice)
        f32x4 K0r[KB1];
        {
            const long b1 = dp.oK1 + (long)(member * 4 + wave) * KB1 * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) { const float4 a = ws4[b1 + kb * 64]; K0r[kb] = (f32x4){a.x, a.y, a.z, a.w}; }
        }
        for (int i = tid; i < 16 * DU_DP; i += 256) lds[DA_A + i] = ws[dp.oA + i];
        if (tid < DU_DP) lds[DA_CW + tid] = ws[dp.oCW + tid];
        if (tid < 192 && (tid & 63) < HPM) lds[DA_VEC + tid] = ws[dp.oVec + (tid >> 6) * MW + member * HPM + (tid & 63)];
        for (int i = tid; i < SPM * NT * DS_STRIDE; i += 256) lds[DAT + i] = 0.f;
        if (PRED) du_calibrate(g, ((dp.dbg & 8) || (NT > 1 && !(dp.dbg & 16))) ? -1 : DAPW, wave, lane);     // (one tile per group: measured, see DCtx; dbg bit 16 forces it on)
        __syncthreads();
        for (int i = tid; i < SPM * NT * DU_DP; i += 256) {
            const int s = i / DU_DP, c = i - s * DU_DP;
            long row = own_row(s / SPM, s % SPM); if (row >= ra.n) row = ra.n - 1;
            const float v = (c < d) ? ra.x[row * d + c] : 0.f;
            lds[DAT + s * DS_STRIDE + DS_Z0 + c] = v;
            lds[DAT + s * DS_STRIDE + DS_XS + c] = (c == d) ? (float)ra.t0 : v;
        }
        __syncthreads();
        const float cAlphQ = (float)pb.alphQ, cAlphW = (float)pb.alphW;
        const bool cWantW = want_W(pb);
        const float c16 = (float)(1.0 / 6.0), c26 = (float)(2.0 / 6.0);
        int rA = dp.r; DU_PIN(rA);
        // Owner units are SAMPLES: own sample s = SPM t + j (sample SPM member + j of tile t) belongs to wave s & 3; lane l < 40 of that
        // wave keeps the piece dims 4 l .. 4 l + 3 = (dim tile mt = l / 4, slot sl = l % 4) of the sample.
        const bool pact = lane < 40;
        const int pmt = lane >> 2, psl = lane & 3, pi = pact ? 4 * lane : 0;
        const int pd_lane = d >> 2, pd_e = d & 3;                     // where g[d] = dPhi/dt sits

        // Requests for the 8 members' partial gradients of own sample s (evaluation parity parG): this lane's 16 bytes of each.
        auto g_request = [&](int s, int parG, u32x4 (&pv)[G]) {
            const int t = s / SPM, j = s % SPM;
            // (per-lane parts of an address go into the VECTOR offset: a per-lane scalar offset makes hipcc serialise the access
            // in a waterfall loop over its distinct values -- 80 iterations for these 8 loads, 11 k cycles in the first version)
            const int lp = (psl * 16 + SPM * member + j) * 16 + (pact ? pmt : 0) * 1024;
            const int sb = xG + ((parG * NT + t) * G * DU_KBD) * 1024;
#pragma unroll
            for (int mem = 0; mem < G; ++mem) pv[mem] = du_ld(g, lp, sb + mem * DU_KBD * 1024);
        };
        // g of own sample s: fixed-order sum of the partials + A^T (A s) + c.  have: pv already holds the answer of an earlier request
        // (issued in front of the previous tile's P2, one L2 round trip off the critical path); it is used if it shows no sentinel.
        auto gather_g = [&](int s, int parG, bool have, u32x4 (&pv)[G]) -> f32x4 {
            int spins = 0;
            const DWait tw = du_wait_begin(g, PRED ? DPW_G : -1, s / SPM);
            while (true) {
                if (!have) g_request(s, parG, pv);
                have = false;
                bool bad = du_bad(pv[0]);
                if (!__any(bad && pact)) {
#pragma unroll
                    for (int mem = 1; mem < G; ++mem) bad |= du_bad(pv[mem]);
                    if (!__any(bad && pact)) break;
                }
                if (du_spin(g, spins, DUK_G)) break;
            }
            du_wait_end(g, tw, lane);
            f32x4 gs = du_f(pv[0]);
#pragma unroll
            for (int mem = 1; mem < G; ++mem) gs += du_f(pv[mem]);
            const float4 z = L4[(DAT + s * DS_STRIDE + DS_AZC + pi) >> 2];
            gs += (f32x4){z.x, z.y, z.z, z.w};
            return gs;
        };

        // the owner's step, state part: gradient of evaluation e-1 -> RK update -> stage state of evaluation e published.
        // Leaves this lane's share of sum p^2 (q0) and its candidate for dPhi/dt (gdv) for the cost part.
        auto own_state = [&](int s, int e, float hs, int pst, int pk, float t_pub, bool have, u32x4 (&pv)[G]) {
            const int t = s / SPM, j = s % SPM;
            const int parG = (e - 1) & 1, parS = e & 1;
            const bool rk_last = (pst == nstage - 1);
            const float rk_wa = (nstage == 1) ? 1.f : ((pst == 0 || pst == 3) ? c16 : c26);
            const float rk_wx = (pst < 2) ? 0.5f : 1.f;
            const int sbase = DAT + s * DS_STRIDE;
            DTL(40 * t + 0);
            const float4 z04 = L4[(sbase + DS_Z0 + pi) >> 2], zA4 = L4[(sbase + DS_ZA + pi) >> 2];      // (in flight while the partials are polled)
            if (!have) g_request(s, parG, pv);
            unsigned qa = 0, qb = 0;
            if (pst != nstage) {   // the cost scalars of this sample ride along (consumed by own_costs, behind P2): their round trip is off the critical path
                const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(g.xrs, 0, xQ + (((((e - 1) % 3) * NT + t) * G + member) * (2 * SPM) + 2 * j) * 4, 16);
                qa = v2[0]; qb = v2[1];
            }
            const f32x4 gs = gather_g(s, parG, true, pv);
            // (round 5) "the stage state of this tile is about to be published": the workgroup's waves that gather it start polling NOW -- one
            // owner's step and one store + landing later the payload is there, so a poll is in flight when it lands (the shortest hop there
            // is) but none was during the thousands of cycles before (which the co-resident role's stores paid for in the CU's memory queue)
            if (FLAGS && lane == 0) __hip_atomic_store(reinterpret_cast<int*>(lds) + DAFL + t, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            DTL(40 * t + 1);
            float q0 = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) if (pact && pi + e4 < d) q0 += gs[e4] * gs[e4];
            const float gdv = pd_e == 0 ? gs[0] : (pd_e == 1 ? gs[1] : (pd_e == 2 ? gs[2] : gs[3]));
            const bool pctrl = (pst == nstage);                   // the evaluation just answered was a control evaluation (intermediates)
            const long orow = rr.row0 + own_row(t, j);            // this sample's row in the caller's batch
            const bool orow_ok = own_row(t, j) < ra.n;
            if (pact) {
                const float z0[4] = {z04.x, z04.y, z04.z, z04.w};
                float zA[4] = {zA4.x, zA4.y, zA4.z, zA4.w}, zn[4];
                f32x4 xs;
                // straight-line form of the three cases (selects on wave-uniform conditions, the same operations as the branches:
                //   controls only:  x = z0 (the state stands);   last stage:  x = z_{k+1} = (z0 | zA) + wa K;   else:  zA += wa K, x = z0 + wx K)
                const bool xFromA = !pctrl && rk_last && nstage != 1;      // x starts from the accumulator
                const bool accum = !pctrl && !rk_last;                     // the accumulator takes this stage
                const float cx = pctrl ? 0.f : (rk_last ? rk_wa : rk_wx);
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int i = pi + e4;
                    const float K = hs * -gs[e4];                          // dx = -grad_p H = -p (src/OCflow.py:134, :143-184)
                    const float xb = xFromA ? zA[e4] : z0[e4];
                    const float x_ = pctrl ? z0[e4] : xb + cx * K;
                    const float an = (pst == 0 ? z0[e4] : zA[e4]) + rk_wa * K;
                    zA[e4] = accum ? an : zA[e4];
                    zn[e4] = (rk_last && !pctrl) ? x_ : z0[e4];
                    xs[e4] = (i < d) ? x_ : (i == d ? t_pub : 0.f);
                    if (i >= d) { zn[e4] = 0.f; zA[e4] = 0.f; }
                }
                if (ZF && orow_ok) {
                    // intermediates (src/OCflow.py:37-55), time-major: the state at the end of step pk; the controls -p at (z_{k+1}, t_k)
                    if (pctrl) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) if (pi + e4 < d) ra.ctrlFull[((long)(pk + 1) * rr.n_total + orow) * ra.cdim + pi + e4] = -gs[e4];
                    } else if (rk_last) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) if (pi + e4 < d) ra.zFull[((long)(pk + 1) * rr.n_total + orow) * (d + 4) + pi + e4] = zn[e4];
                    }
                }
                const int lp = (psl * 16 + SPM * member + j) * 16 + pmt * 1024;
                du_st(g, lp, xS + ((parS * NT + t) * DU_KBD) * 1024, xs);
                du_st_sent(g, lp, xS + (((parS ^ 1) * NT + t) * DU_KBD) * 1024);
                if (!pctrl) {
                    if (rk_last) L4[(sbase + DS_Z0 + pi) >> 2] = make_float4(zn[0], zn[1], zn[2], zn[3]);
                    else L4[(sbase + DS_ZA + pi) >> 2] = make_float4(zA[0], zA[1], zA[2], zA[3]);
                }
                L4[(sbase + DS_XS + pi) >> 2] = make_float4(xs[0], xs[1], xs[2], xs[3]);
                // training: the stage input of evaluation e (index e-1); the terminal evaluation is recorded only on a tape
                // (a row of d + 1 floats as a buffer of its own: the scalar unit forms the row's address -- the sample is wave-uniform -- and the range
                // check drops the entries beyond d: no per-lane 64-bit address, no per-entry test)
                if (REC && ra.sAll && e <= ra.nt * nstage + (ra.tapeSc ? 1 : 0) && own_row(t, j) < ra.n)
                    rec_row(ra.sAll + (((long)(e - 1)) * rr.n_total + rr.row0 + own_row(t, j)) * (d + 1), pi, xs[0], xs[1], xs[2], xs[3]);
                // ... and, with an activation record, grad Phi of evaluation e-1 (index e-2)
                if (REC && ra.act && own_row(t, j) < ra.n)
                    rec_row(ra.act + 4 * ra.actRows * MW + (((long)(e - 2)) * rr.n_total + rr.row0 + own_row(t, j)) * (d + 1), pi, gs[0], gs[1], gs[2], gs[3]);
            }
            DTL(40 * t + 2);
            // (behind the store: off the critical path) what the cost part needs, parked in LDS -- P1 and P2 run between the two
            if (pst != nstage) {
                const float sp2 = sum64(q0);
                const float gt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gdv), pd_lane));
                if (lane == 0) L4[(sbase + DS_OC) >> 2] = make_float4(sp2, gt, __uint_as_float(qa), __uint_as_float(qb));
            }
        };
        // (round 6) The owner's step in TWO parts, for the FINE geometry (one tile per group, one own sample per member: OSPLIT).  On the path partial
        // gradients -> stage state only what the stage state needs: the fixed-order sum, K = h (-g), x = base + cx K (base = the step's start state
        // or its RK accumulator: one LDS read at a wave-uniform offset) and the store.  The accumulator, the state at the step's end, the trajectory /
        // record rows and sum p^2 are done by own_book behind P2 from the gradient parked in the sample's A^T z slot (dead between this sum and the
        // next azc_step).  Same expressions on the same operands: the same bits.  Measured on one lease (profiles/r6/05_variants_tried.txt, o1):
        // 256 rows 2.62 -> 2.53 ms, 128 rows 2.31 -> 2.29, a 256-wide network at n = 1024 2.82 -> 2.70; with two tiles per group it LOSES in every
        // geometry (default: +12 %, 256-wide at n = 2048: +3 %; also 512 rows of the default geometry: +2 %) -- there the owner waves' work behind P2
        // holds the next tile's stage-state gather back -- so those launches keep the one-part step.
        auto own_state_split = [&](int s, int e, float hs, int pst, int pk, float t_pub, u32x4 (&pv)[G]) {
            const int t = s / SPM, j = s % SPM;
            const int parG = (e - 1) & 1, parS = e & 1;
            const bool rk_last = (pst == nstage - 1);
            const float rk_wa = (nstage == 1) ? 1.f : ((pst == 0 || pst == 3) ? c16 : c26);
            const float rk_wx = (pst < 2) ? 0.5f : 1.f;
            const int sbase = DAT + s * DS_STRIDE;
            const bool pctrl = (pst == nstage);
            const bool xFromA = !pctrl && rk_last && nstage != 1;
            const float cx = pctrl ? 0.f : (rk_last ? rk_wa : rk_wx);
            DTL(40 * t + 0);
            const float4 b4 = L4[(sbase + (xFromA ? DS_ZA : DS_Z0) + pi) >> 2];                            // (in flight while the partials are polled)
            g_request(s, parG, pv);
            const f32x4 gs = gather_g(s, parG, true, pv);
            if (FLAGS && lane == 0) __hip_atomic_store(reinterpret_cast<int*>(lds) + DAFL + t, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            DTL(40 * t + 1);
            if (pact) {
                const float xb[4] = {b4.x, b4.y, b4.z, b4.w};
                f32x4 xs;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int i = pi + e4;
                    const float K = hs * -gs[e4];
                    const float x_ = pctrl ? xb[e4] : xb[e4] + cx * K;
                    xs[e4] = (i < d) ? x_ : (i == d ? t_pub : 0.f);
                }
                const int lp = (psl * 16 + SPM * member + j) * 16 + pmt * 1024;
                du_st(g, lp, xS + ((parS * NT + t) * DU_KBD) * 1024, xs);
                du_st_sent(g, lp, xS + (((parS ^ 1) * NT + t) * DU_KBD) * 1024);
                L4[(sbase + DS_XS + pi) >> 2] = make_float4(xs[0], xs[1], xs[2], xs[3]);
                L4[(sbase + DS_AZC + pi) >> 2] = make_float4(gs[0], gs[1], gs[2], gs[3]);                   // parked for own_book
            }
            DTL(40 * t + 2);
        };
        // ... the rest of that step, behind P2: leaves sum p^2 (sp2) and dPhi/dt (gt) of the evaluation just answered for the cost part
        auto own_book = [&](int s, int e, float hs, int pst, int pk, float& sp2, float& gt) {
            const int t = s / SPM, j = s % SPM;
            const bool rk_last = (pst == nstage - 1);
            const float rk_wa = (nstage == 1) ? 1.f : ((pst == 0 || pst == 3) ? c16 : c26);
            const float rk_wx = (pst < 2) ? 0.5f : 1.f;
            const int sbase = DAT + s * DS_STRIDE;
            const bool pctrl = (pst == nstage);
            const bool xFromA = !pctrl && rk_last && nstage != 1;
            const bool accum = !pctrl && !rk_last;
            const float cx = pctrl ? 0.f : (rk_last ? rk_wa : rk_wx);
            const float4 z04 = L4[(sbase + DS_Z0 + pi) >> 2], zA4 = L4[(sbase + DS_ZA + pi) >> 2], g4 = L4[(sbase + DS_AZC + pi) >> 2];
            const f32x4 gs = {g4.x, g4.y, g4.z, g4.w};
            float q0 = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) if (pact && pi + e4 < d) q0 += gs[e4] * gs[e4];
            const float gdv = pd_e == 0 ? gs[0] : (pd_e == 1 ? gs[1] : (pd_e == 2 ? gs[2] : gs[3]));
            const long orow = rr.row0 + own_row(t, j);
            const bool orow_ok = own_row(t, j) < ra.n;
            if (pact) {
                const float z0[4] = {z04.x, z04.y, z04.z, z04.w};
                float zA[4] = {zA4.x, zA4.y, zA4.z, zA4.w}, zn[4];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int i = pi + e4;
                    const float K = hs * -gs[e4];
                    const float xb = xFromA ? zA[e4] : z0[e4];
                    const float x_ = pctrl ? z0[e4] : xb + cx * K;
                    const float an = (pst == 0 ? z0[e4] : zA[e4]) + rk_wa * K;
                    zA[e4] = accum ? an : zA[e4];
                    zn[e4] = (rk_last && !pctrl) ? x_ : z0[e4];
                    if (i >= d) { zn[e4] = 0.f; zA[e4] = 0.f; }
                }
                if (ZF && orow_ok) {
                    if (pctrl) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) if (pi + e4 < d) ra.ctrlFull[((long)(pk + 1) * rr.n_total + orow) * ra.cdim + pi + e4] = -gs[e4];
                    } else if (rk_last) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) if (pi + e4 < d) ra.zFull[((long)(pk + 1) * rr.n_total + orow) * (d + 4) + pi + e4] = zn[e4];
                    }
                }
                if (!pctrl) {
                    if (rk_last) L4[(sbase + DS_Z0 + pi) >> 2] = make_float4(zn[0], zn[1], zn[2], zn[3]);
                    else L4[(sbase + DS_ZA + pi) >> 2] = make_float4(zA[0], zA[1], zA[2], zA[3]);
                }
                if (REC && ra.sAll && e <= ra.nt * nstage + (ra.tapeSc ? 1 : 0) && own_row(t, j) < ra.n) {
                    const float4 x4 = L4[(sbase + DS_XS + pi) >> 2];
                    rec_row(ra.sAll + (((long)(e - 1)) * rr.n_total + rr.row0 + own_row(t, j)) * (d + 1), pi, x4.x, x4.y, x4.z, x4.w);
                }
                if (REC && ra.act && own_row(t, j) < ra.n)
                    rec_row(ra.act + 4 * ra.actRows * MW + (((long)(e - 2)) * rr.n_total + rr.row0 + own_row(t, j)) * (d + 1), pi, gs[0], gs[1], gs[2], gs[3]);
            }
            sp2 = 0.f; gt = 0.f;
            if (pst != nstage) {
                sp2 = sum64(q0);
                gt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gdv), pd_lane));
            }
        };
        // ... cost part (off the critical path: it runs behind P1, while the u0 exchange travels): sum p^2, dPhi/dt, the x-only terms
        // from role B -> the four cost integrals of evaluation e-1
        // (pre: qa_ / qb_ hold the answer of a request issued just before -- DU_X_SWAP -- instead of the one that rode with the partial gradients)
        auto own_costs = [&](int s, int e, float hs, int pst, int pk, bool pre = false, unsigned qa_ = 0u, unsigned qb_ = 0u, float sp2_ = 0.f, float gt_ = 0.f) {
            const int t = s / SPM, j = s % SPM;
            const int parG = (e - 1) & 1;
            const bool rk_last = (pst == nstage - 1);
            const float rk_wa = (nstage == 1) ? 1.f : ((pst == 0 || pst == 3) ? c16 : c26);
            const int sbase = DAT + s * DS_STRIDE;
            float4 oc = make_float4(sp2_, gt_, 0.f, 0.f);            // (OSPLIT: own_book hands sum p^2 and dPhi/dt over in registers)
            if (!OSPLIT) oc = L4[(sbase + DS_OC) >> 2];
            const float sp2 = oc.x, gt = oc.y;
            unsigned qa = (pre || OSPLIT) ? qa_ : __float_as_uint(oc.z), qb = (pre || OSPLIT) ? qb_ : __float_as_uint(oc.w);
            bool have = true;
            // (q, w) of this sample at the state of evaluation e-1, from role B of this member (every lane loads the same 8 bytes)
            float q_ = 0.f, w_ = 0.f;
            {
                const int ob = xQ + (((((e - 1) % 3) * NT + t) * G + member) * (2 * SPM) + 2 * j) * 4;
                int spins = 0;
                while (true) {
                    if (!have) { const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(g.xrs, 0, ob, 16); qa = v2[0]; qb = v2[1]; }
                    have = false;
                    q_ = __uint_as_float(qa); w_ = __uint_as_float(qb);
                    if (!__any(qa == DU_SENT || qb == DU_SENT)) break;
                    if (du_spin(g, spins, DUK_Q)) break;
                }
            }
            if (lane < 4) {
                // calcLHQW of the point-agent problems (Cross2D.py:73-87 returns the scaled Q, SwarmTraj.py:71-87 the raw one)
                const float Qs = cAlphQ * q_;
                const float Wv = cWantW ? w_ : 0.f;
                float Lg = 0.5f * sp2 + Qs;
                if (cWantW) Lg = Lg + cAlphW * Wv;
                const float H = -Lg + sp2;
                if (REC && ra.tapeSc && lane == 0 && own_row(t, j) < ra.n)      // tape: dPhi/dt - H and the x-only terms of evaluation e-1 (index e-2)
                    *reinterpret_cast<float4*>(ra.tapeSc + (((long)(e - 2)) * rr.n_total + rr.row0 + own_row(t, j)) * 4) = make_float4(gt - H, q_, w_, 0.f);
                const float val = (lane == 0) ? Lg : (lane == 1) ? fabsf(gt - H) : (lane == 2) ? (PD == 2 ? Qs : q_) : Wv;
                const float K = hs * val;
                float* cz = lds + sbase + DS_CZ + lane;                     // [0..3] value, [4..7] RK accumulator
                const float cz0 = cz[0], czA = cz[4];
                if (rk_last) {
                    const float cn = (nstage == 1 ? cz0 : czA) + rk_wa * K;
                    cz[0] = cn;
                    if (ZF && own_row(t, j) < ra.n) ra.zFull[((long)(pk + 1) * rr.n_total + rr.row0 + own_row(t, j)) * (d + 4) + d + lane] = cn;
                } else cz[4] = (pst == 0 ? cz0 : czA) + rk_wa * K;
            }
            DTL(40 * t + 3);
        };

        // z = A s and A^T z + c of own sample s at its CURRENT stage state (needed when the gradient of this evaluation arrives).
        // LDS reads in batches that are in flight together (a dependent read costs ~130 cycles when waited for alone).
        auto azc_step = [&](int s, bool fin) {
            const int sbase = DAT + s * DS_STRIDE;
            {
                const int q = lane >> 2, part = lane & 3;            // row of A, quarter of the dims (10 float4 each)
                const int ia = ((DA_A + q * DU_DP) >> 2) + part * 10, ix = ((sbase + DS_XS) >> 2) + part * 10;
                float acc = 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 a[5], x4[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) { a[i] = L4[ia + 5 * h + i]; x4[i] = L4[ix + 5 * h + i]; }
#pragma unroll
                    for (int i = 0; i < 5; ++i) acc += (a[i].x * x4[i].x + a[i].y * x4[i].y) + (a[i].z * x4[i].z + a[i].w * x4[i].w);
                }
                acc = sum4(acc);
                if (part == 0) lds[sbase + DS_ZQ + q] = acc;
            }
            float lin = 0.f;
            if (pact) {
                const float4 c4 = L4[(DA_CW + pi) >> 2];
                float o[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int qb = 0; qb < 16; qb += 4) {                 // rows of A beyond r are zero
                    if (qb < rA) {
                        const float4 z4 = L4[(sbase + DS_ZQ + qb) >> 2];
                        float4 a[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[i] = L4[(DA_A + (qb + i) * DU_DP + pi) >> 2];
                        const float z[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) { o[0] += a[i].x * z[i]; o[1] += a[i].y * z[i]; o[2] += a[i].z * z[i]; o[3] += a[i].w * z[i]; }
                    }
                }
                L4[(sbase + DS_AZC + pi) >> 2] = make_float4(o[0], o[1], o[2], o[3]);
                if (fin) { const float4 x4 = L4[(sbase + DS_XS + pi) >> 2]; lin = (c4.x * x4.x + c4.y * x4.y) + (c4.z * x4.z + c4.w * x4.w); }
            }
            if (fin) {                                            // Phi's linear and quadratic terms at the final time (src/Phi.py:91-96)
                const float ls = sum64(lin);
                if (lane == 0) {
                    float qd = 0.f;
                    for (int q = 0; q < rA; ++q) { const float z = lds[sbase + DS_ZQ + q]; qd += 0.5f * z * z; }
                    lds[sbase + DS_PHX] = ls;
                    lds[sbase + DS_PHX + 1] = qd;
                }
            }
        };

        // ---- evaluation 1: the stage states come straight from x
        for (int s = wave; s < SPM * NT; s += 4) {
            const int t = s / SPM, j = s % SPM, sbase = DAT + s * DS_STRIDE;
            if (pact) {
                const float4 x4 = L4[(sbase + DS_XS + pi) >> 2];
                du_st(g, (psl * 16 + SPM * member + j) * 16 + pmt * 1024, xS + ((1 * NT + t) * DU_KBD) * 1024, (f32x4){x4.x, x4.y, x4.z, x4.w});
            }
            if (REC && ra.sAll && own_row(t, j) < ra.n)
                for (int i = lane; i <= d; i += 64) ra.sAll[(rr.row0 + own_row(t, j)) * (d + 1) + i] = lds[sbase + DS_XS + i];
            if (ZF && own_row(t, j) < ra.n) {                 // z_0 = [x, 0, 0, 0, 0]; the controls of slot 0 stay zero (SURVEY 8a note 3)
                for (int i = lane; i < d + 4; i += 64) ra.zFull[(rr.row0 + own_row(t, j)) * (d + 4) + i] = (i < d) ? lds[sbase + DS_Z0 + i] : 0.f;
                for (int i = lane; i < ra.cdim; i += 64) ra.ctrlFull[(rr.row0 + own_row(t, j)) * ra.cdim + i] = 0.f;
            }
        }

        double tk = ra.t0;
        int e = 0;
        float p_hs = 0.f; int p_st = 0, p_k = 0;
        const int nsub = nstage + (ZF ? 1 : 0);             // evaluations per step: the RK stages (+ the control evaluation of intermediates)
        u32x4 pf[G];                                               // the G members' partial gradients of an own sample (this lane's 16 bytes of each)
        for (int k = 0; k <= ra.nt; ++k) {
            const bool fin = (k == ra.nt);
            const double t1k = tk + ra.h;
            const double hsd = t1k - tk;                           // stepRK4 re-derives h = t1 - t0 (src/OCflow.py:170)
            for (int st = 0; st < (fin ? 1 : nsub); ++st) {
                ++e;
                const int par = e & 1;
                DTL_EPOCH(e);
                // time of this evaluation: stepRK4's t0, t0 + h/2, t0 + h/2, t0 + h in double (src/OCflow.py:157-184); the terminal
                // evaluation runs at tspan[1] (src/OCflow.py:62); the control evaluation of intermediates at the step's START time with the
                // step's END state (src/OCflow.py:51-55: tk is advanced behind it)
                const double te = fin ? ra.t1 : ((nstage == 1 || st == 0 || st == nstage) ? tk : (st == 3 ? tk + hsd : tk + hsd / 2));
                // tile after tile: while this workgroup multiplies tile t, role B works on the tile before it
                for (int t = 0; t < NT; ++t) {
                    int sown = -1;                                  // this wave's own sample of the tile, if any: sample s belongs to wave s & 3
#pragma unroll
                    for (int j = 0; j < SPM; ++j) if (((SPM * t + j) & 3) == wave) sown = SPM * t + j;
                    if (e > 1 && sown >= 0) { if (OSPLIT) own_state_split(sown, e, p_hs, p_st, p_k, (float)te, pf); else own_state(sown, e, p_hs, p_st, p_k, (float)te, false, pf); }
                    // ================= P1: o = K0[H_c,:] s + b0 ; u0 = sigma(o), tanh(o) =================
                    DTL(40 * t + 4);
                    // (the two waves that own nothing of this tile fetch the stage states: the owners have just stored theirs, and a load issued
                    // behind a store is not answered before the store has been acknowledged -- vmcnt retires in order -- which occasionally
                    // takes thousands of cycles that every member of the group then waits for: 5.54 -> 5.29 ms)
                    if (e > 1) {
                        // the waves that own nothing of this tile wait for the local owner's flag (LDS: no request leaves the CU), then gather
                        auto wait_owner = [&]() {
                            // (one tile per group only -- measured, tools/r5_ab.sh: 512 rows 3.34 -> 3.16 ms, 128 rows 2.32 -> 2.30; with two tiles
                            // per group the early polls are worth more than they cost: 1024 rows 5.14 -> 5.29 with the flag)
                            if (!FLAGS || g.slow) return;
                            int spins = 0;                  // (one tile per group: du_spin naps 64 clocks; bounded like every wait of this kernel)
                            while (__hip_atomic_load(reinterpret_cast<int*>(lds) + DAFL + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != e) {
                                if (du_spin(g, spins, DUK_S)) break;
                            }
                        };
                        if (SPM == 2) { if (sown < 0) { wait_owner(); du_gather2<DU_KBD>(g, (wave ^ (wave >> 1)) & 1, lane, xS + ((par * NT + t) * DU_KBD) * 1024, DA_SF >> 2, DUK_S); } }
                        else if (SPM == 4) du_gather<DU_KBD>(g, wave, lane, xS + ((par * NT + t) * DU_KBD) * 1024, DA_SF >> 2, DUK_S);      // (every wave is an owner)
                        else { const int wh = ((wave - (t & 3)) & 3) - 1; if (wh >= 0 && wh < 2) { wait_owner(); du_gather2<DU_KBD>(g, wh, lane, xS + ((par * NT + t) * DU_KBD) * 1024, DA_SF >> 2, DUK_S); } }
                    } else du_gather<DU_KBD>(g, wave, lane, xS + ((par * NT + t) * DU_KBD) * 1024, DA_SF >> 2, DUK_S);
                    DTLB(40 * t + 13);
                    __syncthreads();
                    DTL(40 * t + 5);
                    {
                        float4 b0s;                                                // bias of this lane's 4 features 16 mt + 4 slot + e
                        f32x4 acc = du_gemm_lds<KB1, false>(K0r, (DA_SF >> 2) + kh * KB1 * 64 + lane, [&](int kb) {
                            if (kb == KB1 - 3) b0s = L4[(DA_VEC >> 2) + 4 * ft + slot];
                            if (KS > 1 && kb == 1) {
                                // fine form: the slots of the previous evaluation are reset HERE (S of this evaluation is staged: every reader is
                                // done), a whole product + hop earlier than in the default form: P2 is only 64 MFMAs long there and its vmcnt(0) in front
                                // of the V store would wait for these acknowledgements (H1 is unchanged: that wait still lies between them and V)
                                const int fr = (((par ^ 1) * NT + t) * KBM + MTM * member + ft) * 1024;
                                if (kh == 0) { du_st_sent(g, vb, xU + fr); du_st_sent(g, vb, xV + fr); } else du_st_sent(g, vb, xT + fr);
                            }
                        });
                        if (KS > 1) {       // the two halves of the contraction meet (fixed order: lower + upper k-blocks)
                            L4[(DAPX >> 2) + wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                            __syncthreads();
                            const float4 o4 = L4[(DAPX >> 2) + (wave ^ MTM) * 64 + lane];
                            const f32x4 ot = {o4.x, o4.y, o4.z, o4.w};
                            acc = kh ? ot + acc : acc + ot;
                        }
                        DTL(40 * t + 6);
                        const float b0v[4] = {b0s.x, b0s.y, b0s.z, b0s.w};
                        f32x4 sg, th;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) { float s_, t_; act_pair(acc[e4] + b0v[e4], s_, t_); sg[e4] = s_; th[e4] = t_; }
                        const int fo = ((par * NT + t) * KBM + MTM * member + ft) * 1024;
                        // (fine form: both waves of a feature tile hold o; one publishes sigma(o), the other tanh(o))
                        if (KS == 1 || kh == 0) du_st(g, vb, xU + fo, sg);
                        if (KS == 1 || kh == 1) du_st(g, vb, xT + fo, th);
                        if (REC && ra.act && (!fin || ra.tapeSc)) {    // activation record: 4 features of sample lane & 15 (64-byte runs per sample)
                            const int ro = rec_off(t);
                            if (KS == 1 || kh == 0) rec_store(rec_block(0, e - 1), ro, sg[0], sg[1], sg[2], sg[3]);
                            if (KS == 1 || kh == 1) rec_store(rec_block(1, e - 1), ro, th[0], th[1], th[2], th[3]);
                        }
                    }
                    DTL(40 * t + 7);
#ifdef NOCF_STAMPS
                    if (dp.dbg & 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DTL(40 * t + 18); }      // how long do the five stores take to be acknowledged?
#endif
                    DTL(40 * t + 8);
                    // ================= P2: q = K1[H_c,:] u0 + b1 ; v = tanh(q) . w =================
                    // (the members of a group finish P1 within a few hundred clocks of each other, and a wave arrives here right behind its own U
                    // store: a poll issued at once reaches L2 before the slowest member's fragment and costs a whole second round trip)
                    for (int i = 0; i < g.udelay; ++i) __builtin_amdgcn_s_sleep(1);
                    du_gather<KBM>(g, wave, lane, xU + ((par * NT + t) * KBM) * 1024, DA_UF >> 2, DUK_U);
                    DTLB(40 * t + 14);
                    __syncthreads();
                    DTL(40 * t + 10);
                    {
                        // the slots of the previous evaluation (every reader is done: S of this evaluation exists).  Here, not in the P1
                        // epilogue: no load of this wave waits behind them, and the GEMM covers their acknowledgement (V one phase early:
                        // header, H1)
                        const int fr = (((par ^ 1) * NT + t) * KBM + MTM * member + ft) * 1024;
                        float4 b1s, wvs;
                        auto resets = [&](int kb) {
                            if (KS == 1 && kb == 1) { du_st_sent(g, vb, xU + fr); du_st_sent(g, vb, xT + fr); du_st_sent(g, vb, xV + fr); }      // (fine form: in P1)
                            if (kb == KBW - 3) { b1s = L4[((DA_VEC + 64) >> 2) + 4 * ft + slot]; wvs = L4[((DA_VEC + 128) >> 2) + 4 * ft + slot]; }
                        };
                        f32x4 acc = du_gemm_lds<KBW, true>(W, (DA_UF >> 2) + kh * KBW * 64 + lane, resets);
                        DTL(40 * t + 11);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the resets are acknowledged -- long ago -- before V is stored: header, H1)
                        if (KS > 1) {       // (the barrier also orders the tanh(o) reset of the partner wave, acknowledged above, before this wave's V store)
                            L4[(DAPX >> 2) + wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                            __syncthreads();
                            const float4 o4 = L4[(DAPX >> 2) + (wave ^ MTM) * 64 + lane];
                            const f32x4 ot = {o4.x, o4.y, o4.z, o4.w};
                            acc = kh ? ot + acc : acc + ot;
                        }
                        const float b1v[4] = {b1s.x, b1s.y, b1s.z, b1s.w}, wv[4] = {wvs.x, wvs.y, wvs.z, wvs.w};
                        f32x4 v;
#ifdef NOCF_STAMPS
                        asm volatile("" : "+v"(v) : "v"(b1s.x), "v"(wvs.x));
                        DTL(40 * t + 16);
#endif
                        f32x4 tq;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) { tq[e4] = tanh_fast(acc[e4] + b1v[e4]); v[e4] = tq[e4] * wv[e4]; }
#ifdef NOCF_STAMPS
                        asm volatile("" : "+v"(v));
                        DTL(40 * t + 17);
#endif
                        if (KS == 1 || kh == 0) du_st(g, vb, xV + ((par * NT + t) * KBM + MTM * member + ft) * 1024, v);
                        if (PRED) du_anchor(g, t, lane);        // (predictive waiting: this wave's next waits for tile t count from here)
                        DTL(40 * t + 12);
                        if (REC && ra.act && (!fin || ra.tapeSc) && (KS == 1 || kh == 0))      // activation record: tanh(q)
                            rec_store(rec_block(2, e - 1), rec_off(t), tq[0], tq[1], tq[2], tq[3]);
                        if (fin && (KS == 1 || kh == 1)) {     // (fine form: the wave that does not publish v takes the value's part)
                            // w . u_1 = w . (u_0 + hN sigma(q)) over this wave's 16 features (src/Phi.py:91-96): own u_0 fragment from the staged tile
                            const float4 u4 = L4[(DA_UF >> 2) + (MTM * member + ft) * 64 + lane];
                            const float u0[4] = {u4.x, u4.y, u4.z, u4.w};
                            float pr = 0.f, u1[4];
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) { u1[e4] = u0[e4] + hN * sigma_act(acc[e4] + b1v[e4]); pr += wv[e4] * u1[e4]; }
                            if (REC && ra.tapeU1) {                    // tape: u_1 of the terminal evaluation (the value's dw row)
                                const long rw = rowg + 16 * t + (lane & 15);
                                if (rw < ra.n) *reinterpret_cast<float4*>(ra.tapeU1 + (rr.row0 + rw) * MW + HPM * member + 16 * ft + 4 * slot) = make_float4(u1[0], u1[1], u1[2], u1[3]);
                            }
                            pr += __shfl_xor(pr, 16); pr += __shfl_xor(pr, 32);
                            if (lane < 16) {
                                const unsigned pu = __float_as_uint(pr);
                                const int off = xP + (((t * G + member) * MTM + ft) * 16 + lane) * 4;
                                if (g.fast) __builtin_amdgcn_raw_buffer_store_b32(pu, g.xrs, off, 0, 0);
                                else __builtin_amdgcn_raw_buffer_store_b32(pu, g.xrs, off, 0, 16);
                            }
                        }
                    }
                    // (behind P2, while v travels and role B multiplies) the owner of a sample integrates its costs of the previous evaluation
                    // and forms z = A s, A^T z + c for its NEXT step -- wave-local: no other wave reads what it writes.  (In front of the U
                    // gather these two delayed it: they take as long as the u0 hop.  The waves that own nothing of this tile go on: with two
                    // tiles they are the next tile's owners.)
                    if (sown >= 0) {
                        // z / A^T z FIRST (round 6): it is vector work, and the SIMD's ALU is free while v travels (role B polls for it); the cost
                        // part is mostly the wait for role B's cost scalars, requested here so that the round trip runs under that work
                        const bool oc = e > 1 && p_st != nstage;
                        unsigned pqa = DU_SENT, pqb = DU_SENT;
                        if (oc) {
                            const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(g.xrs, 0, xQ + (((((e - 1) % 3) * NT + sown / SPM) * G + member) * (2 * SPM) + 2 * (sown % SPM)) * 4, 16);
                            pqa = v2[0]; pqb = v2[1];
                        }
                        float sp2b = 0.f, gtb = 0.f;
                        if (OSPLIT && e > 1) own_book(sown, e, p_hs, p_st, p_k, sp2b, gtb);      // (reads the parked gradient: before azc_step rewrites its slot)
                        azc_step(sown, fin);
                        if (oc) own_costs(sown, e, p_hs, p_st, p_k, true, pqa, pqb, sp2b, gtb);
                    }
                    DTL(40 * t + 9);
                }
                p_hs = (float)hsd; p_st = st; p_k = k;
                if (fin) break;
            }
            tk += ra.h;
        }
        // ---- terminal costs of the own samples (src/OCflow.py:58-76): gradient and Phi of the terminal evaluation
        for (int s = wave; s < SPM * NT; s += 4) {
            const int t = s / SPM, j = s % SPM, sbase = DAT + s * DS_STRIDE;
            const f32x4 gs = gather_g(s, e & 1, false, pf);
            float r2 = 0.f, hg = 0.f;
            if (pact) {
                const float4 z = L4[(sbase + DS_Z0 + pi) >> 2];
                const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) if (pi + e4 < d) { const float res = zz[e4] - pb.xtarget[pi + e4]; r2 += res * res; hg += fabsf(gs[e4] - ra.a0 * res); }
            }
            const float cG = 0.5f * sum64(r2), hj = sum64(hg);
            // w . u_1: the G members x MTM feature tiles = 32 partials of this sample (lanes 0..31)
            float ph;
            {
                unsigned u = 0;
                int spins = 0;
                const int off = xP + ((t * G * MTM + (lane & (G * MTM - 1))) * 16 + SPM * member + j) * 4;
                while (true) {
                    u = __builtin_amdgcn_raw_buffer_load_b32(g.xrs, off, 0, 16);
                    if (!__any(u == DU_SENT)) break;
                    if (du_spin(g, spins, DUK_P)) break;
                }
                ph = sum64(lane < G * MTM ? __uint_as_float(u) : 0.f);
            }
            const long row = own_row(t, j);
            if (REC && ra.tapeSc && row < ra.n) {               // tape: grad Phi and Phi - alph0 G of the terminal evaluation (block nt * nstage)
                const long blk = (long)ra.nt * nstage;
                if (pact) {
                    float* dst = ra.act + 4 * ra.actRows * MW + (blk * rr.n_total + rr.row0 + row) * (d + 1);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) if (pi + e4 <= d) dst[pi + e4] = gs[e4];
                }
                if (lane == 0) {
                    const float phi = ph + lds[sbase + DS_PHX + 1] + lds[sbase + DS_PHX] + dp.cb;
                    *reinterpret_cast<float4*>(ra.tapeSc + (blk * rr.n_total + rr.row0 + row) * 4) = make_float4(phi - ra.a0 * cG, 0.f, 0.f, 0.f);
                }
            }
            if (row < ra.n) {
                if (lane == 0 && ra.persample) {
                    const float phi = ph + lds[sbase + DS_PHX + 1] + lds[sbase + DS_PHX] + dp.cb;
                    const float* cz = lds + sbase + DS_CZ;
                    float* op = ra.persample + row * 7;
                    op[0] = cz[0]; op[1] = cG; op[2] = cz[1];
                    op[3] = fabsf(phi - ra.a0 * cG);
                    op[4] = hj;
                    op[5] = cz[2]; op[6] = cz[3];
                }
                if (ra.z_out)
                    for (int i = lane; i < d + 4; i += 64)
                        ra.z_out[row * (d + 4) + i] = (i < d) ? lds[sbase + DS_Z0 + i] : lds[sbase + DS_CZ + (i - d)];
            }
        }
#ifdef NOCF_ACC
        acc_dump();
#endif
    } else {
        // =====================================================================================================
        // role B: x-only cost terms, P3, P4
        // =====================================================================================================
        for (int i = tid; i < DU_KBD * MTM * 64; i += 256) L4[(DB_K4 >> 2) + i] = ws4[dp.oK4 + (long)member * DU_KBD * MTM * 64 + i];
        if (tid < HPM) lds[DB_VEC + tid] = ws[dp.oVec + 2 * MW + member * HPM + tid];
        if (PRED) du_calibrate(g, ((dp.dbg & 8) || (NT > 1 && !(dp.dbg & 16))) ? -1 : DBPW, wave, lane);
        __syncthreads();
        const DXPar xp = du_x_params(pb, PD);
        const int nsub = nstage + (ZF ? 1 : 0);
        const int E = ra.nt * nsub + 1;
        u32x4 spf = {0u, 0u, 0u, 0u};                              // prefetched own-state piece of tile spf_t (-1: none)
        int spf_t = -1;
        for (int e = 1; e <= E; ++e) {
            const bool fin = (e == E);
            const bool stage = !fin && ((e - 1) % nsub) < nstage;         // an RK stage: its running costs are integrated (not: terminal, control)
            const int par = e & 1;
            DTL_EPOCH(e);
            for (int t = 0; t < NT; ++t) {
                DTL(40 * t + 20);
                if (stage) {
                    // ================= x-only cost terms of the own samples of tile t at the state of evaluation e =================
                    if (wave < SPM) {                                   // 40 SPM pieces of 16 B: dims 4 l .. 4 l + 3 of own sample j
                        const int p = tid < 40 * SPM ? tid : 0, j = p / 40, r_ = p - 40 * j, pmt = r_ >> 2, sl = r_ & 3;
                        const int off = xS + ((par * NT + t) * DU_KBD + pmt) * 1024 + (sl * 16 + SPM * member + j) * 16;
                        u32x4 v = spf;
                        bool have = spf_t == t;                         // requested before the previous tile's P4 stores (see there)
                        int spins = 0;
                        while (true) {
                            if (!have) v = du_ld(g, off, 0);
                            have = false;
                            if (!__any(du_bad(v))) break;
                            if (du_spin(g, spins, DUK_S)) break;
                        }
                        spf_t = -1;
                        const f32x4 f = du_f(v);
                        if (tid < 40 * SPM) {                           // scatter into the [entry][4] layout, every agent twice (entries i and i + N)
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const int i = 4 * r_ + e4;
                                if (i < d) {
                                    const int ag = i / PD, k = i - PD * ag;
                                    lds[DB_XA + (j * 128 + ag) * 4 + k] = f[e4];
                                    lds[DB_XA + (j * 128 + ag + xp.N) * 4 + k] = f[e4];
                                }
                            }
                        }
                    }
                    __syncthreads();
                    DTL(40 * t + 21);
                    {
                        float q_ = 0.f, w_ = 0.f;
                        // (two own samples: wave = (sample, half of the partner range); one: the four waves take a quarter each)
                        // (four: wave = sample, the whole range)
                        if (SPM == 2) du_x_wave<PD, 2>(pb, xp, (DB_XA >> 2) + (wave & 1) * 128, lane, wave >> 1, q_, w_);
                        else if (SPM == 4) du_x_wave<PD, 1>(pb, xp, (DB_XA >> 2) + wave * 128, lane, 0, q_, w_);
                        else du_x_wave<PD, 4>(pb, xp, DB_XA >> 2, lane, wave, q_, w_);
                        q_ = sum64(q_); w_ = sum64(w_);
                        if (lane == 0) { lds[DB_XP + 2 * wave] = q_; lds[DB_XP + 2 * wave + 1] = w_; }
                    }
                    __syncthreads();
                    DTL(40 * t + 22);
                    if (wave == 0 && lane < SPM) {                      // sample j = lane: the parts in a fixed order
                        float q_, w_;
                        if (SPM == 2) { q_ = lds[DB_XP + 2 * lane] + lds[DB_XP + 2 * (lane + 2)]; w_ = lds[DB_XP + 2 * lane + 1] + lds[DB_XP + 2 * (lane + 2) + 1]; }
                        else if (SPM == 4) { q_ = lds[DB_XP + 2 * lane]; w_ = lds[DB_XP + 2 * lane + 1]; }
                        else { q_ = (lds[DB_XP] + lds[DB_XP + 2]) + (lds[DB_XP + 4] + lds[DB_XP + 6]); w_ = (lds[DB_XP + 1] + lds[DB_XP + 3]) + (lds[DB_XP + 5] + lds[DB_XP + 7]); }
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        const u32x2 pay = {__float_as_uint(q_), __float_as_uint(w_)};
                        const int off = xQ + ((((e % 3) * NT + t) * G + member) * (2 * SPM) + 2 * lane) * 4;          // (three slots: see the reset in P3)
                        if (g.fast) __builtin_amdgcn_raw_buffer_store_b64(pay, g.xrs, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b64(pay, g.xrs, off, 0, 16);      // (the slot is reset below, behind the V gather)
                    }
                    DTL(40 * t + 23);
                }
                // ================= P3: a = w + hN K1[:,H_c]^T v ; y = tanh(o) . a =================
                const int fo = ((par * NT + t) * KBM + MTM * member + ft) * 1024;
                DTL(40 * t + 24);
                du_gather<KBM>(g, wave, lane, xV + ((par * NT + t) * KBM) * 1024, DB_VF >> 2, DUK_V, PRED ? DPW_V : -1, t);
                DTLB(40 * t + 15);
                __syncthreads();
                DTL(40 * t + 25);
                u32x4 thv = {DU_SENT, DU_SENT, DU_SENT, DU_SENT};
                float4 wvs;
                auto mid = [&](int kb) {
                    if (kb == KBW - 3) wvs = L4[(DB_VEC >> 2) + 4 * ft + slot];
                    if (kb != 1) return;
                    // V(e) is complete, so every owner has read the partial gradients of the previous evaluation (it published S(e) after
                    // them): reset those slots now (behind the second k-block: the stores issue in the MFMAs' shadow).  The tanh(o) load is
                    // younger than these stores, and vmcnt retires in order: its wait lies between the resets and the payloads stored at the
                    // end of P4 (header, H1).  The cost scalars have THREE slots (e mod 3): the owner integrates the costs of evaluation e-1
                    // behind P2 of evaluation e, i.e. possibly after V(e) is complete, so the slot reset here is the one of evaluation e-2
                    // (its reader finished before it published S(e)); it next takes the scalars of evaluation e+1.
                    // (fine form: the tanh(o) request goes FIRST -- its answer is then not queued behind the resets' acknowledgements, which P3's 64
                    // MFMAs are too short to cover; the wait that H1 needs sits in front of P4's stores instead)
                    if (KS > 1) thv = du_ld(g, vb, xT + fo);
                    const int gR = xG + ((((par ^ 1) * NT + t) * G + member) * DU_KBD) * 1024;
#pragma unroll
                    for (int mi = 0; mi < 3; ++mi) { const int mt = wave + 4 * mi; if (mt < DU_KBD) du_st_sent(g, vb, gR + mt * 1024); }
                    if (wave == 0 && lane < SPM) {
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        const u32x2 sen = {DU_SENT, DU_SENT};
                        const int offr = xQ + (((((e + 1) % 3) * NT + t) * G + member) * (2 * SPM) + 2 * lane) * 4;
                        if (g.fast) __builtin_amdgcn_raw_buffer_store_b64(sen, g.xrs, offr, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b64(sen, g.xrs, offr, 0, 16);
                    }
                    if (KS == 1) thv = du_ld(g, vb, xT + fo);
                };
                f32x4 acc = du_gemm_lds<KBW, true>(W, (DB_VF >> 2) + kh * KBW * 64 + lane, mid);
                // (round 4 pinned the first look at the tanh(o) answer behind the product with an empty asm; measured again in round 6 on one box,
                // the pin costs 1.6 % at n = 1024 and gains nothing at 128 ... 512 rows: taken out)
                DTL(40 * t + 26);
                {
                    int spins = 0;
                    while (__any(du_bad(thv))) {
                        if (du_spin(g, spins, DUK_T)) break;
                        thv = du_ld(g, vb, xT + fo);
                    }
                }
                if (KS == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the resets above have landed: see there; fine form: in front of P4's stores)
                if (KS > 1) {               // the two halves of the contraction meet (fixed order: lower + upper k-blocks)
                    L4[(DBPX >> 2) + wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    __syncthreads();
                    const float4 o4 = L4[(DBPX >> 2) + (wave ^ MTM) * 64 + lane];
                    const f32x4 ot = {o4.x, o4.y, o4.z, o4.w};
                    acc = kh ? ot + acc : acc + ot;
                }
                {
                    const f32x4 th = du_f(thv);
                    const float4 a4 = make_float4(wvs.x + hN * acc[0], wvs.y + hN * acc[1], wvs.z + hN * acc[2], wvs.w + hN * acc[3]);
                    float4 y;
                    y.x = th[0] * a4.x; y.y = th[1] * a4.y;
                    y.z = th[2] * a4.z; y.w = th[3] * a4.w;
                    if (KS == 1 || kh == 0) L4[(DB_YF >> 2) + ft * 64 + lane] = y;
                    if (REC && ra.act && (!fin || ra.tapeSc) && (KS == 1 || kh == 1))          // activation record: a = w + hN K1' v
                        rec_store(rec_block(3, e - 1), rec_off(t), a4.x, a4.y, a4.z, a4.w);
                }
                DTL(40 * t + 27);
                // (in front of the barrier: they do not depend on y, and the other waves' epilogues cover them)
                // the own-state pieces of the NEXT tile's cost pass: requested here, in front of P4 and its stores (a load issued behind
                // a store is not answered before the store is: vmcnt retires in order), consumed at the next tile's entry
                if (wave < SPM && !fin) {
                    const int tn = (t + 1 < NT) ? t + 1 : 0, pn = (t + 1 < NT) ? par : (par ^ 1);
                    const int en = (t + 1 < NT) ? e : e + 1;
                    if (en < E && ((en - 1) % nsub) < nstage) {
                        const int p = tid < 40 * SPM ? tid : 0, j = p / 40, r_ = p - 40 * j, pmt = r_ >> 2, sl = r_ & 3;
                        spf = du_ld(g, xS + ((pn * NT + tn) * DU_KBD + pmt) * 1024 + (sl * 16 + SPM * member + j) * 16, 0);
                        spf_t = tn;
                    }
                }
                const int gP = xG + (((par * NT + t) * G + member) * DU_KBD) * 1024;
                if (KS > 1) {
                    // ================= P4 (fine form): partial g = K0[H_c,:]^T y, the wave's dim tiles wave, wave+4 (, wave+8) as ONE stream =================
                    // 8 MFMAs per dim tile are too few to run tile after tile (750 cycles per tile measured for 256 of issue: every tile paid its
                    // own operand wait, fence and store): the tiles' chains are interleaved, one fence, then the stores
                    const bool third = wave + 8 < DU_KBD;
                    float4 w0[MTM], w1[MTM], w2[MTM];
#pragma unroll
                    for (int kb = 0; kb < MTM; ++kb) {
                        w0[kb] = L4[(DB_K4 >> 2) + (wave * MTM + kb) * 64 + lane];
                        w1[kb] = L4[(DB_K4 >> 2) + ((wave + 4) * MTM + kb) * 64 + lane];
                        w2[kb] = L4[(DB_K4 >> 2) + ((third ? wave + 8 : wave) * MTM + kb) * 64 + lane];
                    }
                    __syncthreads();
                    DTL(40 * t + 28);
                    float4 bf[MTM];
#pragma unroll
                    for (int kb = 0; kb < MTM; ++kb) bf[kb] = L4[(DB_YF >> 2) + kb * 64 + lane];
                    f32x4 a00, a01, a10, a11;
                    mfma_v0(a00, w0[0].x, bf[0].x); mfma_v0(a10, w1[0].x, bf[0].x);
                    mfma_v0(a01, w0[0].y, bf[0].y); mfma_v0(a11, w1[0].y, bf[0].y);
                    mfma_v(a00, w0[0].z, bf[0].z); mfma_v(a10, w1[0].z, bf[0].z);
                    mfma_v(a01, w0[0].w, bf[0].w); mfma_v(a11, w1[0].w, bf[0].w);
#pragma unroll
                    for (int kb = 1; kb < MTM; ++kb) {
                        mfma_v(a00, w0[kb].x, bf[kb].x); mfma_v(a10, w1[kb].x, bf[kb].x);
                        mfma_v(a01, w0[kb].y, bf[kb].y); mfma_v(a11, w1[kb].y, bf[kb].y);
                        mfma_v(a00, w0[kb].z, bf[kb].z); mfma_v(a10, w1[kb].z, bf[kb].z);
                        mfma_v(a01, w0[kb].w, bf[kb].w); mfma_v(a11, w1[kb].w, bf[kb].w);
                    }
                    // (the fence directly behind the stream: accumulators that are live across a branch get copied by the compiler, and a copy
                    // two states behind an MFMA reads what the matrix pipe is still writing -- tools/mfma_hazard_check.py)
                    DU_FENCE4(a00, a01, a10, a11);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (H1: the resets issued in P3 have landed before a payload of this evaluation is stored)
                    du_st(g, vb, gP + wave * 1024, a00 + a01);
                    du_st(g, vb, gP + (wave + 4) * 1024, a10 + a11);
                    if (third) {
                        f32x4 a20, a21;
                        mfma_v0(a20, w2[0].x, bf[0].x); mfma_v0(a21, w2[0].y, bf[0].y);
                        mfma_v(a20, w2[0].z, bf[0].z); mfma_v(a21, w2[0].w, bf[0].w);
#pragma unroll
                        for (int kb = 1; kb < MTM; ++kb) {
                            mfma_v(a20, w2[kb].x, bf[kb].x); mfma_v(a21, w2[kb].y, bf[kb].y);
                            mfma_v(a20, w2[kb].z, bf[kb].z); mfma_v(a21, w2[kb].w, bf[kb].w);
                        }
                        DU_FENCE2(a20, a21);
                        du_st(g, vb, gP + (wave + 8) * 1024, a20 + a21);
                    }
                } else {
                float4 wf[2][MTM];
#pragma unroll
                for (int kb = 0; kb < MTM; ++kb) wf[0][kb] = L4[(DB_K4 >> 2) + (wave * MTM + kb) * 64 + lane];     // P4's first weights
                __syncthreads();
                DTL(40 * t + 28);
                // ================= P4: partial g = K0[H_c,:]^T y for the dim tiles wave, wave+4, wave+8 =================
                {
                    float4 bf[MTM];
#pragma unroll
                    for (int kb = 0; kb < MTM; ++kb) bf[kb] = L4[(DB_YF >> 2) + kb * 64 + lane];
#pragma unroll
                    for (int mi = 0; mi < 3; ++mi) {
                        const int mt = wave + 4 * mi;
                        if (mt < DU_KBD) {
                            if (mi < 2 && mt + 4 < DU_KBD) {
#pragma unroll
                                for (int kb = 0; kb < MTM; ++kb) wf[(mi + 1) & 1][kb] = L4[(DB_K4 >> 2) + ((mt + 4) * MTM + kb) * 64 + lane];
                            }
                            const float4 (&w4)[MTM] = wf[mi & 1];
                            f32x4 a0, a1;
                            mfma_v0(a0, w4[0].x, bf[0].x); mfma_v0(a1, w4[0].y, bf[0].y);
                            mfma_v(a0, w4[0].z, bf[0].z); mfma_v(a1, w4[0].w, bf[0].w);
#pragma unroll
                            for (int kb = 1; kb < MTM; ++kb) {
                                mfma_v(a0, w4[kb].x, bf[kb].x); mfma_v(a1, w4[kb].y, bf[kb].y);
                                mfma_v(a0, w4[kb].z, bf[kb].z); mfma_v(a1, w4[kb].w, bf[kb].w);
                            }
                            DU_FENCE2(a0, a1);
                            du_st(g, vb, gP + mt * 1024, a0 + a1);
                        }
                    }
                }
                }
                if (PRED) du_anchor(g, t, lane);
                if (dp.dbg & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                DTL(40 * t + 29);
            }
        }
#ifdef NOCF_ACC
        acc_dump();
#endif
    }
}

#ifdef NOCF_ACC
#undef DTL
#define DTL(id) do { } while (0)                   // (the lap counters exist in the forward only)
#endif
#include "nocf_duo_bwd.inc"

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int du_env_int(const char* name, int dflt) { return nocf_env_int(name, dflt); }      // (cached: nocf_kernels.hip)

long duo_rows_per_launch(void) { return 32L * 16 * DU_NTMAX; }

// G: members per group of the forward (8: the form of rounds 3-4; 16: the fine form, DuoCfg); the adjoint always runs with 8
// (129 ... 256 hidden units: four members of 64, zero-padded to 256, the forward only -- G is then ignored; 257 ... 512: padded to 512; the
// adjoint and the tape: exactly 512)
static int make_duo_plan(int d, int m_real, int nTh, int r, int n_agents, long n, DuoPlan* out, bool bwd = false, bool dw = false, int G = DU_G) {
    if (nTh != 2 || m_real <= 128 || m_real > 64 * DU_G || d + 1 > DU_DP || r > 16 || r < 1 || n < 1 || n_agents > 64 || n_agents < 1) return NOCF_E_SHAPE;
    if ((bwd || dw) && m_real != 64 * DU_G) return NOCF_E_SHAPE;
    const int m = m_real <= 256 ? 256 : 64 * DU_G;
    if (m == 256) G = 4;
    else if ((G != DU_G && G != DU_GMAX) || (G != DU_G && (bwd || dw))) return NOCF_E_SHAPE;
    DuoPlan dp;
    memset(&dp, 0, sizeof(dp));
    dp.d = d; dp.D1 = d + 1; dp.r = r; dp.nAg = n_agents; dp.G = G; dp.KBM = m / 16; dp.mReal = m_real;
    const long ntiles = (n + 15) / 16;
    dp.ngroups = (int)std::min<long>((dw ? 16 : 32) * DU_G / G, ntiles);     // (512 workgroups fill the chip: 32 groups of 16, 16 groups of 32 (dw, fine form))
    dp.dw = dw ? 1 : 0;
    dp.NT = (int)((ntiles + dp.ngroups - 1) / dp.ngroups);
    if (dp.NT > (m == 256 ? 2 : DU_NTMAX)) return NOCF_E_SHAPE;       // (four own samples per member and tile: LDS for two tiles)
    dp.hN = 1.0f;
    if (bwd && r > 10) return NOCF_E_SHAPE;                                        // (the adjoint keeps 10 rows of A in LDS)
    const int fine = G == DU_GMAX ? 1024 : 0;                                      // (the waves' partial sums: see the kernel)
    // (adjoint: + the column-sum cells of the epilogues, 2 x 64 floats behind role A's carve and 64 behind role B's)
    const int ldsA = bwd ? DAB_T + 2 * dp.NT * DSB_STRIDE + 128 : DA_T + fine + 4 * DPW_WORDS + (16 / G) * dp.NT * DS_STRIDE, ldsB = DB_END + (bwd ? 64 : 4 * DPW_WORDS);
    dp.ldsFloats = std::max(std::max(ldsA, ldsB), dw ? DC_END : 0);
#ifdef NOCF_ACC
    if (!bwd) dp.ldsFloats += 256;                                                 // (the lap counters of the diagnostic build)
#endif
    if ((size_t)dp.ldsFloats * 4 > 80 * 1024) return NOCF_E_LDS;                // two workgroups per CU
    long o = 0;                                                                    // floats
    dp.oPlan = o; o += 256;
    dp.oErr = o; o += 64;                                                          // (uint index == float index)
    dp.oXcc = o; o += 32 * 16;
    dp.oCen = o; o += 8 * 520;                                                    // CU census of the role map (uints)
    o += 32 * 32;                                                                  // progress counters of the weight-gradient roles: [group][2] at oCen + 8 * 520, 64 B apart
    const long gw = m / 64;                                                        // (image sizes depend on the width only)
    const long nW = gw * 4 * (m / 16) * 64, nK1 = gw * 4 * DU_KBD * 64, nK4 = gw * DU_KBD * 4 * 64;   // float4s
    dp.oW2 = o / 4; o += nW * 4;
    dp.oW3 = o / 4; o += nW * 4;
    dp.oK1 = o / 4; o += nK1 * 4;
    dp.oK4 = o / 4; o += nK4 * 4;
    dp.oA = o; o += 16 * DU_DP;
    dp.oVec = o; o += 3 * m;
    dp.oCW = o; o += DU_DP;
    o = (o + 63) / 64 * 64;
    dp.oX = o;
    dp.xStride = duo_x_layout(dp.NT, nullptr, bwd, G, dp.KBM);
    *out = dp;
    return 0;
}

static size_t duo_ws_bytes_of(const DuoPlan& dp) { return (size_t)(dp.oX + (long)dp.ngroups * dp.xStride) * sizeof(float); }

int duo_workspace_bytes(int d, int m, int nTh, int r, int n_agents, long n, size_t* bytes) {
    DuoPlan dp, db;
    const int rc = make_duo_plan(d, m, nTh, r, n_agents, std::min<long>(n, duo_rows_per_launch()), &dp);
    if (rc) return rc;
    size_t b = duo_ws_bytes_of(dp);
    if (make_duo_plan(d, m, nTh, r, n_agents, std::min<long>(n, duo_rows_per_launch()), &db, true) == 0) b = std::max(b, duo_ws_bytes_of(db));   // (the adjoint's exchange area is larger)
    if (make_duo_plan(d, m, nTh, r, n_agents, std::min<long>(n, duo_rows_per_launch()), &db, false, false, DU_GMAX) == 0) b = std::max(b, duo_ws_bytes_of(db));   // (the fine form's partial-gradient area too)

    if (bytes) *bytes = b;
    return 0;
}

template <int PD, bool REC, bool ZF, int GM, int KBMT = DU_KBM, int MODE = 0>
static const void* duo_fn() { return reinterpret_cast<const void*>(rollout_duo_kernel<PD, REC, ZF, GM, KBMT, MODE>); }
template <int GM, int MODE = 0>
static const void* duo_pick(bool c2, bool rec, bool zf) {
    return c2 ? (rec ? duo_fn<2, true, false, GM, DU_KBM, MODE>() : (zf ? duo_fn<2, false, true, GM, DU_KBM, MODE>() : duo_fn<2, false, false, GM, DU_KBM, MODE>()))
              : (rec ? duo_fn<3, true, false, GM, DU_KBM, MODE>() : (zf ? duo_fn<3, false, true, GM, DU_KBM, MODE>() : duo_fn<3, false, false, GM, DU_KBM, MODE>()));
}

// Which form for a launch of n rows?  NOCF_DUO_G = 8 / 16 forces one; otherwise the fine form where the batch has at most 16 tiles, i.e.
// where it has ONE tile per group (measured on the MI355X, profiles/r5/05_proxy_table.txt: 128 rows 2.35 ms against 2.95, 256 rows 2.77 against
// 2.95; with several tiles per group the two roles' tiles overlap on every SIMD, the MFMA and vector pipes exclude each other, and twice the
// workgroups per tile mean twice the epilogue / gather instructions per tile: 512 rows 3.9 against 3.3, 1024 rows 7.9 against 5.2)
static int duo_pick_G(long n) {
    const int forced = du_env_int("NOCF_DUO_G", 0);
    if (forced == DU_G || forced == DU_GMAX) return forced;
    return (n + 15) / 16 <= 16 ? DU_GMAX : DU_G;
}

int duo_launch(const NocfPhi* phi, const DevProb& pb, const RollArgs& ra_in, float* ws, size_t ws_bytes, hipStream_t st,
               const unsigned** errp, int debug, hipEvent_t ev0, hipEvent_t ev1) {
    if (pb.kind == NOCF_PROB_QUADCOPTER || (ra_in.zFull && ra_in.sAll)) return 1;
    const bool narrow = phi->m <= 256;                             // four members per group (DuoCfg): evaluation, intermediates, the recording forward
    const bool padded = phi->m != 256 && phi->m != 64 * DU_G;      // (zero-padded widths: evaluation and intermediates -- the records have the real width's row length)
    if ((narrow || padded) && ra_in.tapeSc) return 1;              // (no tape: the split-role ADJOINT exists for m = 512 only)
    if (padded && ra_in.sAll) return 1;
    if (ra_in.act && (long)ra_in.n * 512 * 4 >= (1L << 31)) return 1;      // (the record's per-evaluation blocks are addressed with 32-bit offsets)
    const int GM = narrow ? 4 : duo_pick_G(ra_in.n);
    // (512 workgroups per launch: 32 groups of 16, 16 of 32, 64 of 8; up to four tiles per group, two with four own samples per member)
    const long chunk = narrow ? duo_rows_per_launch() : duo_rows_per_launch() * DU_G / GM;
    DuoPlan dp0;
    if (make_duo_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, std::min<long>(ra_in.n, chunk), &dp0, false, false, GM) != 0) return 1;
    if (ws_bytes < duo_ws_bytes_of(dp0)) return 1;
    const bool c2 = pb.kind == NOCF_PROB_CROSS2D;
    const bool rec = ra_in.sAll != nullptr;
    const bool zf = ra_in.zFull != nullptr;
    // (the ONE instantiation: one launch of 32 one-tile groups in the default geometry, i.e. every CU holds the two roles of one member -- the
    // case the flag is for; measured, tools/r5_ab.sh: 512 rows 3.33 -> 3.13 ms, but 300 / 384 rows (19 / 24 groups, no census pairing) 3.40 -> 3.49)
    const bool one = GM == DU_G && dp0.NT == 1 && dp0.ngroups == 32 && ra_in.n <= chunk;
    // (MODE 0 has no predictive-waiting code: it serves every launch sequence in which some launch has several tiles per group -- a single
    // launch of <= 2048 rows always has NT of its first chunk; a chunked call's LAST chunk may have one tile per group and then simply polls)
    const int mode = GM != DU_G ? 0 : (dp0.NT > 1 ? 0 : (one ? 1 : 2));
    const bool n1 = narrow && dp0.NT == 1 && ra_in.n <= chunk;      // 256-wide geometry, one tile per group: the instantiation with the two-part owner's step
    const void* fk = narrow ? (n1 ? (c2 ? (rec ? duo_fn<2, true, false, 4, 16, 1>() : (zf ? duo_fn<2, false, true, 4, 16, 1>() : duo_fn<2, false, false, 4, 16, 1>()))
                                        : (rec ? duo_fn<3, true, false, 4, 16, 1>() : (zf ? duo_fn<3, false, true, 4, 16, 1>() : duo_fn<3, false, false, 4, 16, 1>())))
                                  : (c2 ? (rec ? duo_fn<2, true, false, 4, 16>() : (zf ? duo_fn<2, false, true, 4, 16>() : duo_fn<2, false, false, 4, 16>()))
                                        : (rec ? duo_fn<3, true, false, 4, 16>() : (zf ? duo_fn<3, false, true, 4, 16>() : duo_fn<3, false, false, 4, 16>()))))
                            : (GM == DU_G ? (mode == 1 ? duo_pick<DU_G, 1>(c2, rec, zf) : (mode == 2 ? duo_pick<DU_G, 2>(c2, rec, zf) : duo_pick<DU_G, 0>(c2, rec, zf)))
                                          : duo_pick<DU_GMAX>(c2, rec, zf));
    const int wpg = 2 * GM;
    // residency: all 16 x ngroups workgroups spin on each other, so every one of them must be resident at once: two per CU
    // (256 registers per lane, <= 80 KB LDS).  The grid is checked against what the runtime says fits; the stream must be
    // otherwise idle (a concurrent kernel on another stream can take the CUs: the bounded polls then time out and the host raises).
    int dev = 0, cus = 0, perCU = 0;
    if (hipGetDevice(&dev) || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) return 1;
    const size_t ldsBytes0 = (size_t)dp0.ldsFloats * 4;
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024));
    if (e) return (int)e;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, fk, 256, ldsBytes0) != hipSuccess) return 1;
    const int grid0 = 8 * wpg * ((dp0.ngroups + 7) / 8);
    if ((long)perCU * cus < grid0) {
        if (debug) fprintf(stderr, "[nocf] duo kernel: grid %d does not fit (%d workgroups per CU x %d CUs)\n", grid0, perCU, cus);
        return 1;
    }
    DevPhi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev};
    for (long r0 = 0; r0 < ra_in.n; r0 += chunk) {
        const long cn = std::min<long>(chunk, ra_in.n - r0);
        DuoPlan dp;
        int rc = make_duo_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, cn, &dp, false, false, GM);
        if (rc) return rc;
        dp.cb = phi->cb;
        dp.fast = du_env_int("NOCF_DUO_FAST", 1);
        dp.mapmode = du_env_int("NOCF_DUO_MAP", 3);                 // bit 0: A / B of a member adjacent in the static map; bit 1: CU census (see the kernel)
        dp.dbg = (du_env_int("NOCF_DUO_DBG", 0) & 0xff) | ((du_env_int("NOCF_DUO_UDELAY", dp.NT == 1 ? 0 : 6) & 15) << 8) |
                 ((du_env_int("NOCF_DUO_PWSH", GM == DU_GMAX ? 1 : 0) & 3) << 12);     // (measured, tools/r5_knobs.sh: fine form 2.35 -> 2.32 / 2.78 -> 2.66 ms at 7/8 of the interval, default form 3.33 -> 3.41; 15/16 oversleeps everywhere)   // (measured: n = 1024 5.21 -> 5.15 ms with 6, nothing with one tile)
        dp.spin_max = du_env_int("NOCF_DUO_SPIN_MAX", 1000000);
        // (every chunk: the plan record changes with the chunk's rows; the same launch clears the error words / tables and fills the
        // exchange area with the sentinel)
        hipLaunchKernelGGL(duo_pack_kernel, dim3(1024), dim3(256), 0, st, dp, P, ws, r0 == 0 ? 1 : 0);
        e = hipGetLastError();
        if (e) return (int)e;
        RollArgs ra = ra_in;
        ra.x = ra_in.x + r0 * phi->d; ra.n = cn;
        if (ra.z_out) ra.z_out = ra_in.z_out + r0 * (phi->d + 4);
        if (ra.persample) ra.persample = ra_in.persample + r0 * 7;
        DuoRun rr{r0, ra_in.n};
        const DuoPlan* dpp = reinterpret_cast<const DuoPlan*>(ws + dp.oPlan);
        const size_t ldsBytes = (size_t)dp.ldsFloats * 4;
        if (debug) fprintf(stderr, "[nocf] duo kernel: rows %ld..%ld, %d groups x %d workgroups, %d tile(s) of 16 samples, LDS %zu B/workgroup, %d workgroups/CU fit\n",
                           r0, r0 + cn, dp.ngroups, wpg, dp.NT, ldsBytes, perCU);
        void* args[] = {(void*)&dpp, (void*)&pb, (void*)&ws, (void*)&ra, (void*)&rr};
        if (ev0 && r0 == 0) (void)hipEventRecord(ev0, st);
        e = hipLaunchKernel(fk, dim3(8 * wpg * ((dp.ngroups + 7) / 8)), dim3(256), args, ldsBytes, st);
        if (e) return (int)e;
        if (ev1 && r0 + chunk >= ra_in.n) (void)hipEventRecord(ev1, st);
        if (debug >= 2) {
            unsigned w[2] = {0, 0};
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(w, ws + dp.oErr, 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "[nocf] duo kernel: error word 0x%x, %u of %d workgroups paired by the CU census\n", w[0], w[1], wpg * dp.ngroups);
        }
    }
    *errp = reinterpret_cast<const unsigned*>(ws) + dp0.oErr;
    return 0;
}

// ---- the adjoint (nocf_duo_bwd.inc).  Same plan, same images, same residency rule as the forward.
template <int PD, bool DW>
static const void* duo_bwd_fn() { return reinterpret_cast<const void*>(rollout_duo_bwd_kernel<PD, DW>); }

int duo_bwd_launch(const NocfPhi* phi, const DevProb& pb, const DuoBwdHost& h, float* ws, size_t ws_bytes, hipStream_t st,
                   const unsigned** errp, int debug, hipEvent_t ev0, hipEvent_t ev1) {
    if (pb.kind == NOCF_PROB_QUADCOPTER) return 1;
    // dw: the kernel with the two weight-gradient roles (dK1 / dK0 accumulated in the kernel): 32 workgroups per group, 16 groups,
    // 1024 rows per launch
    const bool dw = h.dK1 != nullptr && h.dK0 != nullptr && h.dw_scratch != nullptr && du_env_int("NOCF_DUO_DW", 0) != 0 &&
                    h.dw_scratch_floats >= duo_dw_scratch_floats();
    const long chunk = dw ? duo_rows_per_launch() / 2 : duo_rows_per_launch();
    DuoPlan dp0;
    if (make_duo_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, std::min<long>(h.n, chunk), &dp0, true, dw) != 0) return 1;
    if (ws_bytes < duo_ws_bytes_of(dp0)) return 1;
    const void* fk = pb.kind == NOCF_PROB_CROSS2D ? (dw ? duo_bwd_fn<2, true>() : duo_bwd_fn<2, false>()) : (dw ? duo_bwd_fn<3, true>() : duo_bwd_fn<3, false>());
    int dev = 0, cus = 0, perCU = 0;
    if (hipGetDevice(&dev) || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) return 1;
    const size_t ldsBytes0 = (size_t)dp0.ldsFloats * 4;
    hipError_t e = hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024));
    if (e) return (int)e;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, fk, 256, ldsBytes0) != hipSuccess) return 1;
    const int wpg = dw ? 32 : 16;
    const int grid0 = 8 * wpg * ((dp0.ngroups + 7) / 8);
    if ((long)perCU * cus < grid0) {
        if (debug) fprintf(stderr, "[nocf] duo adjoint kernel: grid %d does not fit (%d workgroups per CU x %d CUs)\n", grid0, perCU, cus);
        return 1;
    }
    DevPhi P{phi->K0, phi->b0, phi->K, phi->b, phi->w, phi->A, phi->cw, phi->cb_dev};
    const int nstage = (h.stepper == NOCF_RK4) ? 4 : 1;
    if (!h.csum && !h.Wb) return 1;
    if (h.csum) {                                                  // the launches' groups write their partials one behind the other; rows nobody writes stay 0
        if (h.csum_floats < duo_bwd_colsum_floats(h.n)) return 1;
        e = hipMemsetAsync(h.csum, 0, h.csum_floats * sizeof(float), st);
        if (e) return (int)e;
    }
    long gbase = 0;
    for (long r0 = 0; r0 < h.n; r0 += chunk) {
        const long cn = std::min<long>(chunk, h.n - r0);
        DuoPlan dp;
        int rc = make_duo_plan(phi->d, phi->m, phi->nTh, phi->r, pb.nAgents, cn, &dp, true, dw);
        if (rc) return rc;
        dp.cb = phi->cb;
        dp.fast = du_env_int("NOCF_DUO_FAST", 1);
        dp.mapmode = du_env_int("NOCF_DUO_MAP", 3);
        dp.dbg = (du_env_int("NOCF_DUO_DBG", 0) & 0xff) | ((du_env_int("NOCF_DUO_BWD_UDELAY", 0) & 15) << 8);
        dp.spin_max = du_env_int("NOCF_DUO_SPIN_MAX", 1000000);
        hipLaunchKernelGGL(duo_pack_kernel, dim3(1024), dim3(256), 0, st, dp, P, ws, r0 == 0 ? 1 : 0);
        e = hipGetLastError();
        if (e) return (int)e;
        DuoBwdArgs ba;
        ba.sAll = h.s_all; ba.zT = h.z_final; ba.hs = h.hs; ba.tape = h.tape; ba.tapeU1 = h.tapeU1; ba.tapeSc = h.tapeSc;
        ba.R = ((long)h.nt * nstage + 1) * h.n; ba.n = cn; ba.nt = h.nt; ba.nstage = nstage;
        ba.a0 = h.a0; ba.a3 = h.a3; ba.a4 = h.a4; ba.a5 = h.a5; ba.inv_n = h.inv_n;
        ba.Y = h.Y; ba.Ab = h.Ab; ba.Wb = h.Wb; ba.Qb = h.Qb; ba.Ob = h.Ob; ba.Gb = h.Gb; ba.lam0 = h.lam0;
        ba.stamps = h.stamps;
        ba.dK1p = dw ? h.dw_scratch : nullptr;
        ba.dK0p = dw ? h.dw_scratch + (size_t)2 * 16 * 512 * 512 : nullptr;
        if (h.csum && (size_t)(gbase + dp.ngroups) * 3 * 512 > h.csum_floats) return 1;       // (cannot happen with duo_bwd_colsum_floats(n) floats: checked, not assumed)
        ba.csum = h.csum ? h.csum + (size_t)gbase * 3 * 512 : nullptr;
        gbase += dp.ngroups;
        DuoRun rr{r0, h.n};
        const DuoPlan* dpp = reinterpret_cast<const DuoPlan*>(ws + dp.oPlan);
        const size_t ldsBytes = (size_t)dp.ldsFloats * 4;
        if (debug) fprintf(stderr, "[nocf] duo adjoint kernel: rows %ld..%ld, %d groups x 16 workgroups, %d tile(s) of 16 samples, LDS %zu B/workgroup, %d workgroups/CU fit\n",
                           r0, r0 + cn, dp.ngroups, dp.NT, ldsBytes, perCU);
        void* args[] = {(void*)&dpp, (void*)&pb, (void*)&ws, (void*)&ba, (void*)&rr};
        if (ev0 && r0 == 0) (void)hipEventRecord(ev0, st);
        e = hipLaunchKernel(fk, dim3(8 * wpg * ((dp.ngroups + 7) / 8)), dim3(256), args, ldsBytes, st);
        if (e) return (int)e;
        if (dw) {                                                  // the groups' partial sums of this launch -> dK1 / dK0 (fixed order)
            hipLaunchKernelGGL(duo_dw_reduce_kernel, dim3(1024), dim3(256), 0, st, ba.dK1p, ba.dK0p, dp.ngroups, phi->d + 1, h.dK1, h.dK0, r0 > 0 ? 1 : 0);
            e = hipGetLastError();
            if (e) return (int)e;
        }
        if (ev1 && r0 + chunk >= h.n) (void)hipEventRecord(ev1, st);
    }
    if (h.dw_done) *h.dw_done = dw ? 1 : 0;
    *errp = reinterpret_cast<const unsigned*>(ws) + dp0.oErr;
    return 0;
}

size_t duo_dw_scratch_floats(void) { return (size_t)2 * 16 * 512 * (512 + DU_DP); }
// [groups of all launches][3][512]: at most 32 groups per 512 rows (the launches of the kernel with the weight-gradient roles take 512 rows and 16 groups)
size_t duo_bwd_colsum_floats(long n) { return (size_t)((n + 511) / 512) * 32 * 3 * 512; }
