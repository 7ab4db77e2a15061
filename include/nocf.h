/*
 * nocf.h -- C ABI of libnocf.so: the MI355X (gfx950) OCflow rollout hot path.
 *
 * The reference (donken/NeuralOC) is pure Python and has no FFI; this header is the
 * boundary a maintainer binds with ctypes (stub in INTEGRATION.md).  Each entry point
 * names the reference interface it replaces (file:line into donken/NeuralOC).
 *
 * Conventions
 *   - every pointer marked "device" is a HIP device pointer on the current device;
 *     all tensors are dense row-major fp32 unless stated otherwise
 *   - functions enqueue work on `stream` (a hipStream_t passed as void*, 0 = default
 *     stream) and return without synchronising; nothing is allocated or freed -- with ONE opt-in exception since version 113: with
 *     NOCF_LANE_ONE=1 in the environment the first rollout of a small network (m <= 32: one wavefront per sample) on a device allocates
 *     256 bytes of device memory for the life of the process: the self-resetting ticket words (one per stream, at most 64) with which that
 *     kernel's last workgroup forms the cost sums itself instead of a second launch (measured slower on the MI355X: off by default)
 *   - return value: 0 ok; <0 argument error (NOCF_E_*); >0 a hipError_t
 *   - no exceptions, no aborts.  Process-global state: the diagnostic environment knobs (NOCF_LANE, NOCF_FIXED, NOCF_DUO,
 *     NOCF_MONO, NOCF_DEBUG ...) are read from the environment once, at first use, and cached (nocf_debug_reload_env drops the
 *     cache); the measurement hooks nocf_profile_begin/_end, nocf_last_rollout_kernel, nocf_last_rollout_status_async and
 *     nocf_debug_set_* keep a list of HIP events / the last launch's name and error-word address / a diagnostic buffer pointer
 *     and are NOT thread-safe.  The compute entry points themselves keep no state between calls (the caller owns the workspace)
 */
#ifndef NOCF_H
#define NOCF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NOCF_VERSION 113            /* major*100 + minor */

#define NOCF_E_NULL      (-1)       /* required pointer is NULL                     */
#define NOCF_E_SHAPE     (-2)       /* d/m/nTh/n/nt out of the supported range      */
#define NOCF_E_PROB      (-3)       /* unknown problem kind / obstacle combination  */
#define NOCF_E_WORKSPACE (-4)       /* workspace too small (see nocf_workspace_bytes) */
#define NOCF_E_STEPPER   (-5)       /* stepper is neither NOCF_RK4 nor NOCF_RK1     */
#define NOCF_E_LDS       (-6)       /* network too deep/wide for the 160 KiB LDS plan */

/* stepper, src/OCflow.py:46-49 ("rk4" / "rk1") */
#define NOCF_RK4 4
#define NOCF_RK1 1

/* problem classes, src/problem/{Cross2D,SwarmTraj,Quadcopter}.py */
#define NOCF_PROB_CROSS2D    0
#define NOCF_PROB_SWARMTRAJ  1
#define NOCF_PROB_QUADCOPTER 2

/* obstacle strings of the problem constructors */
#define NOCF_OBS_NONE         0
#define NOCF_OBS_SOFTCORRIDOR 1    /* Cross2D.py:43-48   */
#define NOCF_OBS_HARDCORRIDOR 2    /* Cross2D.py:49-52   */
#define NOCF_OBS_BLOCKS       3    /* SwarmTraj.py:46-50 */

/* The value network Phi (src/Phi.py:57-87).  Pointers are the tensors of the reference
 * state_dict, unmodified: N.layers.0.weight, N.layers.0.bias, N.layers.{1..nTh-1}.weight
 * stacked, ...bias stacked, w.weight, A, c.weight; cb = c.bias[0]. */
typedef struct NocfPhi {
    int32_t d;          /* space dimension; inputs are (x,t) in R^{d+1}              */
    int32_t m;          /* hidden width                                               */
    int32_t nTh;        /* number of ResNet layers, >= 2 (src/Phi.py:25-27)           */
    int32_t r;          /* rows of A = min(10, d+1) (src/Phi.py:75)                   */
    const float* K0;    /* device [m, d+1]                                            */
    const float* b0;    /* device [m]                                                 */
    const float* K;     /* device [nTh-1, m, m]                                       */
    const float* b;     /* device [nTh-1, m]                                          */
    const float* w;     /* device [m]                                                 */
    const float* A;     /* device [r, d+1]                                            */
    const float* cw;    /* device [d+1]                                               */
    float cb;           /* c.bias as a host value (used when cb_dev is NULL)          */
    const float* cb_dev;/* optional: device address of c.bias; when set the kernels   */
                        /* read the bias there and the caller needs no device-to-host */
                        /* copy (a synchronisation per call) to fill `cb`             */
} NocfPhi;

/* One problem object (the attributes the reference's calcLHQW/calcGradpH/calcCtrls read). */
typedef struct NocfProb {
    int32_t kind;       /* NOCF_PROB_*                                                */
    int32_t obstacle;   /* NOCF_OBS_*                                                 */
    int32_t n_agents;   /* d / agentDim                                               */
    int32_t training;   /* prob.training: masks and thresholds differ */
    /* Python-side floats are doubles; the kernels derive their fp32 constants from these the
     * way torch does when a Python scalar meets an fp32 tensor (thresholds such as 3.2*r or
     * 2.0+r are formed in double, then rounded once). */
    double r;           /* agent radius                                               */
    double alph_Q;
    double alph_W;
    double mass;        /* Quadcopter only                                            */
    double grav;        /* Quadcopter only                                            */
    const float* xtarget; /* device [d]                                               */
} NocfProb;

int nocf_version(void);

/* measurement hook: name of the rollout kernel the last nocf_rollout_f32 / nocf_rollout_record_f32 call of this process
 * launched ("rollout_duo_kernel", "rollout_mono_kernel", "rollout_kernel<shape-specialised>", "rollout_kernel<generic>",
 * "rollout_lane_kernel", or "none"); a static string, not thread-safe (bench.py labels its roofline with it) */
const char* nocf_last_rollout_kernel(void);

/* Asynchronous status of the last rollout / adjoint call on the calling thread's CURRENT DEVICE (per-call state is kept per device:
 * threads driving different GPUs do not see each other's; a backward that a framework runs on a worker thread shares the forward's).
 * The split-role kernels' workgroups wait for each other with bounded polls; when one times out
 * (the GPU was shared with another kernel, so that not all workgroups were resident) the kernel sets an error word, finishes, and
 * every output of the call -- persample rows, the cost sums, z_out, zFull, ctrlFull -- is turned into NaN on the stream (the training
 * tape and s_all are not: call nocf_poison_if_failed_f32 on what is derived from them).  This call enqueues a
 * 4-byte device-to-host copy of that word into `host_word` (pinned host memory) on `stream`; once the stream has reached it,
 * *host_word != 0 means the rollout failed (0x3000 + the exchange kind that timed out).  Returns 1 when a copy was enqueued, 0 when
 * the last rollout kernel has no such word (*host_word is set to 0), or an error code.  The Python layer raises RuntimeError from it. */
int nocf_last_rollout_status_async(uint32_t* host_word, void* stream);

/* The library reads its NOCF_* environment knobs once (first use) and caches them; this drops the cache so that the next call reads
 * the environment again (tests that switch kernels between calls; the Python layer calls it when NOCF_ENV_WATCH=1). */
void nocf_debug_reload_env(void);
/* Override one NOCF_* knob for THIS process's library instance (clear != 0: drop the override again).  Unlike the environment it is not
 * inherited by child processes (compiler children, self-launched ranks) and survives nocf_debug_reload_env: the in-process fallback of a
 * process that shares its GPU (neuraloc_amd/_lib.py: duo_guard) uses it to switch the weight-stationary kernels off. */
int nocf_set_knob(const char* name, int32_t value, int32_t clear);

/* bytes of scratch `workspace` a call with these shapes needs (packed weight images) */
size_t nocf_workspace_bytes(int32_t d, int32_t m, int32_t nTh);

/* bytes a nocf_rollout_f32 call over n samples can use: >= nocf_workspace_bytes; the extra room holds the
 * activation-exchange buffers of the weight-sliced group kernel (without it the per-tile kernel runs) */
size_t nocf_rollout_workspace_bytes(int32_t d, int32_t m, int32_t nTh, int64_t n);

/* number of control components per sample: d (Cross2D, SwarmTraj) or 4*n_agents (Quadcopter) */
int nocf_ctrl_dim(const NocfProb* prob, int32_t d);

/*
 * OCflow -- replaces src/OCflow.py:7-95 (driver), :104-140 (ocOdefun), :143-184 (steppers)
 * together with Phi.getGrad (src/Phi.py:99-138), Phi.forward (:91-96) and the problem's
 * calcLHQW / calcGradpH / calcCtrls for every stage, fused into one launch.
 *
 *   x          device [n, d]          initial states (not modified)
 *   t0,t1,nt   tspan and number of steps; h=(t1-t0)/nt, times advance in double like the reference
 *   alph       host  [6]              only [0],[3],[4],[5] are used (SURVEY 8a note 7)
 *   z_out      device [n, d+4]        final z = [x(T) | L | HJt | Q | W]           (nullable)
 *   persample  device [n, 7]          per-sample [L, G, HJt, HJfin, HJgrad, Q, W] = the
 *                                     reference's noMean=True list, src/OCflow.py:66-76 (nullable)
 *   cost_sums  device [8]             sums over the n samples of the 7 columns above, then n.
 *                                     Deterministic (fixed-order, fp64-accumulated) reduction.
 *                                     The caller forms means / Jc (and all-reduces first when
 *                                     the batch is sharded over GPUs).                (nullable)
 *   zFull      device [nt+1, n, d+4]  intermediates=True: z after every step, slot 0 = z0.
 *                                     TIME-MAJOR; the reference's [n, d+4, nt+1] is the
 *                                     permute(1,2,0) view of it.                       (nullable)
 *   ctrlFull   device [nt+1, n, a]    controls, a = nocf_ctrl_dim(); slot 0 stays zero and slot
 *                                     k+1 uses (z_{k+1}, t_k) exactly like src/OCflow.py:51-55.
 *                                     Required iff zFull is given.
 */
int nocf_rollout_f32(const NocfPhi* phi, const NocfProb* prob,
                     const float* x, int64_t n,
                     double t0, double t1, int32_t nt, int32_t stepper,
                     const float* alph,
                     float* z_out, float* persample, float* cost_sums,
                     float* zFull, float* ctrlFull,
                     void* workspace, size_t workspace_bytes, void* stream);

/*
 * nocf_rollout_f32 that also forms the means: cost_means device [8] = the 7 batch means [L, G, HJt, HJfin, HJgrad, Q, W] and Jc
 * (src/OCflow.py:80-90; what nocf_cost_means_f32 computes from cost_sums, same arithmetic), written by the launch that reduces the
 * per-sample table -- for callers with no all-reduce between the sums and the means (one device, the whole batch): one launch less per call.
 * cost_means nullable; it needs cost_sums.
 */
int nocf_rollout_means_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                           double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                           float* z_out, float* persample, float* cost_sums, float* cost_means, float* zFull, float* ctrlFull,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * Several rollouts that differ only in their start time and step count, in ONE launch (round 5): the second segments of a shock sweep.
 * The reference's shocked rollout (src/plotter.py:815-824, driven by evalOC.py:113-122) is OCflow on [0, t_s] with int(t_s nt) steps, the
 * shock added to the end state, and OCflow on [t_s, 1] with 1 + nt - int(t_s nt) steps; a sweep over shock times (BASELINE config 5) repeats
 * that per t_s.  Here rows [k * rows_per_seg, (k + 1) * rows_per_seg) of x are segment k: integrated over [t0s[k], t1] with nts[k] steps of
 * h_k = (t1 - t0s[k]) / nts[k], exactly what nocf_rollout_f32(x_k, ..., t0s[k], t1, nts[k], ...) computes for those rows (bit for bit: same
 * kernel, same per-tile arithmetic), but the launch covers all segments' tiles at once -- 9 x 512 rows fill the chip where 512 rows occupy an
 * eighth of it.
 *   nseg <= 16 segments; rows_per_seg a multiple of 16; (nseg - 1) * rows_per_seg < n <= nseg * rows_per_seg (a ragged last segment).
 *   t0s, nts       HOST arrays of nseg entries (copied into the kernel arguments)
 *   slot0s         HOST array (nullable = zeros): the first time slot of segment k in zFull / ctrlFull.  With slot0s[k] = int(t_s nt) + 1 the
 *                  caller's buffer holds, per row, the reference's concatenation cat(traj1, traj2) (src/plotter.py:822): it fills the slots
 *                  in front of slot0s[k] with the unshocked segment
 *   zFull          device [max(slot0s + nts) + 1, n, d+4], time-major as in nocf_rollout_f32; segment k writes slots slot0s[k] .. + nts[k] only
 *   ctrlFull       likewise                      cost_sums   device [nseg, 8]: one row of sums per segment
 * Returns NOCF_E_SHAPE when the network / problem has no one-CU weight-stationary kernel (singlequad has; nothing is launched then, and
 * the caller runs the segments one by one).
 */
int nocf_rollout_segments_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                              int32_t nseg, int64_t rows_per_seg, const double* t0s, double t1, const int32_t* nts, const int32_t* slot0s,
                              int32_t stepper, const float* alph, float* z_out, float* persample, float* cost_sums, float* zFull, float* ctrlFull,
                              void* workspace, size_t workspace_bytes, void* stream);

/* 1 when nocf_rollout_segments_f32 has a kernel for this network / problem (host-side, nothing is launched), 0 otherwise: lets a caller
 * decide before it allocates and fills the segments' buffers (neuraloc_amd/shock.py). */
int nocf_segments_supported(const NocfPhi* phi, const NocfProb* prob);

/*
 * The last lines of OCflow (src/OCflow.py:80-90): out[0..6] = cost_sums[0..6] / cost_sums[7] (the batch means
 * [L, G, HJt, HJfin, HJgrad, Q, W]) and out[7] = Jc = L + alph[0] G + alph[3] HJt + alph[4] HJfin + alph[5] HJgrad.
 * cost_sums is what nocf_rollout_f32 wrote (after the all-reduce when the batch is sharded).  alph: host [6].
 */
int nocf_cost_means_f32(const float* cost_sums, const float* alph, float* out, void* stream);

/*
 * Training: the pair below replaces `Jc = OCflow(...); Jc.backward()` of trainOC.py:172-173
 * for every problem class and any depth nTh >= 2 whose activations fit the LDS (NOCF_E_LDS otherwise).
 *
 * nocf_rollout_record_f32 = nocf_rollout_f32 that also records the stage input s=[x,t] of every RK evaluation:
 *   s_all  device [nt*nstage, n, d+1]   (nstage = 4 for rk4, 1 for rk1);  z_out is required.
 *
 * nocf_rollout_bwd_f32 runs the exact adjoint of the discrete scheme.  It does not reduce the parameter
 * gradients itself: it streams, for every sample and evaluation, the vectors whose outer products they are,
 *   rows = (nt*nstage + 2) * n        (the two extra blocks are the terminal grad-Phi and Phi terms)
 *   Y, Ob, Wb                  device [rows, m]
 *   V, Ab, Qb, U0              device [nTh-1, rows, m]   one slab per residual layer i = 1..nTh-1
 *   Gb, Sx                     device [rows, d+1]
 *   (the kernel does not write the last n rows of Y, V, Ab, Gb: the caller zeroes those)
 *   PHIb                       device [n]         cotangent of Phi(x_T, t1);  lam0 device [n, d] = dJc/dx0 (nullable)
 * and the caller contracts them with library GEMMs:
 *   dK0 = Y'Gb + Ob'Sx   db0 = sum Ob   dK_i = V_i'Ab_i + Qb_i'U0_i   db_i = sum Qb_i   dw = sum Wb
 *   dc.weight = sum Gb + sum PHIb s_T
 *   dc.bias = sum PHIb   dM = Gb'Sx + 1/2 sum PHIb s_T s_T'   dA = A (dM + dM')
 *   hs      device [nt] fp32 step sizes as the forward used them: (float)((tk+h)-tk), tk += h in double
 *   inv_n   1 / (global batch size) -- the means of src/OCflow.py:80-86 run over all shards
 */
int nocf_rollout_record_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                            double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                            float* z_out, float* persample, float* cost_sums, float* s_all,
                            void* workspace, size_t workspace_bytes, void* stream);

int nocf_rollout_bwd_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                         const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                         float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                         float* PHIb, float* lam0, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Activation record (optional, training of two-layer networks that the split-role kernel takes: m = 512, point-agent problems).
 * nocf_rollout_record_act_f32 = nocf_rollout_record_f32 that also stores, for every RK evaluation and sample, the activations of
 * grad Phi -- u0 = sigma(o), tanh(o), tanh(q), a = w + hN K1' v ([nt*nstage, n, m] each) and grad Phi ([nt*nstage, n, d+1]) -- into
 *   act_rec  device [nocf_activation_record_floats(...)] floats (four sections of nt*nstage*n*m, one of nt*nstage*n*(d+1)); NULL: no record
 *   recorded host int32: 1 when the kernel this call launched wrote the record (only the split-role kernel does), else 0
 * nocf_rollout_bwd_act_f32 = nocf_rollout_bwd_f32 that, given a record the forward launch WROTE (recorded == 1), loads these
 * activations instead of re-running grad Phi's forward sweep at every evaluation (four of its eight GEMM phases and the weights they
 * stream; the terminal evaluation is still recomputed).  act_rec NULL: exactly nocf_rollout_bwd_f32.
 * nocf_activation_record_floats returns 0 for shapes without a recording kernel (pass NULL then).
 * (src/OCflow.py:7-95 forward, trainOC.py:172-173 backward: what autograd keeps as saved tensors of the unrolled graph)
 */
size_t nocf_activation_record_floats(int32_t d, int32_t m, int32_t nTh, int64_t n, int32_t nt, int32_t stepper);
int nocf_rollout_record_act_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                                double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                                float* z_out, float* persample, float* cost_sums, float* s_all, float* act_rec, int32_t* recorded,
                                void* workspace, size_t workspace_bytes, void* stream);
int nocf_rollout_bwd_act_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                             const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                             float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                             float* PHIb, float* lam0, const float* act_rec, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Training tape + split-role adjoint (round 4; the path trainOC.py:172-174 takes for the networks the split-role kernel runs: m = 512,
 * nTh = 2, point-agent problems).  The tape is the activation record above with the TERMINAL evaluation as one more block and three
 * scalars per evaluation -- everything autograd would keep of the unrolled graph of src/OCflow.py:7-95 -- laid out as
 *     R = (nt*nstage + 1) * n rows, block e < nt*nstage = RK evaluation e, block nt*nstage = the terminal evaluation (src/OCflow.py:58-64)
 *     [0, 4 R m)                  u0 | tanh(o) | tanh(q) | a          four sections [R, m]
 *     [4 R m, + R (d+1))          grad Phi                            [R, d+1]   (padded to a multiple of 4 floats)
 *     then n m                    u_1 of the terminal evaluation      [n, m]
 *     then 4 R                    (dPhi/dt - H, q, w, 0) per row; terminal block: (Phi - alph0 G, 0, 0, 0)
 * nocf_tape_floats: size of that buffer in floats, 0 when the shape has no tape-writing kernel (use the record calls above then).
 * nocf_rollout_tape_f32 = nocf_rollout_record_f32 with s_all of nt*nstage + 1 blocks (the last: [z(T), t1]) and the tape;
 *   *recorded = 1 when the launched kernel wrote the tape (else only the first nt*nstage blocks of s_all are written: fall back to
 *   nocf_rollout_bwd_f32).
 * nocf_rollout_bwd_tape_f32: the adjoint of the discrete scheme on the split-role layout.  Writes the row vectors whose outer products
 *   are the weight gradients, one row per (block, sample) as on the tape:
 *     Y = tanh(o).a, Ab = abar0, Wb = dw row, Qb = qbar, Ob = obar   device [R, m];   Gb = gbar   device [R, d+1]
 *   The value's rows of nocf_rollout_bwd_f32 (cotangent phib of Phi(z(T), T)) are NOT in them: the caller adds phib.tanh(q).w to Qb,
 *   phib.Y to Ob and phib.u_1 (tape) to Wb on the n rows of the terminal block.  Then, with the tape's U0 = u0, TH1 = tanh(q), Sx = s_all,
 *     dK0 = Y'Gb + Ob'Sx,  db0 = colsum Ob,  dK1 = diag(w) TH1'Ab + Qb'U0,  db1 = colsum Qb,  dw = colsum Wb,
 *     dc.weight = colsum Gb + phib'sT,  d(A'A) = Gb'Sx + (sT.phib)'sT / 2,  phib = alph4 sign(tape scalar of the terminal block) / n_total.
 *   lam0 device [n, d] = dJc/dx0 (nullable).  Returns NOCF_E_SHAPE when the shape / problem / residency does not qualify (nothing launched).
 *   A timed-out exchange is reported like the forward's (nocf_last_rollout_status_async); nocf_poison_if_failed_f32 turns a buffer
 *   into NaN on the stream if the last launch on this device failed (call it on the gradients before they are used).
 *   Weight-gradient roles (optional): with dK1 device [m, m], dK0 device [m, d+1] and dw_scratch device [nocf_dw_scratch_floats()]
 *   the two large weight gradients are accumulated IN THE KERNEL by two more role workgroups per member that consume the row streams
 *   behind progress counters (csrc/nocf_duo_bwd.inc): on return (*dw_done = 1) dK1 and dK0 hold the complete sums INCLUDING the value's
 *   rows (the caller must then not add phib.v / phib.y to Qb / Ob before using them for anything else; Wb's value rows and the column sums
 *   stay the caller's).  *dw_done = 0: the kernel without those roles ran (NOCF_DUO_DW=0, or NULL pointers) and dK1 / dK0 are untouched.
 */
size_t nocf_tape_floats(int32_t d, int32_t m, int32_t nTh, int64_t n, int32_t nt, int32_t stepper);
int nocf_rollout_tape_f32(const NocfPhi* phi, const NocfProb* prob, const float* x, int64_t n,
                          double t0, double t1, int32_t nt, int32_t stepper, const float* alph,
                          float* z_out, float* persample, float* cost_sums, float* s_all, float* tape, int32_t* recorded,
                          void* workspace, size_t workspace_bytes, void* stream);
int nocf_rollout_bwd_tape_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper,
                              const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                              const float* tape, float* Y, float* Ab, float* Wb, float* Qb, float* Ob, float* Gb, float* lam0,
                              float* dK1, float* dK0, float* dw_scratch, size_t dw_scratch_floats, int32_t* dw_done,
                              void* workspace, size_t workspace_bytes, void* stream);
size_t nocf_dw_scratch_floats(void);
/*
 * nocf_rollout_bwd_tape_sums_f32 = nocf_rollout_bwd_tape_f32 with the COLUMN SUMS formed in the kernel (round 5): the epilogues that hold the dw rows,
 * qbar and obar add them up (16 rows across a DPP row, then per wave in LDS; rows beyond the batch add nothing) and every group writes one partial:
 *     colsum   device [G, 3, m], G = colsum_floats / (3 m) >= nocf_bwd_colsum_floats(n) / (3 m):  (colsum Wb | colsum Qb | colsum Ob) per group of
 *              every launch, zero for the rest; the caller sums over G in index order (fixed order: deterministic).
 * The dw rows are then not streamed at all (no Wb: [R, m] floats less to allocate and to write) and db1 / db0 / dw need no pass over Qb / Ob / Wb.
 * As in nocf_rollout_bwd_tape_f32 the value's rows are not in them: the caller adds phib'(tanh(q).w), phib'Y and phib'u_1 of the terminal block's n rows.
 */
size_t nocf_bwd_colsum_floats(int64_t n);
int nocf_rollout_bwd_tape_sums_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper,
                                   const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                                   const float* tape, float* Y, float* Ab, float* Qb, float* Ob, float* Gb, float* lam0,
                                   float* dK1, float* dK0, float* dw_scratch, size_t dw_scratch_floats, int32_t* dw_done,
                                   float* colsum, size_t colsum_floats, void* workspace, size_t workspace_bytes, void* stream);
int nocf_poison_if_failed_f32(float* buf, int64_t count, void* stream);

/*
 * Parameter gradients of the stand-alone value call Phi(s) (src/Phi.py:91-96 under torch autograd: `net(x).backward(gout)`), any depth:
 * the rows whose outer products are the gradients, in the layout of nocf_rollout_bwd_f32 with nt = 0 (two blocks of n rows; the SECOND
 * block carries the value's rows, the first one is zero cotangents):  dK0 = Ob' Sx, db0 = colsum(Ob), dK_i = Qb_i' U0_i, db_i = colsum(Qb_i),
 * dw = colsum(Wb), dc.weight = gout' s, dc.bias = sum(gout), d(A'A) = 1/2 (s . gout)' s.  dPhi/ds is nocf_phi_grad_f32 times gout.
 *   s     device [n, d+1];   gout  device [n]: the cotangent of Phi(s), an INPUT
 *   Y, Ob, Wb  device [2 n, m];  V, Ab, Qb, U0  device [nTh-1, 2 n, m];  Gb, Sx  device [2 n, d+1]  (Y, V, Ab, Gb: second block zeroed by the caller)
 */
int nocf_phi_value_bwd_f32(const NocfPhi* phi, const float* s, int64_t n, float* gout,
                           float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * The vector-Jacobian product of the stand-alone gradient call (src/Phi.py:99-138 under torch autograd: `net.getGrad(x).backward(gbar)`), any depth:
 * sbar = (d grad Phi / d s)' gbar and, in the FIRST block of n rows of the streams (layout of nocf_rollout_bwd_f32 with nt = 0), the rows of the
 * parameter gradients:  dK0 = Y' Gb + Ob' Sx, db0 = colsum(Ob), dK_i = V_i' Ab_i + Qb_i' U0_i, db_i = colsum(Qb_i), dw = colsum(Wb),
 * dc.weight = colsum(Gb), d(A'A) = Gb' Sx  (Gb = gbar, Sx = s).
 *   s, gbar, sbar  device [n, d+1];   streams as in nocf_phi_value_bwd_f32 (2 n rows each; only the first n are written)
 */
int nocf_phi_grad_bwd_f32(const NocfPhi* phi, const float* s, int64_t n, const float* gbar, float* sbar,
                          float* Y, float* Ob, float* V, float* Ab, float* Qb, float* U0, float* Wb, float* Gb, float* Sx,
                          void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same adjoint for SMALL networks (nTh = 2, m <= 32, d+1 <= 32, Cross2D agents: the shapes the lane kernel of
 * nocf_rollout_f32 takes), one wavefront per sample with every weight-gradient row in registers: nothing is streamed and
 * nothing is left to contract.  Returns NOCF_E_SHAPE for any other shape (use nocf_rollout_bwd_f32 then).
 *   gpart  device [n, nocf_small_grad_floats(d, m)]: per-sample gradient vectors
 *          [dK0 (m x (d+1)) | db0 (m) | dK1 (m x m) | db1 (m) | dw (m) | dc.weight (d+1) | dc.bias (1) | dM ((d+1) x (d+1))],
 *          dM = d/d(A'A) (dA = A (dM + dM')); the caller sums them over the samples
 *   lam0   device [n, d] = dJc/dx0 (nullable)
 */
int64_t nocf_small_grad_floats(int32_t d, int32_t m);
int nocf_rollout_bwd_small_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                               const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                               float* gpart, float* lam0, void* stream);

/*
 * The same adjoint for MEDIUM two-layer networks (nTh = 2, 32 < m <= 128, d+1 <= 32, every problem class: the shapes the one-CU
 * weight-stationary kernel of nocf_rollout_f32 takes, e.g. singlequad; reference: trainOC.py:172-174 on src/OCflow.py:7-140).  One
 * workgroup per 16 samples keeps the network in its registers, takes grad Phi's activations from the forward's activation record
 * (act_rec of nocf_rollout_record_act_f32, requested one evaluation ahead; null or `recorded` = 0: it re-runs grad Phi at the recorded
 * stage inputs) and ACCUMULATES the weight gradients in the kernel (MFMA outer products with the samples on the k axis): no row
 * stream, no library GEMM.  Every workgroup writes one partial gradient vector; the caller adds the partial vectors (fixed order).
 *   nocf_mid_grad_rows   number of partial vectors for a batch of n rows (min(ceil(n / 16), 1024)), or 0 when the shape has no such kernel
 *                        (NOCF_E_SHAPE from the launch then: use nocf_rollout_bwd_act_f32)
 *   act_rec nullable: the activation record of the forward launch (only when that launch reported recorded = 1)
 *   gpart  device [gpart_rows, nocf_small_grad_floats(d, m)], the layout of nocf_rollout_bwd_small_f32
 *   lam0   device [n, d] = dJc/dx0 (nullable);   workspace: nocf_workspace_bytes (same as the forward's)
 */
int64_t nocf_mid_grad_rows(int32_t d, int32_t m, int32_t nTh, int32_t r, int32_t n_agents, int64_t n);
int nocf_rollout_bwd_mid_f32(const NocfPhi* phi, const NocfProb* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                             const float* alph, double inv_n, const float* s_all, const float* z_final, const float* hs,
                             const float* act_rec, float* gpart, int64_t gpart_rows, float* lam0, void* workspace, size_t workspace_bytes,
                             void* stream);

/*
 * C[m, n] (+)= sum_k A[k, 0..m) (x) B[k, 0..n) for small outputs (m, n <= 512, at most 64 tiles of 64 x 64) and very many rows: the contraction of the rows that
 * nocf_rollout_bwd_f32 streams into the weight gradients of a SMALL network (what torch autograd does with one mm per
 * parameter in the backward of trainOC.py:173).  Two launches, deterministic.  Wider networks contract with library GEMMs.
 *   A device [K, m], B device [K, n] row-major;  C device [m, n];  accumulate != 0 adds to C
 *   scratch device [scratch_floats]: 4096 floats per workgroup and 64x64 output tile (up to 1024 of them, one workgroup per
 *           >= 256 rows and tile)
 */
int nocf_contract_f32(const float* A, const float* B, int64_t K, int32_t m, int32_t n, float* C, int32_t accumulate,
                      float* scratch, size_t scratch_floats, void* stream);

/*
 * out[n] (+)= column sums of X [K, n]: the bias, w and c.weight gradients are the sums of the adjoint's rows over every sample and
 * evaluation (torch autograd's sum-to-size in the backward of trainOC.py:173).  Two launches, fixed order (deterministic), reads
 * X once at HBM speed.  scratch device [scratch_floats]: n floats per row slice (up to 2048 slices of >= 128 rows).
 */
int nocf_colsum_f32(const float* X, int64_t K, int32_t n, float* out, int32_t accumulate,
                    float* scratch, size_t scratch_floats, void* stream);

/* Phi.getGrad -- replaces src/Phi.py:99-138.  s: device [n, d+1] -> grad: device [n, d+1] */
int nocf_phi_grad_f32(const NocfPhi* phi, const float* s, int64_t n, float* grad,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Phi.forward -- replaces src/Phi.py:91-96.  s: device [n, d+1] -> value: device [n] */
int nocf_phi_forward_f32(const NocfPhi* phi, const float* s, int64_t n, float* value,
                         void* workspace, size_t workspace_bytes, void* stream);

/*
 * Problem physics on given (x, p) -- replaces calcLHQW / calcGradpH / calcCtrls of
 * src/problem/Cross2D.py:69-165, SwarmTraj.py:68-167, Quadcopter.py:65-174.
 *   x, p     device [n, d]
 *   lhqw     device [n, 4]   columns L, H, Q, W   (Q as the reference returns it: scaled by
 *                            alph_Q for Cross2D, un-scaled for SwarmTraj/Quadcopter)  (nullable)
 *   gradpH   device [n, d]                                                          (nullable)
 *   ctrls    device [n, a]                                                          (nullable)
 */
int nocf_prob_eval_f32(const NocfProb* prob, int32_t d, const float* x, const float* p, int64_t n,
                       float* lhqw, float* gradpH, float* ctrls, void* stream);

/*
 * Double precision -- the reference's `--prec double` (trainOC.py:76-79,96; evalOC.py:19,28-31): net, problem and states go
 * through .to(torch.float64) and the same OCflow (src/OCflow.py:7-95) runs.  Same arguments and layouts as nocf_rollout_f32 with
 * every buffer in double (alph: host [6] doubles; cost_sums: 7 column sums then n, as doubles).  Evaluation only (the adjoint is
 * fp32).  c.bias is read on the device (cb_dev is required).  workspace: nocf_workspace_bytes_f64(d, m, nTh) bytes.
 */
typedef struct NocfPhi64 {
    int32_t d, m, nTh, r;
    const double* K0;    /* device [m, d+1]      */
    const double* b0;    /* device [m]           */
    const double* K;     /* device [nTh-1, m, m] */
    const double* b;     /* device [nTh-1, m]    */
    const double* w;     /* device [m]           */
    const double* A;     /* device [r, d+1]      */
    const double* cw;    /* device [d+1]         */
    const double* cb_dev;/* device [1]           */
} NocfPhi64;

typedef struct NocfProb64 {
    int32_t kind, obstacle, n_agents, training;
    double r, alph_Q, alph_W, mass, grav;
    const double* xtarget; /* device [d] */
} NocfProb64;

size_t nocf_workspace_bytes_f64(int32_t d, int32_t m, int32_t nTh);

int nocf_rollout_f64(const NocfPhi64* phi, const NocfProb64* prob,
                     const double* x, int64_t n,
                     double t0, double t1, int32_t nt, int32_t stepper,
                     const double* alph,
                     double* z_out, double* persample, double* cost_sums,
                     double* zFull, double* ctrlFull,
                     void* workspace, size_t workspace_bytes, void* stream);

/*
 * Training in double precision (`trainOC.py --prec double`, trainOC.py:44,76-79,172-174): the recording forward and the adjoint of the
 * discrete RK scheme, every tensor a double.  Same semantics, row streams and contractions as nocf_rollout_record_f32 /
 * nocf_rollout_bwd_f32 (rows = (nt * nstage + 2) * n; the caller zeroes the last n rows of Y, V, Ab, Gb and contracts in double);
 * any depth that fits the LDS, all problem classes, rk4 / rk1.
 *   s_all  device [nt * nstage, n, d+1];   hs  device [nt] step sizes (tk + h) - tk as the forward formed them
 */
int nocf_rollout_record_f64(const NocfPhi64* phi, const NocfProb64* prob, const double* x, int64_t n,
                            double t0, double t1, int32_t nt, int32_t stepper, const double* alph,
                            double* z_out, double* persample, double* cost_sums, double* s_all,
                            void* workspace, size_t workspace_bytes, void* stream);
int nocf_rollout_bwd_f64(const NocfPhi64* phi, const NocfProb64* prob, int64_t n, int32_t nt, int32_t stepper, double t1,
                         const double* alph, double inv_n, const double* s_all, const double* z_final, const double* hs,
                         double* Y, double* Ob, double* V, double* Ab, double* Qb, double* U0, double* Wb, double* Gb, double* Sx,
                         double* PHIb, double* lam0, void* workspace, size_t workspace_bytes, void* stream);

/* Phi.forward (src/Phi.py:91-96) and Phi.getGrad (:99-138) in double: s device [n, d+1] -> value device [n] (nullable),
 * grad device [n, d+1] (nullable; at least one of the two) */
int nocf_phi_f64(const NocfPhi64* phi, const double* s, int64_t n, double* value, double* grad,
                 void* workspace, size_t workspace_bytes, void* stream);

/* calcLHQW / calcGradpH / calcCtrls in double: the arguments of nocf_prob_eval_f32 with double buffers */
int nocf_prob_eval_f64(const NocfProb64* prob, int32_t d, const double* x, const double* p, int64_t n,
                       double* lhqw, double* gradpH, double* ctrls, void* stream);

/* Measurement hooks (bench.py): between begin and end every nocf_rollout_f32 call records a pair
 * of HIP events on its launch stream immediately around the rollout kernel; end synchronises on
 * them and returns the summed kernel time and the number of launches.  Not thread-safe. */
int nocf_profile_begin(void);
int nocf_profile_end(double* total_ms, int32_t* launches);

/* Diagnostic builds (-DNOCF_STAMPS, libnocf_stamps.so) only: device buffer of 12 uint64 per
 * workgroup receiving per-phase shader-cycle totals.  The production library returns an error. */
int nocf_debug_set_stamp_buffer(void* device_buf);
/* diagnostic builds only: [8 waves][64 points] shader-clock timeline of one evaluation of the adjoint kernel */
int nocf_debug_set_timeline_buffer(void* device_buf);

/* layout probe used by the tests: D = sum_k A_k B_k through the same 4x4x1 MFMA tile code
 * the rollout uses.  a: device [4, K], b: device [K, 64] -> out: device [4, 64] */
int nocf_selftest_mfma(const float* a, const float* b, int32_t K, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NOCF_H */
