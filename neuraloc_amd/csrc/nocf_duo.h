// nocf_duo.h -- host interface between nocf_kernels.hip (C ABI, dispatch) and nocf_duo.hip (the split-role
// weight-stationary rollout kernel).  Internal to libnocf.so; nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "nocf.h"
#include "nocf_dev.h"

// 0 when a network of this shape on a point-agent problem can take the duo kernel (then *bytes = workspace it needs for a
// rollout of n rows), else an NOCF_E_* code
int duo_workspace_bytes(int d, int m, int nTh, int r, int n_agents, long n, size_t* bytes);

// Launches the rollout of ra.n rows (in chunks of at most duo_rows_per_launch() rows) on `st`.  Returns 0 and sets
// *errp (device address of the launch's error word: non-zero after a timed-out exchange) when the kernel was launched,
// 1 when the shape / problem / workspace / residency does not qualify (nothing was launched: the caller takes another
// kernel), or a HIP / NOCF_E_* error code.  ev0 / ev1 (optional) are recorded on the stream right around the rollout kernel.
int duo_launch(const NocfPhi* phi, const DevProb& pb, const RollArgs& ra, float* ws, size_t ws_bytes, hipStream_t st,
               const unsigned** errp, int debug, hipEvent_t ev0, hipEvent_t ev1);

long duo_rows_per_launch(void);

// The adjoint on the same layout (nocf_duo_bwd.inc): same return convention as duo_launch.  All pointers are device memory; the
// tape is the one the recording forward (RollArgs::act / tapeU1 / tapeSc with the terminal block) wrote for the same rows.
struct DuoBwdHost {
    const float *s_all, *z_final, *hs, *tape, *tapeU1, *tapeSc;
    long n; int nt, stepper;
    float a0, a3, a4, a5, inv_n;
    float *Y, *Ab, *Wb, *Qb, *Ob, *Gb, *lam0;
    unsigned long long* stamps;
    // optional: dK1 [m][m] and dK0 [m][d+1] accumulated in the kernel (the weight-gradient roles), with a scratch buffer of
    // duo_dw_scratch_floats() floats for the groups' partial sums; *dw_done = 1 when that kernel ran
    float *dK1, *dK0, *dw_scratch; size_t dw_scratch_floats; int* dw_done;
    // optional: the column sums of the dw rows, qbar and obar formed in the kernel's epilogues -- [G][3][m] partials, G = csum_floats / (3 m) >=
    // duo_bwd_colsum_floats(n) / (3 m), zeroed here, to be summed over G in index order by the caller.  With it Wb may be null (not streamed).
    float* csum; size_t csum_floats;
};
size_t duo_dw_scratch_floats(void);
size_t duo_bwd_colsum_floats(long n);
int duo_bwd_launch(const NocfPhi* phi, const DevProb& pb, const DuoBwdHost& h, float* ws, size_t ws_bytes, hipStream_t st,
                   const unsigned** errp, int debug, hipEvent_t ev0, hipEvent_t ev1);

// NOCF_* knob, read from the environment once and cached (nocf_kernels.hip); nocf_debug_reload_env() drops the cache
int nocf_env_int(const char* name, int dflt);
