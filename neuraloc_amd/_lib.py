"""ctypes binding of libnocf.so (include/nocf.h).  There is NO fallback: if the HIP library
is missing or a call fails, the product path raises."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NOCF_LIB_PATH: diagnostics only (e.g. the -DNOCF_STAMPS build); the default is the in-tree production library
LIB_PATH = os.environ.get("NOCF_LIB_PATH") or os.path.join(_HERE, "csrc", "libnocf.so")

NOCF_RK4, NOCF_RK1 = 4, 1
PROB_CROSS2D, PROB_SWARMTRAJ, PROB_QUADCOPTER = 0, 1, 2
OBS_CODES = {None: 0, "softcorridor": 1, "hardcorridor": 2, "blocks": 3}

_ERRORS = {-1: "NOCF_E_NULL (required pointer is NULL)", -2: "NOCF_E_SHAPE (d/m/nTh/n/nt unsupported)",
           -3: "NOCF_E_PROB (unknown problem kind / obstacle)", -4: "NOCF_E_WORKSPACE (workspace too small)",
           -5: "NOCF_E_STEPPER", -6: "NOCF_E_LDS (network does not fit the LDS plan)"}

fp = C.POINTER(C.c_float)


class NocfPhi(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("nTh", C.c_int32), ("r", C.c_int32),
                ("K0", C.c_void_p), ("b0", C.c_void_p), ("K", C.c_void_p), ("b", C.c_void_p),
                ("w", C.c_void_p), ("A", C.c_void_p), ("cw", C.c_void_p), ("cb", C.c_float), ("cb_dev", C.c_void_p)]


class NocfProb(C.Structure):
    _fields_ = [("kind", C.c_int32), ("obstacle", C.c_int32), ("n_agents", C.c_int32), ("training", C.c_int32),
                ("r", C.c_double), ("alph_Q", C.c_double), ("alph_W", C.c_double),
                ("mass", C.c_double), ("grav", C.c_double), ("xtarget", C.c_void_p)]


class NocfPhi64(C.Structure):
    _fields_ = [("d", C.c_int32), ("m", C.c_int32), ("nTh", C.c_int32), ("r", C.c_int32),
                ("K0", C.c_void_p), ("b0", C.c_void_p), ("K", C.c_void_p), ("b", C.c_void_p),
                ("w", C.c_void_p), ("A", C.c_void_p), ("cw", C.c_void_p), ("cb_dev", C.c_void_p)]


class NocfProb64(C.Structure):
    _fields_ = [("kind", C.c_int32), ("obstacle", C.c_int32), ("n_agents", C.c_int32), ("training", C.c_int32),
                ("r", C.c_double), ("alph_Q", C.c_double), ("alph_W", C.c_double),
                ("mass", C.c_double), ("grav", C.c_double), ("xtarget", C.c_void_p)]


_lib = None


def lib():
    """the loaded library; raises RuntimeError when it was not built (python __graft_entry__.py)"""
    global _lib
    if _lib is not None:
        watch_env(_lib)
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"neuraloc_amd: HIP library {LIB_PATH} is missing. Build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                           "There is no CPU fallback.")
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


# shapes (d, m, nTh, r, agents) the shipped library already specialises (FIXED_SHAPES / FIXED_SHAPES_TRAIN in nocf_kernels.hip)
_BUILTIN_SHAPES = {(150, 512, 2, 10, 50), (12, 128, 2, 10, 1), (40, 32, 2, 10, 20), (60, 32, 2, 10, 30), (96, 32, 2, 10, 32),
                   (4, 16, 2, 5, 2), (4, 32, 2, 5, 2), (24, 32, 2, 10, 12), (8, 32, 2, 9, 4), (12, 32, 2, 10, 6),
                   (16, 32, 2, 10, 8), (20, 32, 2, 10, 10)}
_jit_libs = {}
_jit_started = set()


def _jit_mode():
    """NOCF_JIT: "0" never; "1" compile the shape's library NOW (blocking, about a minute) and use it; "auto" (default) use a cached
    per-shape library when there is one, otherwise start its compilation in a child process, keep the generic instantiation for THIS
    process (one process never changes kernels for a shape mid-run) and let the next process find the cache."""
    v = os.environ.get("NOCF_JIT", "auto").strip().lower()
    return "0" if v in ("", "0", "off", "no") else ("1" if v in ("1", "on", "yes") else "auto")


def _jit_paths(key):
    csrc = os.path.dirname(os.path.abspath(LIB_PATH))
    out_dir = os.path.join(csrc, "jit")
    so = os.path.join(out_dir, "libnocf_d%d_m%d_t%d_r%d_a%d.so" % key)
    src = os.path.join(csrc, "nocf_kernels.hip")
    inc = os.path.join(os.path.dirname(os.path.dirname(csrc)), "include")
    deps = [src, os.path.join(inc, "nocf.h")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".inc") or f.endswith(".h")]
    return csrc, out_dir, so, src, inc, deps


def _jit_cmd(key, out):
    import shutil
    csrc, _, _, src, inc, _ = _jit_paths(key)
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return None
    return [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNOCF_JIT_ONLY",
            "-DNOCF_XS_D=%d" % key[0], "-DNOCF_XS_M=%d" % key[1], "-DNOCF_XS_T=%d" % key[2], "-DNOCF_XS_R=%d" % key[3],
            "-DNOCF_XS_A=%d" % key[4], "-I" + inc, "-I" + csrc, "-o", out, src]


def _jit_fresh(key):
    """path of the cached per-shape library if it exists and is newer than every source it was built from"""
    _, _, so, _, _, deps = _jit_paths(key)
    try:
        return so if os.path.getmtime(so) >= max(os.path.getmtime(f) for f in deps) else None
    except OSError:
        return None


def _profiler_attached():
    """a profiler / tool library is preloaded into this process (rocprofv3 --pmc and friends): its library would initialise the GPU
    inside every child process too, and a chain of execs of GPU-initialised processes (sh -> hipcc -> clang -> lld) is what must never
    happen on a shared pool"""
    pre = os.environ.get("LD_PRELOAD", "")
    return any(t in pre for t in ("rocprof", "roctracer", "rocprofiler")) or any(
        os.environ.get(k) for k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))


def _compiler_env():
    """the environment the compiler child gets: this process's, without anything that preloads a tool library or steers a profiler"""
    return {k: v for k, v in os.environ.items()
            if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE") and not k.startswith(("ROCP_", "ROCPROFILER_", "ROCTRACER_", "ROCPROF_"))}


def _jit_start_background(key):
    """one compilation at a time per cache directory (lock file), detached child with a scrubbed environment, atomic rename on success;
    never raises; does nothing under a profiler"""
    import subprocess
    import sys
    import time
    if key in _jit_started:
        return False
    _jit_started.add(key)
    if _profiler_attached():
        return False
    try:
        _, out_dir, so, _, _, _ = _jit_paths(key)
        os.makedirs(out_dir, exist_ok=True)
        lock = os.path.join(out_dir, ".compiling")
        try:
            if time.time() - os.path.getmtime(lock) > 1800:
                os.unlink(lock)                                   # a compilation that died
        except OSError:
            pass
        tmp = "%s.%d.tmp" % (so, os.getpid())
        cmd = _jit_cmd(key, tmp)
        if cmd is None:
            return False
        try:                                                      # a compilation of these sources that failed (its log is kept): not again
            _, _, _, _, _, deps = _jit_paths(key)
            if not os.path.exists(so) and os.path.getmtime(so + ".log") >= max(os.path.getmtime(f) for f in deps):
                return False
        except OSError:
            pass
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
            os.close(fd)
        except OSError:
            return False                                          # somebody (another rank, another shape) is compiling: next time
        import shlex
        sh = "%s > %s 2>&1 && mv -f %s %s; rm -f %s %s" % (" ".join(shlex.quote(c) for c in cmd), shlex.quote(so + ".log"),
                                                           shlex.quote(tmp), shlex.quote(so), shlex.quote(tmp), shlex.quote(lock))
        subprocess.Popen(["/bin/sh", "-c", sh], stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                         start_new_session=True, env=_compiler_env())
        print("[neuraloc_amd] shape d=%d m=%d nTh=%d r=%d agents=%d has no specialised kernels in the shipped library: compiling them in the "
              "background (about a minute, cached under csrc/jit/; this process keeps the generic instantiation, NOCF_JIT=1 waits for "
              "the compiler instead, NOCF_JIT=0 switches this off)" % key, file=sys.stderr, flush=True)
        return True
    except Exception:                                             # noqa: BLE001 -- an optimisation must never take the rollout down
        return False


def lib_for(d, m, nTh, r, n_agents, fwd=False):
    """The library to run a rollout of this shape with.  fwd: the call is a forward rollout (evaluation, intermediates, the recording
    forward) -- two-layer networks of 129 ... 512 hidden units take the split-role kernel there, which only the SHIPPED library contains
    (a per-shape build has the per-tile kernels only: it serves such a network's adjoint).  The shipped one specialises the shapes of the reference's checkpoints and
    initProb defaults (_BUILTIN_SHAPES) and takes any other shape with its generic instantiation; such a shape gets its own library --
    the same source compiled by hipcc with -DNOCF_XS_* (the plan as a compile-time constant: 1.3-1.8x on the tile kernels, no
    scratch), cached under csrc/jit/ -- automatically: see _jit_mode.  Same C ABI, same entry points."""
    key = (int(d), int(m), int(nTh), int(r), int(n_agents))
    mode = _jit_mode()
    if mode == "0" or key in _BUILTIN_SHAPES or "NOCF_LIB_PATH" in os.environ:
        return lib()
    if fwd and key[2] == 2 and 128 < key[1] <= 512 and os.environ.get("NOCF_DUO", "1") not in ("0",) and not _duo_fallback["on"]:
        return lib()                              # (after duo_guard's in-process fallback the per-shape per-tile build serves the forward again)
    L = _jit_libs.get(key)
    if L is not None:
        watch_env(L)
        return L
    if mode == "auto" and key[1] == 512 and key[2] == 2:
        # wide two-layer networks run on the split-role kernel of the shipped library (nocf_duo.hip), which a per-shape build does not
        # contain: specialising the tile kernels for them would be a step down
        L = _jit_libs[key] = lib()
        return L
    so = _jit_fresh(key)
    if so is None and mode == "1":
        import subprocess
        import sys
        _, out_dir, so, _, _, _ = _jit_paths(key)
        os.makedirs(out_dir, exist_ok=True)
        tmp = "%s.%d.tmp" % (so, os.getpid())
        cmd = _jit_cmd(key, tmp)
        if cmd is None:
            raise RuntimeError("NOCF_JIT=1 but hipcc was not found (set HIPCC)")
        print("[neuraloc_amd] NOCF_JIT=1: specialising the kernels for shape d=%d m=%d nTh=%d r=%d agents=%d (about a minute, once)" % key,
              file=sys.stderr, flush=True)
        if _profiler_attached():
            raise RuntimeError("NOCF_JIT=1 under a profiler: compile the shape's library first (run once without the profiler), "
                               "or set NOCF_JIT=0")
        subprocess.check_call(cmd, env=_compiler_env())
        os.replace(tmp, so)                       # atomic: ranks that compile the same shape at once do not clash
    if so is None:                                # auto, nothing cached yet
        _jit_start_background(key)
        L = _jit_libs[key] = lib()
        return L
    L = _jit_libs[key] = _bind(C.CDLL(so))
    return L


def _bind(L):
    L.nocf_version.restype = C.c_int
    if hasattr(L, "nocf_set_knob"):
        L.nocf_set_knob.restype = C.c_int
        L.nocf_set_knob.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
        if _duo_fallback["on"]:
            L.nocf_set_knob(b"NOCF_DUO", 0, 0)
    L.nocf_last_rollout_kernel.restype = C.c_char_p
    L.nocf_workspace_bytes.restype = C.c_size_t
    L.nocf_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.nocf_rollout_workspace_bytes.restype = C.c_size_t
    L.nocf_rollout_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64]
    L.nocf_ctrl_dim.restype = C.c_int
    L.nocf_ctrl_dim.argtypes = [C.POINTER(NocfProb), C.c_int32]
    L.nocf_rollout_f32.restype = C.c_int
    L.nocf_rollout_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_void_p, C.c_int64,
                                   C.c_double, C.c_double, C.c_int32, C.c_int32, fp,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(L, "nocf_rollout_means_f32"):
        L.nocf_rollout_means_f32.restype = C.c_int
        L.nocf_rollout_means_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_void_p, C.c_int64,
                                             C.c_double, C.c_double, C.c_int32, C.c_int32, fp,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(L, "nocf_rollout_segments_f32"):
        L.nocf_rollout_segments_f32.restype = C.c_int
        L.nocf_rollout_segments_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_void_p, C.c_int64,
                                                C.c_int32, C.c_int64, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, fp,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(L, "nocf_segments_supported"):
        L.nocf_segments_supported.restype = C.c_int
        L.nocf_segments_supported.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb)]
    L.nocf_small_grad_floats.restype = C.c_int64
    L.nocf_small_grad_floats.argtypes = [C.c_int32, C.c_int32]
    L.nocf_rollout_bwd_small_f32.restype = C.c_int
    L.nocf_rollout_bwd_small_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_int64, C.c_int32, C.c_int32, C.c_double,
                                             fp, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(L, "nocf_rollout_bwd_mid_f32"):
        L.nocf_mid_grad_rows.restype = C.c_int64
        L.nocf_mid_grad_rows.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64]
        L.nocf_rollout_bwd_mid_f32.restype = C.c_int
        L.nocf_rollout_bwd_mid_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_int64, C.c_int32, C.c_int32, C.c_double,
                                               fp, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(L, "nocf_phi_value_bwd_f32"):
        L.nocf_phi_value_bwd_f32.restype = C.c_int
        L.nocf_phi_value_bwd_f32.argtypes = [C.POINTER(NocfPhi), C.c_void_p, C.c_int64, C.c_void_p] + [C.c_void_p] * 9 + [C.c_void_p, C.c_size_t, C.c_void_p]
        L.nocf_phi_grad_bwd_f32.restype = C.c_int
        L.nocf_phi_grad_bwd_f32.argtypes = [C.POINTER(NocfPhi), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p] + [C.c_void_p] * 9 + [C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_contract_f32.restype = C.c_int
    L.nocf_contract_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_cost_means_f32.restype = C.c_int
    L.nocf_cost_means_f32.argtypes = [C.c_void_p, fp, C.c_void_p, C.c_void_p]
    L.nocf_rollout_record_f32.restype = C.c_int
    L.nocf_rollout_record_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_void_p, C.c_int64,
                                          C.c_double, C.c_double, C.c_int32, C.c_int32, fp,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_rollout_bwd_f32.restype = C.c_int
    L.nocf_rollout_bwd_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_int64, C.c_int32, C.c_int32, C.c_double,
                                       fp, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_void_p] * 11 + \
                                      [C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_activation_record_floats.restype = C.c_size_t
    L.nocf_activation_record_floats.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32]
    L.nocf_rollout_record_act_f32.restype = C.c_int
    L.nocf_rollout_record_act_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_void_p, C.c_int64,
                                              C.c_double, C.c_double, C.c_int32, C.c_int32, fp,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32),
                                              C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_rollout_bwd_act_f32.restype = C.c_int
    L.nocf_rollout_bwd_act_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_int64, C.c_int32, C.c_int32, C.c_double,
                                           fp, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_void_p] * 11 + \
                                          [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(L, "nocf_tape_floats"):
        L.nocf_tape_floats.restype = C.c_size_t
        L.nocf_tape_floats.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32]
        L.nocf_rollout_tape_f32.restype = C.c_int
        L.nocf_rollout_tape_f32.argtypes = L.nocf_rollout_record_act_f32.argtypes
        L.nocf_rollout_bwd_tape_f32.restype = C.c_int
        L.nocf_rollout_bwd_tape_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_int64, C.c_int32, C.c_int32,
                                                fp, C.c_double] + [C.c_void_p] * 11 + \
                                               [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)] + \
                                               [C.c_void_p, C.c_size_t, C.c_void_p]
        L.nocf_dw_scratch_floats.restype = C.c_size_t
        if hasattr(L, "nocf_bwd_colsum_floats"):                  # (a per-shape library cached by an older build has no such entry: train.py then streams the dw rows)
            L.nocf_bwd_colsum_floats.restype = C.c_size_t
            L.nocf_bwd_colsum_floats.argtypes = [C.c_int64]
            L.nocf_rollout_bwd_tape_sums_f32.restype = C.c_int
            L.nocf_rollout_bwd_tape_sums_f32.argtypes = [C.POINTER(NocfPhi), C.POINTER(NocfProb), C.c_int64, C.c_int32, C.c_int32,
                                                         fp, C.c_double] + [C.c_void_p] * 10 + \
                                                        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)] + \
                                                        [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.nocf_poison_if_failed_f32.restype = C.c_int
        L.nocf_poison_if_failed_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    for name in ("nocf_phi_grad_f32", "nocf_phi_forward_f32"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(NocfPhi), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_prob_eval_f32.restype = C.c_int
    L.nocf_prob_eval_f32.argtypes = [C.POINTER(NocfProb), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.nocf_profile_begin.restype = C.c_int
    L.nocf_profile_end.restype = C.c_int
    L.nocf_profile_end.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.nocf_colsum_f32.restype = C.c_int
    L.nocf_colsum_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_workspace_bytes_f64.restype = C.c_size_t
    L.nocf_workspace_bytes_f64.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.nocf_rollout_f64.restype = C.c_int
    L.nocf_rollout_f64.argtypes = [C.POINTER(NocfPhi64), C.POINTER(NocfProb64), C.c_void_p, C.c_int64,
                                   C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(L, "nocf_rollout_bwd_f64"):
        L.nocf_rollout_record_f64.restype = C.c_int
        L.nocf_rollout_record_f64.argtypes = [C.POINTER(NocfPhi64), C.POINTER(NocfProb64), C.c_void_p, C.c_int64,
                                              C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.nocf_rollout_bwd_f64.restype = C.c_int
        L.nocf_rollout_bwd_f64.argtypes = [C.POINTER(NocfPhi64), C.POINTER(NocfProb64), C.c_int64, C.c_int32, C.c_int32, C.c_double,
                                           C.POINTER(C.c_double), C.c_double, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_void_p] * 11 + \
                                          [C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_prob_eval_f64.restype = C.c_int
    L.nocf_prob_eval_f64.argtypes = [C.POINTER(NocfProb64), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.nocf_phi_f64.restype = C.c_int
    L.nocf_phi_f64.argtypes = [C.POINTER(NocfPhi64), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.nocf_debug_reload_env.restype = None
    L.nocf_last_rollout_status_async.restype = C.c_int
    L.nocf_last_rollout_status_async.argtypes = [C.c_void_p, C.c_void_p]
    L.nocf_debug_set_stamp_buffer.restype = C.c_int
    L.nocf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    L.nocf_selftest_mfma.restype = C.c_int
    L.nocf_selftest_mfma.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    return L


# ---- asynchronous rollout status (include/nocf.h: nocf_last_rollout_status_async).  The split-role kernel reports a timed-out
# exchange through an error word; its copy is enqueued behind every rollout and looked at WITHOUT synchronising: at the next call
# into this package, in check_errors() (neuraloc_amd.check_errors: what a driver calls after it has synchronised anyway), or
# when the status has already arrived.  A failed rollout's outputs are NaN in any case.
_pending = []          # [event, pinned host word, description]
_free_words = []


_env_sig = None


def watch_env(L):
    """NOCF_ENV_WATCH=1 (tests): the library caches its NOCF_* knobs; drop the cache when one of them has changed since the last call"""
    global _env_sig
    if os.environ.get("NOCF_ENV_WATCH", "0") in ("", "0"):
        return
    sig = (id(L),) + tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("NOCF_")))
    if sig != _env_sig:
        _env_sig = sig
        L.nocf_debug_reload_env()


def track_rollout_status(L, device, what):
    word = _free_words.pop() if _free_words else torch.zeros(1, dtype=torch.int32).pin_memory()
    rc = L.nocf_last_rollout_status_async(C.c_void_p(word.data_ptr()), stream_ptr(device))
    if rc == 1:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        _pending.append((ev, word, what))
    else:
        _free_words.append(word)
        if rc != 0:
            check(rc, "nocf_last_rollout_status_async")


def check_errors(sync=False):
    """Raise RuntimeError if a rollout whose status has arrived (sync=True: of every rollout launched so far, after waiting for it)
    reported a timed-out exchange.  Called at the start of every call into the package; call it with sync=True where results are
    consumed on the host."""
    while _pending:
        ev, word, what = _pending[0]
        if sync:
            ev.synchronize()
        elif not ev.query():
            return
        _pending.pop(0)
        code = int(word[0])
        word[0] = 0
        _free_words.append(word)
        if code != 0:
            for _e, w, _w in _pending:
                _free_words.append(w)
            _pending.clear()
            raise RuntimeError(f"{what}: the weight-stationary rollout kernel timed out waiting for another workgroup (error word "
                               f"0x{code:x}); its outputs are NaN.  The kernel needs the whole GPU: all of its workgroups must be resident at "
                               "once, so nothing else may run on the device while it does (another stream or process took compute units), "
                               "or set NOCF_DUO=0 in the environment BEFORE the first rollout of the process to use the per-tile kernel "
                               "(the library reads its knobs once)")


# ---- probation of the weight-stationary kernels.  They need every one of their workgroups resident at once; on a GPU that this
# process shares (a second rank or another job on the same device) their exchange times out.  A timed-out rollout is normally reported
# asynchronously (above), i.e. after its NaN results have been handed out.  So the first launches of such a kernel in a process are checked
# SYNCHRONOUSLY: if one timed out, the split-role kernels are switched off for the rest of the process (NOCF_DUO=0, the per-tile kernels
# need no co-residency) and the caller repeats the call -- a fresh launch in this process, nothing is re-executed.  After the probation
# (a device that is ours) a later timeout raises as before.
_duo_probation = {"left": int(os.environ.get("NOCF_DUO_PROBATION", "3"))}


def duo_guard(L, what):
    """call right behind track_rollout_status; True: the launch timed out during probation, the split-role kernels are now off, repeat the call.
    Only the launch that was JUST tracked can trigger the fallback: older pending rollouts are drained first and a failure among them raises
    as it would have anyway (their NaN results are already out).  The switch is a library-level override (nocf_set_knob), not an environment
    variable: children of this process (compiler, self-launched ranks) do not inherit it; every rank prints its own message."""
    if _duo_probation["left"] <= 0 or not L.nocf_last_rollout_kernel().decode().startswith("rollout_duo"):
        return False
    _duo_probation["left"] -= 1
    mine = _pending.pop() if _pending else None             # the entry track_rollout_status just added
    try:
        check_errors(sync=True)                             # earlier launches: a failure here is theirs and is raised as such
    finally:
        if mine is not None:
            _pending.append(mine)
    try:
        check_errors(sync=True)
        return False
    except RuntimeError as ex:
        if os.environ.get("NOCF_DUO_FALLBACK", "1") in ("0", ""):
            raise
        import sys
        rank = os.environ.get("RANK")
        print(f"[neuraloc_amd]{'' if rank is None else ' rank ' + rank}: {what}: {str(ex).split(';')[0]}; the GPU seems to be shared -- switching "
              "this process to the per-tile kernels (NOCF_DUO=0 as a library override) and repeating the call", file=sys.stderr, flush=True)
        for lib_ in [_lib] + list(_jit_libs.values()):
            if lib_ is not None and hasattr(lib_, "nocf_set_knob"):
                lib_.nocf_set_knob(b"NOCF_DUO", 0, 0)
        _duo_fallback["on"] = True
        _duo_probation["left"] = 0
        return True


_duo_fallback = {"on": False}


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERRORS.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP error {rc}")


def require_device_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the OCflow hot path runs only on an MI355X (ROCm 'cuda' device); "
                           "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} has dtype {t.dtype}: the HIP path computes in fp32 only")
    return t.contiguous()


def require_device_f64(t, name):
    """double precision (the reference's --prec double): every tensor of the call must be float64 on the MI355X"""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the OCflow hot path runs only on an MI355X (ROCm 'cuda' device); "
                           "there is no CPU fallback")
    if t.dtype != torch.float64:
        raise RuntimeError(f"{name} has dtype {t.dtype} in a double-precision call: convert the network, the problem and the "
                           "states together (net.to(torch.float64), initProb(..., cvt) with a float64 cvt), like the reference's --prec double")
    return t.contiguous()


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
