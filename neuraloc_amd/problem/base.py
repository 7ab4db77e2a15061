"""Shared host-side plumbing of the three problem classes: attribute storage (same names the
reference's duck-typed problem objects expose) and the HIP call for the physics."""
import ctypes as C

import torch

from .. import _lib


class _ProblemBase:
    KIND = None
    AGENT_DIM = None

    def _init_common(self, xtarget, obstacle, alph_Q, alph_W, r):
        self.xtarget = xtarget.squeeze()          # ocG assumes a 1-D target (src/OCflow.py:97-101)
        self.d = xtarget.numel()
        self.agentDim = self.AGENT_DIM
        self.obstacle = obstacle
        self.alph_Q = alph_Q
        self.alph_W = alph_W
        self.nAgents = self.d // self.agentDim
        self.r = r
        self.training = True                      # masks / thresholds differ between train and eval

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    # ---- boundary ---------------------------------------------------------------------
    def _c_struct(self, device):
        if self.obstacle not in _lib.OBS_CODES:
            raise ValueError(f"obstacle {self.obstacle!r} is not one the reference implements")
        xt = self.xtarget.detach().to(device=device, dtype=torch.float32).contiguous()
        st = _lib.NocfProb()
        st.kind, st.obstacle = self.KIND, _lib.OBS_CODES[self.obstacle]
        st.n_agents, st.training = self.nAgents, int(bool(self.training))
        st.r, st.alph_Q, st.alph_W = float(self.r), float(self.alph_Q), float(self.alph_W)
        st.mass, st.grav = float(getattr(self, "mass", 1.0)), float(getattr(self, "grav", 9.81))
        st.xtarget = xt.data_ptr()
        return st, [xt]

    def _c_struct64(self, device):
        """the same attributes for the double-precision entry point (xtarget as float64)"""
        if self.obstacle not in _lib.OBS_CODES:
            raise ValueError(f"obstacle {self.obstacle!r} is not one the reference implements")
        xt = self.xtarget.detach().to(device=device, dtype=torch.float64).contiguous()
        st = _lib.NocfProb64()
        st.kind, st.obstacle = self.KIND, _lib.OBS_CODES[self.obstacle]
        st.n_agents, st.training = self.nAgents, int(bool(self.training))
        st.r, st.alph_Q, st.alph_W = float(self.r), float(self.alph_Q), float(self.alph_W)
        st.mass, st.grav = float(getattr(self, "mass", 1.0)), float(getattr(self, "grav", 9.81))
        st.xtarget = xt.data_ptr()
        return st, [xt]

    def _eval64(self, x, p, want):
        """the same calls in double precision (nocf_prob_eval_f64)"""
        x = _lib.require_device_f64(x, "x")
        p = _lib.require_device_f64(p, "p")
        n = x.shape[0]
        st, keep = self._c_struct64(x.device)
        st32, _k = self._c_struct(x.device)
        cdim = _lib.lib().nocf_ctrl_dim(C.byref(st32), self.d)
        kw = dict(device=x.device, dtype=torch.float64)
        lhqw = torch.empty(n, 4, **kw) if "lhqw" in want else None
        gp = torch.empty(n, self.d, **kw) if "gradpH" in want else None
        ct = torch.empty(n, cdim, **kw) if "ctrls" in want else None
        with torch.cuda.device(x.device):
            rc = _lib.lib().nocf_prob_eval_f64(C.byref(st), self.d, _lib.ptr(x), _lib.ptr(p), n,
                                               _lib.ptr(lhqw), _lib.ptr(gp), _lib.ptr(ct), _lib.stream_ptr(x.device))
        _lib.check(rc, "nocf_prob_eval_f64")
        return lhqw, gp, ct

    def _eval(self, x, p, want):
        if isinstance(x, torch.Tensor) and x.dtype == torch.float64:
            return self._eval64(x, p, want)
        x = _lib.require_device_f32(x, "x")
        p = _lib.require_device_f32(p, "p")
        n = x.shape[0]
        st, keep = self._c_struct(x.device)
        cdim = _lib.lib().nocf_ctrl_dim(C.byref(st), self.d)
        lhqw = torch.empty(n, 4, device=x.device) if "lhqw" in want else None
        gp = torch.empty(n, self.d, device=x.device) if "gradpH" in want else None
        ct = torch.empty(n, cdim, device=x.device) if "ctrls" in want else None
        with torch.cuda.device(x.device):
            rc = _lib.lib().nocf_prob_eval_f32(C.byref(st), self.d, _lib.ptr(x), _lib.ptr(p), n,
                                               _lib.ptr(lhqw), _lib.ptr(gp), _lib.ptr(ct), _lib.stream_ptr(x.device))
        _lib.check(rc, "nocf_prob_eval_f32")
        return lhqw, gp, ct

    def calcLHQW(self, x, p):
        """(L, H, Q, W), each n-by-1."""
        lhqw, _, _ = self._eval(x, p, ("lhqw",))
        return lhqw[:, 0:1], lhqw[:, 1:2], lhqw[:, 2:3], lhqw[:, 3:4]

    def calcGradpH(self, x, p):
        _, gp, _ = self._eval(x, p, ("gradpH",))
        return gp

    def calcCtrls(self, x, p):
        _, _, ct = self._eval(x, p, ("ctrls",))
        return ct

    def calcQ(self, x):
        """sum over agents of the obstacle cost, as calcLHQW reports it"""
        lhqw, _, _ = self._eval(x, torch.zeros_like(x), ("lhqw",))
        return lhqw[:, 2:3]

    def calcW(self, x, p=None):
        lhqw, _, _ = self._eval(x, torch.zeros_like(x), ("lhqw",))
        return lhqw[:, 3:4]
