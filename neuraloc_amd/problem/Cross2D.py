"""2-D agents: corridors, swaps, crossings (interface of src/problem/Cross2D.py:10-66)."""
from .. import _lib
from .base import _ProblemBase


class Cross2D(_ProblemBase):
    KIND = _lib.PROB_CROSS2D
    AGENT_DIM = 2

    def __init__(self, xtarget, obstacle=None, alph_Q=1.0, alph_W=1.0, r=0.5):
        self._init_common(xtarget, obstacle, alph_Q, alph_W, r)

    def __repr__(self):
        return "Cross2D Object"

    def __str__(self):
        return "Cross2D Object \n d = {:} \n nAgents = {:} \n xtarget = {:} \n obstacle:{:}".format(
            self.d, self.nAgents, self.xtarget, self.obstacle)
