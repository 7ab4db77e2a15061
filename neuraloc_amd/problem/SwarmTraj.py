"""3-D point agents flying around two blocks (interface of src/problem/SwarmTraj.py:12-66)."""
from .. import _lib
from .base import _ProblemBase


class SwarmTraj(_ProblemBase):
    KIND = _lib.PROB_SWARMTRAJ
    AGENT_DIM = 3

    def __init__(self, xtarget, obstacle=None, alph_Q=1.0, alph_W=1.0, r=0.5):
        self._init_common(xtarget, obstacle, alph_Q, alph_W, r)

    def __repr__(self):
        return "SwarmTraj Object"

    def __str__(self):
        return "SwarmTraj Object \n d = {:} \n nAgents = {:} \n xtarget = {:} \n obstacle:{:}".format(
            self.d, self.nAgents, self.xtarget, self.obstacle)
