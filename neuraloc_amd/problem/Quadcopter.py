"""12-state rigid-body quadcopters (interface of src/problem/Quadcopter.py:7-63)."""
from .. import _lib
from .base import _ProblemBase


class Quadcopter(_ProblemBase):
    KIND = _lib.PROB_QUADCOPTER
    AGENT_DIM = 12

    def __init__(self, xtarget, obstacle=None, alph_Q=1.0, alph_W=1.0, mass=1.0, grav=9.81, r=1.0):
        self._init_common(xtarget, obstacle, alph_Q, alph_W, r)
        self.mass = mass
        self.grav = grav
        if obstacle is not None:
            # the reference prints "not implemented" on every call and uses 0 (Quadcopter.py:116-122)
            raise ValueError("Quadcopter implements only obstacle=None")

    def __repr__(self):
        return "Quadcopter Object"

    def __str__(self):
        return ("Quadcopter Object Optimal Control \n d = {:} \n nAgents = {:} \n xtarget = {:} \n obstacle = {:}  "
                "\n alph_Q = {:}  \n mass = {:} \n grav = {:}").format(
            self.d, self.nAgents, self.xtarget, self.obstacle, self.alph_Q, self.mass, self.grav)
