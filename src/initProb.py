from neuraloc_amd.initProb import initProb, resample  # noqa: F401
from neuraloc_amd.problem import Cross2D, Quadcopter, SwarmTraj  # noqa: F401
