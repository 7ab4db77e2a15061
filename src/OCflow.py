from neuraloc_amd.OCflow import OCflow, ocG  # noqa: F401
from neuraloc_amd.Phi import *  # noqa: F401,F403  (the reference's OCflow module re-exports src.Phi)
