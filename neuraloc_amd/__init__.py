"""neuraloc_amd -- MI355X-native OCflow rollout (the NeuralOC hot path) behind the reference's
own Python interface.  Importing the package does not load the HIP library; the first call does,
and raises if it is missing."""
from .Phi import Phi, ResNN, antiderivTanh, derivTanh
from .OCflow import OCflow, ocG
from .problem import Cross2D, SwarmTraj, Quadcopter
from .initProb import initProb, resample
from .distributed import OCflow_sharded, shard_rows, reduce_cost_sums
from ._lib import check_errors

__all__ = ["Phi", "ResNN", "antiderivTanh", "derivTanh", "OCflow", "ocG", "Cross2D", "SwarmTraj",
           "Quadcopter", "initProb", "resample", "OCflow_sharded", "shard_rows", "reduce_cost_sums", "check_errors"]
