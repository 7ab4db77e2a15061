from neuraloc_amd.problem.Quadcopter import Quadcopter  # noqa: F401
