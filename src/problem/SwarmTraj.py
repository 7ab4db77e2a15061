from neuraloc_amd.problem.SwarmTraj import SwarmTraj  # noqa: F401
