from neuraloc_amd.problem.Cross2D import Cross2D  # noqa: F401
