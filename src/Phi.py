from neuraloc_amd.Phi import Phi, ResNN, antiderivTanh, derivTanh  # noqa: F401

__all__ = ["Phi", "ResNN", "antiderivTanh", "derivTanh"]
