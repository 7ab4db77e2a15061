"""Import shims: `from src.Phi import *`, `from src.OCflow import OCflow`, `from src.initProb import *`,
`from src.problem.Cross2D import *` -- the module paths NeuralOC-style drivers (evalOC.py, trainOC.py,
timeOC.py) use -- resolve to this repository's own MI355X implementation (neuraloc_amd)."""
