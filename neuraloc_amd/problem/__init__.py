from .Cross2D import Cross2D
from .SwarmTraj import SwarmTraj
from .Quadcopter import Quadcopter

__all__ = ["Cross2D", "SwarmTraj", "Quadcopter"]
