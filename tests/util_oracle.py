"""oracle-side helpers shared by the CPU tests"""
from oracle import ocflow_oracle as orc


def spec_of(g, training):
    m = g.meta
    kind = {"Cross2D": orc.KIND_CROSS2D, "SwarmTraj": orc.KIND_SWARM, "Quadcopter": orc.KIND_QUAD}[m["prob_class"]]
    return orc.ProbSpec(kind=kind, xtarget=g.t("xtarget"), obstacle=m["obstacle"], alph_Q=m["alph_Q"],
                        alph_W=m["alph_W"], r=m["r"], training=training)
