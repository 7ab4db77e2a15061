"""Which singlequad rows differ between the mono kernel, the tile kernel and the oracle (diagnostic; GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import neuraloc_amd as na
from oracle import ocflow_oracle as orc
from conftest import load_golden
from util_hip import closed_form_normal, make_net, make_oracle, make_prob

DEV = torch.device("cuda:0")
g = load_golden("singlequad")
net, prob = make_net(g, DEV), make_prob(g, DEV, training=False)
m = g.meta
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
xi = closed_form_normal(n, m["d"], 5); xi[:, 3:] = 0.0
x = (g.t("xInit") + m["var0"] * xi).contiguous()
def table(xx):
    with torch.no_grad():
        _, c = na.OCflow(xx.to(DEV), net, prob, [0.0, 1.0], 12, "rk4", m["alph"], noMean=True)
    return torch.cat(c, 1).cpu().double()
os.environ["NOCF_MONO"] = "1"; mono = table(x)
os.environ["NOCF_MONO"] = "0"; tile = table(x)
P, S = make_oracle(g, False)
want = torch.as_tensor(orc.persample_table(x, P, S, [0.0, 1.0], 12, "rk4", m["alph"])).double()
pert = torch.as_tensor(orc.persample_table(x * (1 + 1e-6), P, S, [0.0, 1.0], 12, "rk4", m["alph"])).double()
def off(a, b): return ((a - b).abs() > 1e-3 + 1e-3 * b.abs())
print("rows off: mono-tile", int(off(mono, tile).any(1).sum()), " mono-oracle", int(off(mono, want).any(1).sum()),
      " tile-oracle", int(off(tile, want).any(1).sum()), " oracle(x(1+1e-6))-oracle", int(off(pert, want).any(1).sum()))
rows = off(mono, tile).any(1).nonzero().flatten().tolist()
for r in rows[:12]:
    cols = off(mono, tile)[r].nonzero().flatten().tolist()
    print(r, cols, " ".join("%d: mono %.6g tile %.6g oracle %.6g pert %.6g |" % (c, mono[r, c], tile[r, c], want[r, c], pert[r, c]) for c in cols))
