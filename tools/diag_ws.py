"""diagnostic: which workspace buffers differ between repeated evaluations of the same model (same parameters)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gpitch_amd.synth import make_problem, pdgp_from_problem

def ranges(mask):
    idx = np.flatnonzero(mask)
    if idx.size == 0: return []
    out = []; s = idx[0]; p = idx[0]
    for i in idx[1:]:
        if i > p + 512: out.append((s, p)); s = i
        p = i
    out.append((s, p)); return out

def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    prob = make_problem(4096, 128, 2, num_partials=3, seed=7)
    for d in prob["kern_act"]:
        d["type"] = "matern32"; d["lengthscales"] = 1.0
    model = pdgp_from_problem(prob)
    model.za.fixed = True; model.zc.fixed = True
    model._pack()
    hh = model._handle
    hh.check(hh.lib.gp_pdgp_set_overlap(model._plan, level))
    ws = model._ws
    print("workspace bytes", ws.numel() * ws.element_size(), ws.dtype)
    snaps = []
    for rep in range(8):
        f = model._elbo(True)
        torch.cuda.synchronize()
        v = ws.view(torch.int64).cpu().numpy().copy()
        snaps.append((f, v))
    base = snaps[0][1]
    for rep in range(1, 8):
        f, v = snaps[rep]
        r = ranges(v != base)
        print("rep %d f=%.15e  differing ranges (byte offsets, merged over 4 KB gaps): %d" % (rep, f, len(r)))
        for (a, b) in r[:24]:
            x = v.view(np.float64)[a:b + 1]; y = base.view(np.float64)[a:b + 1]
            rel = np.abs(x - y).max() / max(np.abs(y).max(), 1e-300)
            print("    [%10d, %10d)  %8d bytes  max rel diff %.2e  count %d" % (a * 8, (b + 1) * 8, (b + 1 - a) * 8, rel, int((x != y).sum())))
            if rep == 1 or (b + 1 - a) < 16:
                for j in np.flatnonzero(x != y)[:4]:
                    print("        +%d: %s vs %s   (%.17g vs %.17g)" % (j * 8, hex(int(v[a + j]) & (2**64 - 1)), hex(int(base[a + j]) & (2**64 - 1)), x[j], y[j]))

main()
