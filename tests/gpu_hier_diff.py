"""Debug aid: run the hierarchical encoder fwd+bwd under conv kernel generation 1 and 2 and diff every tensor."""
import sys, torch
sys.path.insert(0, "tests")
from conftest import load_golden, as_t
from vae_gan_amd import ops, lib as L
from vae_gan_amd.testing import fill_state_dict
from vae_gan_amd.hierarchical_vae_gan.model import HierarchicalEncoder

ops.set_compute_dtype(torch.float32)
g = load_golden("hier_encoder")
print("x", g["x"].shape)
res = {}
for mode in (1, 2):
    L.load().vg_set_conv_kernel(mode)
    enc = HierarchicalEncoder(3, 64, [32, 64, 128], 3)
    enc.load_state_dict(fill_state_dict(enc.state_dict()))
    enc.cuda().train()
    x = as_t(g["x"]).cuda().requires_grad_(True)
    mu, lv = enc(x)
    ys = list(mu) + list(lv)
    # same cotangent layout as the test: outputs flattened in module order
    y = enc(x)
    flat = []
    for t in y:
        flat.extend(t)
    sum((t * as_t(g[f"cot{i}"]).cuda()).sum() for i, t in enumerate(flat)).backward()
    torch.cuda.synchronize()
    out = {f"y{i}": t.detach().double() for i, t in enumerate(flat)}
    out["dx"] = x.grad.double()
    for k, p in enc.named_parameters():
        out["g:" + k] = p.grad.double()
    res[mode] = out
worst = []
for k in res[1]:
    a, b = res[1][k], res[2][k]
    r = float((a - b).norm() / b.norm().clamp_min(1e-30))
    worst.append((r, k))
worst.sort(reverse=True)
for r, k in worst:
    if "bias" in k and ("conv" in k or k.endswith(".0.bias")): continue
    print(f"{r:.3e} {k}  noise={float(g['noise:' + k[2:]]) if ('noise:' + k[2:]) in g else float(g.get('noise:' + k, -1)):.2e}")
