import os, sys
os.environ["PN2_TEST_VERBOSE"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pranet-v2_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_parity as T
import pn2
pn2.load_library()
print("==== bf16 blocks")
T.test_blocks_vs_oracle("bf16")
print("==== fp32 model probes")
from pn2.loss import structure_loss
from oracle import weights as W
for tag in ("96", "352"):
    z = np.load(os.path.join(T.G, f"pranet_v2_{tag}.npz"))
    size, n = int(z["size"]), int(z["n"])
    model = T._fixture_model()
    x, mask = W.synthetic_batch(n, size, seed=1234)
    x, mask = x.to("cuda"), mask.to("cuda")
    outs = model(x)
    losses = [structure_loss(outs[i], outs[i + 4], mask, 1 - mask) for i in range(4)]
    (losses[3] + losses[2] + losses[1] + losses[0]).backward()
    named = dict(model.named_parameters())
    for f in z.files:
        if f.startswith("graw."):
            k = f[5:]
            r32 = torch.from_numpy(z[f]).double(); r64 = torch.from_numpy(z["f64." + f]).double()
            got = named[k].grad.reshape(-1)[:256].cpu().double()
            own = float((r32 - r64).norm() / (r64.norm() + 1e-30)); e = float((got - r64).norm() / (r64.norm() + 1e-30))
            print(f"  {tag} {k:44s} ours {e:.3e} own {own:.3e} ratio {e / max(own, 2.5e-6):.2f}")
