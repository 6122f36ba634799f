"""Per-tensor gradient error of one SAC1 shape against the float64 oracle (helper for tests/test_gpu_fuzz_shapes.py failures).
python tests/_fuzz_diag.py obs act h1 h2 batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import sac1_oracle as so
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters, Learner

obs, act, h1, h2, batch = (int(x) for x in sys.argv[1:6])
opt = HyperParameters()
opt.obs_dim, opt.act_dim, opt.hidden_sizes, opt.batch_size, opt.seed = obs, act, (h1, h2), batch, 7
learner = Learner(opt)
print("fused", learner._lib.ddrl_sac1_is_fused(learner._h))
cfg = so.Config(obs_dim=obs, act_dim=act, hidden1=h1, hidden2=h2, batch=batch, alpha=opt.alpha, gamma=opt.gamma, lr=opt.lr, polyak=opt.polyak)
params = so.init_params(cfg, 7)
rs = np.random.RandomState(11)
for k in params:
    if k.endswith("bias"):
        params[k] = rs.uniform(-0.05, 0.05, params[k].shape).astype(np.float32)
learner.set_weights(list(params.keys()), list(params.values()))
o64 = so.Sac1Oracle(cfg, params, torch.float64)
o32 = so.Sac1Oracle(cfg, params, torch.float32)
b, eps = so.synthetic_batch(cfg, seed=90)
w = o64.step(b, *eps)
o32.step(b, *eps)
losses, (q1, q2, lp) = learner.train(b, eps=eps, return_outputs=True)
print("losses", [l.item() for l in losses], [float(w[k]) for k in ("pi_loss", "q1_loss", "q2_loss")])
g = learner.export(_lib.SAC1_GRAD).cpu().numpy()
g64, g32 = o64.flat("grads"), o32.flat("grads")
gmax = np.abs(g64).max()
off = 0
for n in o64.names:
    sz = o64.grads[n].numel()
    e, e32 = np.abs(g[off:off + sz] - g64[off:off + sz]).max(), np.abs(g32[off:off + sz] - g64[off:off + sz]).max()
    print("%-28s n %7d  |g|max %.3e  err(hip) %.3e (%.2e of global max)   err(torch f32) %.3e" % (n, sz, np.abs(g64[off:off + sz]).max(), e, e / gmax, e32))
    off += sz
