"""Per-tensor diagnosis of one Double-DQN shape of the fuzz (tests/test_gpu_fuzz_shapes.py): HIP gradient vs the float64 and float32
oracles, tensor by tensor, with the rows / columns of the layer-1 gradient that are off — how the relu-sign-at-zero cases of round 6 were read
(DESIGN section 2).  usage: python tests/_fuzz_diag_dqn.py obs acts hidden1 hidden2 batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributed_drl_amd import _lib, dqn
from oracle import dqn_oracle as do
obs, acts, hid, batch = (int(sys.argv[1]), int(sys.argv[2]), (int(sys.argv[3]), int(sys.argv[4])), int(sys.argv[5]))
class Opt:
    obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed, alpha = obs, acts, list(hid), 0.99, 1e-3, 0.995, batch, 2, 0.1
learner = dqn.Learner(Opt, "learner")
cfg = do.Config(obs_dim=obs, n_actions=acts, hidden1=hid[0], hidden2=hid[1], batch=batch)
params = do.init_params(cfg, 2)
rs = np.random.RandomState(3)
for k in params:
    if k.endswith("bias"):
        params[k] = rs.uniform(-0.1, 0.1, params[k].shape).astype(np.float32)
learner.set_weights(list(params.keys()), list(params.values()))
o64, o32 = (do.DqnOracle(cfg, params, dt) for dt in (torch.float64, torch.float32))
b = do.synthetic_batch(cfg, 10)
w = o64.step(b); o32.step(b)
loss, q = learner.train(b, 0, return_outputs=True)
g, g64, g32 = learner.export(_lib.SAC1_GRAD).cpu().numpy(), o64.flat("grads"), o32.flat("grads")
print("shape", obs, acts, hid, batch, "env", {k: v for k, v in os.environ.items() if k.startswith("DDRL_")})
print("loss rel", abs(loss.item() - float(w["q_loss"])) / abs(float(w["q_loss"])), " q max abs diff", np.abs(q.cpu().numpy() - w["q"].numpy()).max())
gmax = np.abs(g64).max()
off = 0
for name, shape in learner.specs:
    n = int(np.prod(shape)); sl = slice(off, off + n); off += n
    e, e32 = np.abs(g[sl] - g64[sl]), np.abs(g32[sl] - g64[sl])
    rel_bad = (e > 1e-3 * np.abs(g64[sl])).mean(); rel_bad32 = (e32 > 1e-3 * np.abs(g64[sl])).mean()
    print("%-24s n %8d  max|g| %.3g  hip: max err %.3g (%.2g of gmax)  frac > 2e-4 gmax %.4f  frac rel > 1e-3 %.4f | torch f32: max err %.3g, frac rel %.4f" %
          (name, n, np.abs(g64[sl]).max(), e.max(), e.max() / gmax, (e > 2e-4 * gmax).mean(), rel_bad, e32.max(), rel_bad32))
    if "dense/kernel" in name and e.max() > 1e-4 * gmax:
        E = e.reshape(shape)
        rows = np.where(E.max(axis=1) > 1e-4 * gmax)[0]; cols = np.where(E.max(axis=0) > 1e-4 * gmax)[0]
        print("   rows with large errors: %d of %d (first %s ... last %s); cols: %d of %d (%s)" % (len(rows), shape[0], rows[:6], rows[-3:], len(cols), shape[1], cols[:12]))
