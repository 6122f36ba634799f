"""Determinism soak of the config-5 learner (wide layer 1: split-K forward + reduce, LDS-DMA staging): two learners from the same
seed run the same n updates over the same rotating batches; parameters, targets and both Adam moments must end bit-identical
and finite.  python tools/dqn_soak.py [n] [variant: ddqn | sqn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distributed_drl_amd import _lib, dqn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sqn = len(sys.argv) > 2 and sys.argv[2] == "sqn"


class O:
    obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed, alpha = 84 * 84 * 4, 4, [400, 300], 0.99, 1e-4, 0.995, 512, 2, 0.1


g = torch.Generator(device="cuda").manual_seed(1)
batches = [{"obs1": torch.rand(512, O.obs_dim, device="cuda", generator=g), "obs2": torch.rand(512, O.obs_dim, device="cuda", generator=g),
            "acts": torch.randint(0, 4, (512,), device="cuda", generator=g).float(), "rews": torch.randn(512, device="cuda", generator=g),
            "done": (torch.rand(512, device="cuda", generator=g) < 0.01).float()} for _ in range(3)]
res = []
for run in range(2):
    l = (dqn.LearnerSQN if sqn else dqn.Learner)(O, "learner")
    for it in range(n):
        l.train(batches[it % 3], it)
    torch.cuda.synchronize()
    res.append([l.export(w).cpu().numpy() for w in (_lib.SAC1_MAIN, _lib.SAC1_TARGET, _lib.SAC1_ADAM_M, _lib.SAC1_ADAM_V)])
    del l
ok = all(np.array_equal(a, b) for a, b in zip(*res)) and all(np.isfinite(a).all() for a in res[0])
print("%s, %d updates x 2 runs at obs 28 224 / batch 512: %s (|main| max %.3g)" % ("sqn" if sqn else "ddqn", n, "bit-identical and finite" if ok else "DIFFER", np.abs(res[0][0]).max()))
sys.exit(0 if ok else 1)
