"""Config 5's Double-DQN update eager vs replayed from a captured graph (kernel boundaries are cheaper inside a graph).
python tools/dqn_graph_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_drl_amd import dqn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200


class O5L:
    obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed = 84 * 84 * 4, 4, [400, 300], 0.99, 1e-3, 0.995, 512, 2


def batch():
    g = torch.Generator(device="cuda").manual_seed(1)
    return {"obs1": torch.rand(512, O5L.obs_dim, device="cuda", generator=g), "obs2": torch.rand(512, O5L.obs_dim, device="cuda", generator=g),
            "acts": torch.randint(0, 4, (512,), device="cuda", generator=g).float(), "rews": torch.randn(512, device="cuda", generator=g),
            "done": (torch.rand(512, device="cuda", generator=g) < 0.01).float()}


b5 = batch()
le = dqn.Learner(O5L, "learner")
for _ in range(3):
    le.train(b5, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    le.train(b5, 0)
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / n
print("eager : %.1f us per update" % (te * 1e6))

lg = dqn.Learner(O5L, "learner")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        lg.train(b5, 0)
    torch.cuda.synchronize()
    for per in (1, 10):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(per):
                lg.train(b5, 0)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n // per):
            g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / (n // per * per)
        print("graph of %2d: %.1f us per update (%.3f of eager)" % (per, tg * 1e6, tg / te))
