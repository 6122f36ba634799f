"""Config 5's learner alone (Double-DQN, algos/dqn/actor_learner.py, observation width 84*84*4 = 28 224, batch 512) for rocprofv3:
n eager updates on a device batch.  python tools/ddqn_cfg5_prof.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_drl_amd import dqn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30


class O5L:
    obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed = 84 * 84 * 4, 4, [400, 300], 0.99, 1e-3, 0.995, 512, 2


l5 = dqn.Learner(O5L, "learner")
b5 = {"obs1": torch.rand(512, O5L.obs_dim, device="cuda"), "obs2": torch.rand(512, O5L.obs_dim, device="cuda"),
      "acts": torch.randint(0, 4, (512,), device="cuda").float(), "rews": torch.randn(512, device="cuda"),
      "done": (torch.rand(512, device="cuda") < 0.01).float()}
for _ in range(3):
    l5.train(b5, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    l5.train(b5, 0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 2.0 * 512 * (4 * O5L.obs_dim * 400 + 5 * (400 * 300 + 300 * 4))
print("ddqn update at config 5's width: %.3f ms per update, %.1f TFLOP/s (%.2f of the f32 MFMA peak)" % (dt * 1e3, fl / dt / 1e12, fl / dt / 1e12 / 157.3))
if len(sys.argv) > 2:   # python tools/ddqn_cfg5_prof.py n ring_transitions: the learner iteration (sample + train), sample + train on one stream
    import distributed_drl_amd as d

    class O5:
        obs_dim, buffer_size, batch_size, save_dir = O5L.obs_dim, int(sys.argv[2]), 512, "."
    rb = d.ReplayBufferDQN(O5, 0, seed=0)
    z = torch.zeros(2048, device="cuda")
    for s0 in range(0, O5.buffer_size, 2048):
        m = min(2048, O5.buffer_size - s0)
        x = torch.randint(0, 256, (m, O5.obs_dim), device="cuda").float()
        rb.store_batch(x, z[:m], z[:m], x, z[:m])
    del x

    def seq(k):
        for _ in range(k):
            l5.train(rb.sample_batch_device(512), 0)
    for name, fn in (("sequential sample -> train", seq),):
        fn(3); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
        print("%-50s %.3f ms per iteration" % (name, (time.perf_counter() - t0) / n * 1e3))
