"""Config 5's learner iteration: sample_batch(512) + train (two calls) against ddrl_dqn_step_ring (the layer-1 GEMMs read the sampled rows
out of the ring), for a small ring (512 rows: TLB-friendly) and a large one.  python tools/fused_iter_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import distributed_drl_amd as d
from distributed_drl_amd import dqn

for cap in (512, 262144):
    class O:
        obs_dim, buffer_size, batch_size, save_dir, act_dim, hidden_size, gamma, lr, polyak, seed = 84 * 84 * 4, cap, 512, ".", 4, [400, 300], 0.99, 1e-3, 0.995, 2
    rb = d.ReplayBufferDQN(O, 0, seed=0)
    for s0 in range(0, cap, 512):
        x = torch.randint(0, 256, (512, O.obs_dim), device="cuda").float()
        rb.store_batch(x, torch.randint(0, 4, (512,), device="cuda").float(), torch.zeros(512, device="cuda"), x, torch.zeros(512, device="cuda"))
    l = dqn.Learner(O, "learner")
    n, v = l.get_weights(); l.set_weights(n[:1], [v[0] / 64])
    for name, fn in (("two calls", lambda: l.train(rb.sample_batch_device(512), 0)), ("one call ", lambda: l.train_from(rb, 0))) * 2:
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize()
        print("ring of %7d rows, %s: %.1f us per iteration" % (cap, name, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
    del rb, l
    torch.cuda.empty_cache()
