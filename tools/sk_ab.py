"""A/B on one box: config 5's Double-DQN update with the stream-K layer-1 wgrad (default) and with DDRL_WIDE_SK=0 (tile-per-workgroup
k_wide<false>), same inputs: gradients and parameters must agree bit for bit is NOT expected (different sum order inside split tiles),
so: max |diff| of the layer-1 gradient relative to its RMS, and the eager time per update.  python tools/sk_ab.py"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from distributed_drl_amd import _lib, dqn

    class O:
        obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed = 84 * 84 * 4, 4, [400, 300], 0.99, 1e-3, 0.995, 512, 2
    l = dqn.Learner(O, "learner")
    g = torch.Generator(device="cuda").manual_seed(0)
    b = {"obs1": torch.randint(0, 256, (512, O.obs_dim), device="cuda", generator=g).float(), "obs2": torch.randint(0, 256, (512, O.obs_dim), device="cuda", generator=g).float(),
         "acts": torch.randint(0, 4, (512,), device="cuda", generator=g).float(), "rews": torch.randn(512, device="cuda", generator=g), "done": torch.zeros(512, device="cuda")}
    n, v = l.get_weights(); l.set_weights(n[:1], [v[0] / 64])
    l.train(b, 0)
    grad = l.export(_lib.SAC1_GRAD).cpu().numpy()
    np.save(sys.argv[1], grad)
    for _ in range(5):
        l.train(b, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        l.train(b, 0)
    torch.cuda.synchronize()
    print("%s: %.1f us per update; wgrad stage %.1f us" % (sys.argv[1], (time.perf_counter() - t0) / 50 * 1e6, l.stage_times(b, 20)[7] * 1e3 - 5.0), flush=True)
else:
    import numpy as np
    for tag, env in (("sk", {}), ("tile", {"DDRL_WIDE_SK": "0"}), ("sk2", {})):
        subprocess.check_call([sys.executable, __file__, "/tmp/g_%s.npy" % tag], env=dict(os.environ, **env))
    a, b, c = (np.load("/tmp/g_%s.npy" % t) for t in ("sk", "tile", "sk2"))
    n1 = 28224 * 400 + 400
    print("stream-K run to run bit-identical:", bool(np.array_equal(a, c)))
    print("layer-1 gradient sk vs tile: max |diff| %.3g, rms %.3g" % (np.abs(a[:n1] - b[:n1]).max(), np.sqrt((b[:n1].astype(np.float64) ** 2).mean())))
    print("other layers identical:", bool(np.array_equal(a[n1:], b[n1:])))
