set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b1; mkdir -p $O
python -m pytest tests/test_gpu_rccl.py -x -q > $O/rccl.log 2>&1; tail -3 $O/rccl.log
python scratch/capture_control.py > $O/capture_control.log 2>&1; grep "restore OFF" $O/capture_control.log
echo "== cfg5 A/B (r5 library vs this tree)"
for i in 1 2; do
DDRL_LIB_PATH=$PWD/tools/ab/libddrl_hip_r5.so python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update" | sed 's/^/r5  /'
python tools/ddqn_cfg5_prof.py 60 2>&1 | grep "ddqn update" | sed 's/^/new /'
done
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd())
from distributed_drl_amd import dqn
class O5L:
    obs_dim, act_dim, hidden_size, gamma, lr, polyak, batch_size, seed = 84 * 84 * 4, 4, [400, 300], 0.99, 1e-3, 0.995, 512, 2
l5 = dqn.Learner(O5L, "learner")
b5 = {"obs1": torch.rand(512, O5L.obs_dim, device="cuda"), "obs2": torch.rand(512, O5L.obs_dim, device="cuda"),
      "acts": torch.randint(0, 4, (512,), device="cuda").float(), "rews": torch.randn(512, device="cuda"), "done": (torch.rand(512, device="cuda") < 0.01).float()}
for _ in range(3): l5.train(b5, 0)
for r in range(3):
    print("stage ms (new):", ["%.1f" % (x * 1e3) for x in l5.stage_times(b5, 20)])
PY
python tools/sk_ab.py 2>&1 | grep -v amdgpu.ids | tail -6
python -m pytest tests/test_gpu_math_fixtures.py tests/test_gpu_sac1.py -x -q -k "config5 or ddqn or sqn or wide" > $O/dqn_tests.log 2>&1; tail -3 $O/dqn_tests.log
