set -u
cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/r6_b14; mkdir -p $O
for i in 1 2; do
DDRL_DQN_GRAPH=0 python tools/ddqn_cfg5_prof.py 100 2>&1 | grep "ddqn update" | sed 's/^/eager: /'
python tools/ddqn_cfg5_prof.py 100 2>&1 | grep "ddqn update" | sed 's/^/graph: /'
done | tee $O/cfg5_graph.txt
python -m pytest tests/test_gpu_math_fixtures.py tests/test_gpu_sac1.py tests/test_gpu_driver.py -x -q -k "dqn or sqn or config5 or wide or stream_k" > $O/tests.log 2>&1; tail -3 $O/tests.log
DDRL_FUZZ_N=60 DDRL_FUZZ_SEED=31 python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu -k dqn 2>&1 | tail -1
