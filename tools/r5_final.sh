# round-5 final GPU batch on the committed tree: smoke, the whole GPU suite, the default bench line, three more fuzz seeds, the long soak
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"; O=gpurun_out/r5z; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
python -m pytest tests -q -m gpu --durations=8 > $O/t_all.log 2>&1; echo rc=$? >> $O/t_all.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
{
for seed in 10 11 12; do
  echo "== DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu"
  DDRL_FUZZ_N=150 DDRL_FUZZ_SEED=$seed timeout 900 python -m pytest tests/test_gpu_fuzz_shapes.py -q -m gpu 2>&1 | tail -n 1
done
echo "== soak 600000 x 256 (50 per graph)"; timeout 1200 python tools/soak.py 600000 256 50 2>&1 | tail -n 1
} > $O/record.txt 2>&1
tail -n 2 $O/smoke.log; tail -n 14 $O/t_all.log; head -c 400 $O/bench.json; echo; tail -n 2 $O/bench.err; cat $O/record.txt
