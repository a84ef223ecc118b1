R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_hard4; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "all_k_entries or cfg4_end_to_end" 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
rm -f $O/ab.txt
run() { env "$@" timeout 300 python bench.py --workload cfg4 --flavour mixture --steps 10 --warmup 3 --cpu-images 0 --no-exchange --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['config']['phase_ms_per_step'])" | tee -a $O/ab.txt; }
for r in 1 2; do
run HSGK_HARD=1
run HSGK_HARD=lds
run HSGK_HARD=1 HSGK_HARD_PREV=0

run HSGK_HARD=1 HSGK_HARD_SKIP=0

done
bash tools/probes/cfg_iter_trace.sh cfg4 mixture > $O/cfg4_mixture_iter_trace.txt 2>&1
tail -42 $O/cfg4_mixture_iter_trace.txt | head -14
HSGK_HARD_PREV=0 bash tools/probes/cfg_iter_trace.sh cfg4 mixture > $O/cfg4_mixture_iter_trace_noprev.txt 2>&1
tail -42 $O/cfg4_mixture_iter_trace_noprev.txt | head -14
