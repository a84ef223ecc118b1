# round 5: the dense pass for all-K exact-queue entries -- parity first, then A/B on cfg4 / cfg5 both flavours
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_hard; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "all_k_entries or cfg4_end_to_end or k256_tile" 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
rm -f $O/ab.txt
for w in cfg4 cfg5; do for fl in iid mixture; do for hard in 1 0; do
  HSGK_HARD=$hard timeout 300 python bench.py --workload $w --flavour $fl --steps 10 --warmup 3 --cpu-images 0 --no-exchange --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $fl hard=$hard', d['ms_per_step'], d['config']['phase_ms_per_step'])" | tee -a $O/ab.txt
done; done; done
for fl in mixture iid; do
bash tools/probes/cfg_iter_trace.sh cfg4 $fl > $O/cfg4_${fl}_iter_trace.txt 2>&1
tail -42 $O/cfg4_${fl}_iter_trace.txt | head -14
done
