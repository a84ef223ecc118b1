R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_hard2; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "all_k_entries or cfg4_end_to_end or f19 or exchange_world2" 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
rm -f $O/ab.txt
for w in cfg4; do for fl in iid mixture; do for hard in 1 0; do
  HSGK_HARD=$hard timeout 300 python bench.py --workload $w --flavour $fl --steps 10 --warmup 3 --cpu-images 0 --no-exchange --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $fl hard=$hard', d['ms_per_step'], d['config']['phase_ms_per_step'])" | tee -a $O/ab.txt
done; done; done
for fl in mixture; do
bash tools/probes/cfg_iter_trace.sh cfg4 $fl > $O/cfg4_${fl}_iter_trace.txt 2>&1
tail -42 $O/cfg4_${fl}_iter_trace.txt | head -14
done
cp hsg_amd/csrc/libhsgk.so /tmp/libkeep.so
cp ab_libs/libqstats.so hsg_amd/csrc/libhsgk.so
timeout 300 python tools/probes/qstats.py cfg4 iid 2>&1 | head -4
cp /tmp/libkeep.so hsg_amd/csrc/libhsgk.so
