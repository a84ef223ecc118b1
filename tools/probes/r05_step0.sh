# round-5 step 0: where does the K > 64 route spend its time on the mixture input?
#   per-iteration kernel table, exact-queue statistics (HSGK_Q_STATS build in ab_libs/libqstats.so), and the
#   bench lines of cfg3 / cfg4 / cfg5 with their extra runs (mixture, labelled + ignore band).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_step0; mkdir -p $O
cd $R
for fl in mixture iid; do
  bash tools/probes/cfg_iter_trace.sh cfg4 $fl > $O/cfg4_${fl}_iter_trace.txt 2>&1
done
bash tools/probes/cfg_iter_trace.sh cfg5 mixture > $O/cfg5_mixture_iter_trace.txt 2>&1
cp hsg_amd/csrc/libhsgk.so /tmp/libkeep.so
cp ab_libs/libqstats.so hsg_amd/csrc/libhsgk.so
for fl in mixture iid; do timeout 300 python tools/probes/qstats.py cfg4 $fl > $O/cfg4_${fl}_qstats.txt 2>&1; done
timeout 300 python tools/probes/qstats.py cfg5 mixture > $O/cfg5_mixture_qstats.txt 2>&1
cp /tmp/libkeep.so hsg_amd/csrc/libhsgk.so
for w in cfg3 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --cpu-images 0 --no-exchange 2>$O/bench_$w.err | tail -1 > $O/bench_$w.json
done
tail -n 40 $O/cfg4_mixture_iter_trace.txt $O/cfg4_mixture_qstats.txt
