for mode in plain noprof; do
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    case $mode in plain) e="";; nogc) e="HSGK_BENCH_NOGC=1";; noprof) e="HSGK_BENCH_NOPROF=1";; esac
    env $e HSGK_BENCH_STEP_TIMES=nosync timeout 200 python bench.py --workload cfg3 --steps 10 --warmup 3 --cpu-images 0 --no-extra --no-exchange 2>&1 | grep -E "step ms|ms_per_step" | sed -e "s/.*\"ms_per_step\": \([0-9.]*\).*/ms_per_step \1/" | tr '\n' ' '; echo " [$mode]"
  done
done
