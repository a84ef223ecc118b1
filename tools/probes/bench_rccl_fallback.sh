# the start-up fallback chain of bench.py on ONE device: two ranks, RCCL start-up failure injected
# (RCCL refuses two ranks on one device anyway) -> retry without the IPC variable -> gloo; the line must still appear
cd $GRAFT_REPO_ROOT
HSGK_BENCH_DEVICE=0 HSGK_BENCH_FAKE_RCCL_FAILURE=1 HSGK_BENCH_STARTUP_S=40 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload cfg3 --steps 5 --warmup 2 --cpu-images 0 --no-extra 2>gpurun_out/fallback.err | tail -1 \
  | python -c "import sys,json; j=json.loads(sys.stdin.read()); print({k: j.get(k) for k in ('value','n_gpus','ms_per_step','dist_backend','rccl_ranks','rccl_startup_retried','rccl_fell_back_to_gloo','rccl_startup_first_error','rccl_startup_second_error','exchange_ms')})"
grep "bench rank" gpurun_out/fallback.err | head
