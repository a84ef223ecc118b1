R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "torch_extension or exchange or segment_reduce or prototype or whole_train or train_step or hierarchy" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for r in 1 2; do
for b in torch ctypes; do HSGK_BINDING=$b timeout 300 python tools/probes/train_step_wall.py $b 2>&1 | tail -1; done
done
