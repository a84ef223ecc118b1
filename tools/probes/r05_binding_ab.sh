R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
HSGK_BINDING=ctypes timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or loss or train or segsort or hierarchy or kmeans_vs or segment_by" 2>&1 | grep -E "passed|failed|Error|error" | tail -3
for r in 1 2 3; do
for b in torch ctypes; do HSGK_BINDING=$b timeout 300 python tools/probes/train_step_wall.py $b 2>&1 | tail -1; done
done
