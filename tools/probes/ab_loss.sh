# same-box A/B of libhsgk builds under ab_libs/ on the loss forward+backward
for round in 1 2; do for v in $(ls $GRAFT_REPO_ROOT/ab_libs | sed 's/lib\(.*\).so/\1/'); do cp $GRAFT_REPO_ROOT/ab_libs/lib$v.so $GRAFT_REPO_ROOT/hsg_amd/csrc/libhsgk.so; echo -n "$v  "; timeout 300 python $GRAFT_REPO_ROOT/tools/probes/loss_time.py ${LOSS_N:-200704} 256 ${LOSS_P:-3072} 2>&1 | tail -1; done; done
