# ablation of the M-step accumulate kernel: rebuild the library ON the GPU box with -DHSGK_ABL=n
for a in 0 1 2 3; do
  touch hsg_amd/csrc/*.hip; make -C hsg_amd/csrc EXTRA=-DHSGK_ABL=$a -j8 > /dev/null 2>&1
  echo "ABL=$a"; python tools/bench_kernels.py --reps 10 --only m 2>&1 | tail -1 | cut -c100-330
done
