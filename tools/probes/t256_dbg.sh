export HSGK_TLAYOUT=1
for a in "-DHSGK_T256_DEBUG=0" "-DHSGK_T256_DEBUG=1" "-DHSGK_T256_DEBUG=2" "-DHSGK_T256_DEBUG=3"; do
  touch hsg_amd/csrc/kmeans.hip; make -C hsg_amd/csrc EXTRA="$a" -j8 > /dev/null 2>&1
  echo "== $a"; python tools/probes/t256_dbg.py 1 2>&1 | grep -v amdgpu
done
