# elimination runs of the first-level E-step kernel on one box: bash tools/probes/ab_tkernel.sh [t]
#   HSGK_T_DEBUG   (tile-order engine) 1 no MFMA, 2 no epilogue, 3 no table reads, 4 = 1 + 2
#   HSGK_EPI_DEBUG (HalfEpi)           1 no label store, 2 no queue, 3 neither
if [ "$1" = "t" ]; then
export HSGK_TLAYOUT=1
bash tools/probes/ab_kernel.sh kmeans.hip assign_half_t "-DHSGK_T_DEBUG=0" "-DHSGK_T_DEBUG=1" "-DHSGK_T_DEBUG=2" "-DHSGK_T_DEBUG=3" "-DHSGK_T_DEBUG=4" "-DHSGK_T_DEBUG=0"
else
bash tools/probes/ab_kernel.sh kmeans.hip assign_half_kernel "-DHSGK_EPI_DEBUG=0" "-DHSGK_EPI_DEBUG=1" "-DHSGK_EPI_DEBUG=2" "-DHSGK_EPI_DEBUG=3" "-DHSGK_EPI_DEBUG=0"
fi
