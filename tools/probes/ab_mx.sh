# elimination runs of the matrix-core sums update (sums_fx.hip, HSGK_MX_DEBUG) on one box: only the FIRST update launch of a
# step is comparable (later launches see labels that depend on the -- then wrong -- sums)
export HSGK_MSTEP=mfma
bash tools/probes/ab_kernel.sh sums_fx.hip update_sums_mfma "-DHSGK_MX_DEBUG=0" "-DHSGK_MX_DEBUG=1" "-DHSGK_MX_DEBUG=2" "-DHSGK_MX_DEBUG=3" "-DHSGK_MX_DEBUG=4" "-DHSGK_MX_DEBUG=0"
