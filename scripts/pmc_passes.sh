set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- $B > gpurun_out/pmc_fetch.log 2>&1
echo fetch done
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- $B > gpurun_out/pmc_write.log 2>&1
echo write done
timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- $B > gpurun_out/pmc_mfma.log 2>&1
echo mfma done
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_prof -- $B > gpurun_out/pmc_prof.log 2>&1
echo stats done
python3 profiles/make_pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_prof gpurun_out/r03_pmc_traffic_new.json
