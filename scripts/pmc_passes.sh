export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_sq -- $B > gpurun_out/pmc_sq.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_grbm -- $B > gpurun_out/pmc_grbm.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- $B > gpurun_out/pmc_fetch.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- $B > gpurun_out/pmc_write.log 2>&1
echo rc=$?
find gpurun_out/pmc_* -name "*.csv" | head -20
