export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_valu -- $B > gpurun_out/pmc_valu.log 2>&1
echo rc=$?
