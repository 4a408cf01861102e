export TMPDIR=/tmp
for s in 0 1 15; do
  PDEKAN_ABLATE=$s timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/abl_$s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/abl_$s.log 2>&1 || exit 1
done
echo done
