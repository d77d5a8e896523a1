for v in "cal whole" "cal" "enc" "enc whole" "dec" "dec whole"; do
  n=$(echo $v | tr ' ' '_')
  PMC_CMD="python tools/p256_wgrad_probe.py $v" PMC_OUT=r03_p256_traffic_$n timeout 300 bash tools/gpu_pmc_traffic_cmd.sh
done > gpurun_out/r03_p256_traffic.txt 2>&1
cat gpurun_out/r03_p256_traffic.txt
