#!/bin/bash
timeout -k 10 500 python tests/manual_config_table.py > gpurun_out/r03_configs.log 2>&1; echo "configs rc=$?"; grep "^C[123]" gpurun_out/r03_configs.log | cut -c1-300
cp gpurun_out/configs.json gpurun_out/r03_configs.json
timeout -k 10 1000 python tests/manual_c5_check.py --steps 2 --timing-steps 20 > gpurun_out/r03_c5_check.log 2>&1; echo "c5 rc=$?"; tail -8 gpurun_out/r03_c5_check.log
cp gpurun_out/c5_check.json gpurun_out/r03_c5_check.json
