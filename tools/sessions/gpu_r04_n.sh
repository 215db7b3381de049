#!/bin/bash
# round 4, session n: XCD-aware tile order on / off, then soaks of the final build (pixel path vs oracle; entropy stage vs walker)
O=gpurun_out/r04n; mkdir -p $O
timeout 600 bash tools/ab_libs.sh libzjhip.so libzjhip_noxcd.so libzjhip.so libzjhip_noxcd.so 2>&1 | tee $O/ab_xcd.txt
timeout 500 python tools/pixel_soak.py --seconds 400 --seed 404 > $O/pixel_soak.txt 2>&1; tail -12 $O/pixel_soak.txt
timeout 400 python tools/entropy_soak.py --seconds 300 --seed 44 > $O/entropy_soak.txt 2>&1; tail -8 $O/entropy_soak.txt
