#!/bin/bash
# long sentences of BASELINE config 5 on their own: phase cycles when the chip is full / one wave per CU / one wave
cd "$(dirname "$0")/.." ; mkdir -p gpurun_out
C="--user 1000 --ignore-space --mgl 24 --space-p 0.1 --law mixed --min-chars 500 --steps 3"
for k in 0 256 1; do
  echo "=== keep=$k" ; timeout 300 python tools/phase_profile.py $C --keep $k 2>&1 | grep -v amdgpu.ids
done > gpurun_out/long.txt 2>&1
echo "=== keep=256 min 1500" >> gpurun_out/long.txt
timeout 300 python tools/phase_profile.py --user 1000 --ignore-space --mgl 24 --space-p 0.1 --law mixed --min-chars 1500 --steps 3 --keep 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/long.txt
cat gpurun_out/long.txt
