#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for d in 0 1 2 3; do
  ( SD_CONV_DEBUG=$d timeout 200 python tools/probe_hand_conv.py --reps 4 --size3d 0 2>&1 | grep "^2D" | cut -c1-80 ) > gpurun_out/c4_probe_dbg$d.log 2>&1
done
paste -d'|' <(cut -c1-46,60-80 gpurun_out/c4_probe_dbg0.log) <(cut -c60-80 gpurun_out/c4_probe_dbg1.log) <(cut -c60-80 gpurun_out/c4_probe_dbg2.log) <(cut -c60-80 gpurun_out/c4_probe_dbg3.log)
