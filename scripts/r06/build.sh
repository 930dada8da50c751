#!/bin/bash
# build the product library (and with "lab" the knob build) from anywhere; prints only errors / warnings
cd /root/repo/spherical_sfm_amd/csrc || exit 1
make -j4 2>&1 | grep -i "error\|warning" -A3 | head -30
[ "$1" = "lab" ] && make lab -j4 2>&1 | grep -i "error\|warning" -A3 | head -30
ls -la --time-style=+%H:%M:%S ../libssfm_hip.so ../libssfm_hip_lab.so | awk '{print $6, $7}'
