#!/bin/bash
# resource usage of every kernel of one .hip source: name, VGPRs, scratch bytes, spilled VGPRs.  usage: resusage.sh file.hip [-Dflags...]
src=$1; shift
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -shared "$@" -Rpass-analysis=kernel-resource-usage "$src" -o /tmp/resusage.so 2>&1 |
  awk '/Function Name:/ {name=$5} / VGPRs: / {v=$4} /ScratchSize/ {sc=$5} /VGPRs Spill:/ {print name, "vgprs", v, "scratch", sc, "spill", $5}' | sort -u | c++filt
