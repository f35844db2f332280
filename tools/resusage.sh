#!/bin/bash
# usage: tools/resusage.sh file.hip  -> kernel, VGPR, AGPR, scratch, occupancy, spills
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -I/root/repo/uno_amd/csrc -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|VGPRs Spill" | sed -e 's/.*remark: [^ ]* //' -e 's/\[-R.*//' | paste - - - - - - | sed -e 's/Function Name: //' -e 's/_ZN3uno[0-9]*//' -e 's/NS_[0-9A-Za-z]*E//' -e 's/\[bytes\/lane\]//' -e 's/\[waves\/SIMD\]//' | tr -s ' \t' ' '
