#!/bin/bash
# Registers, spills, LDS and occupancy of every kernel whose name matches $1 (default: merge_tiles), from the compiler's
# resource remarks (device-only compile of osp_api.hip; extra flags after the pattern).
cd "$(dirname "$0")/../outerspace_amd/csrc" || exit 1
pat=${1:-merge_tiles}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c osp_api.hip -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk -v pat="$pat" '/Function Name:/ { on = index($0, pat) > 0; if (on) { n = $0; sub(/.*Function Name: /, "", n); sub(/ \[-Rpass.*/, "", n); printf "%s\n", substr(n, 1, 90) } }
       on && /(VGPRs:|Spill:|TotalSGPRs:|Occupancy|LDS Size|ScratchSize)/ { l = $0; sub(/.*remark: +/, "", l); sub(/ \[-Rpass.*/, "", l); printf "    %s\n", l }'
