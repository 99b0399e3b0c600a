#!/bin/bash
# tools/kres.sh file.hip : per-kernel registers / scratch / LDS / occupancy as the compiler reports them
cd "$(dirname "$0")/../kwiiyatta_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  grep -E "Function Name|  VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" |
  sed -E 's/^.*remark: +//; s/ \[-Rpass.*$//' |
  awk '/Function Name/{if (l) print l; l=$3; next} {l=l " | " $0} END{print l}'
