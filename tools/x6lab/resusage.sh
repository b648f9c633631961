#!/bin/bash
# per-kernel registers / spills / occupancy / LDS of one lab source:  tools/x6lab/resusage.sh hlab.hip
hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$@" 2>&1 | awk '
/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
/ VGPRs: / {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
/AGPRs: / {a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
/ SGPRs: / {sg=$0; sub(/.* SGPRs: /,"",sg); sub(/ \[.*/,"",sg)}
/ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ \[.*/,"",sc)}
/Occupancy/ {o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
/VGPRs Spill/ {sp=$0; sub(/.*: /,"",sp); sub(/ \[.*/,"",sp)}
/LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-70s vgpr %s agpr %s sgpr %s scratch %s spill %s occ %s lds %s\n", name, v, a, sg, sc, sp, o, l}'
