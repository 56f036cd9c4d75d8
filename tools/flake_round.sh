#!/bin/bash
# round 5: discriminating runs of the B = 8 label flake (tools/label_flake_probe.py under SIU3R_PP_DBG; csrc/postprocess.hip)
#   1 plain loads (the round-4 symptom), 3 + a stream synchronisation between the volume's writer and the argmax, 5 + NaN-filled volume,
#   9 plain argmax twice back to back, 8 plain then system-scope, 0 shipped form
for d in ${FLAKE_MODES:-1 3 5 9 8 0}; do
  SIU3R_PP_DBG=$d DBG_N=${DBG_N:-24} timeout 300 python tools/label_flake_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
done
