#!/bin/bash
# tools/cold_path.cpp on seven never-seen clouds of one configuration (default cfg2_1m_s256); run on the GPU box from the repo root.
set -e
cd "$(dirname "$0")/.."
CFG=${1:-cfg2_1m_s256}
D=$(mktemp -d)
python3 - "$CFG" "$D" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from polishpathplanning_amd import engine, synth
name, d = sys.argv[1], sys.argv[2]
base = sorted(synth.CONFIGS).index(name) + 1
for k in range(8):
    pts, cfg = synth.make_config(name, seed=base + 7919 * (k + 1))
    engine.save_pcd(os.path.join(d, "c%d.pcd" % k), np.ascontiguousarray(pts), binary=True)
PY
/opt/rocm/bin/hipcc -O2 -std=c++17 -I include -o "$D/cold_path" tools/cold_path.cpp -L polishpathplanning_amd -lppp_hip -Wl,-rpath,"$PWD/polishpathplanning_amd"
"$D/cold_path" "$D"/c*.pcd
PPP_COLD_WAITING_CALL=1 "$D/cold_path" "$D"/c*.pcd | tail -1 | sed 's/^/waiting for the bounds (ppp_set_cloud_device): /'
PPP_COLD_COPY_BEFORE=1 "$D/cold_path" "$D"/c*.pcd | tail -1 | sed 's/^/each cloud copied to the device right before it is timed: /'
PPP_COLD_COPY_BEFORE=1 PPP_COLD_WAITING_CALL=1 "$D/cold_path" "$D"/c*.pcd | tail -1 | sed 's/^/  ... and waiting for the bounds: /'
PPP_COLD_STREAM=1 "$D/cold_path" "$D"/c*.pcd | grep "^stream"
rm -rf "$D"
