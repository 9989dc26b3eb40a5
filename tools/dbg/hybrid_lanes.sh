for cfg in "" "LDM_CHUNK=512 LDM_LANES=1" "LDM_CHUNK=128 LDM_LANES=4" "LDM_CHUNK=128 LDM_LANES=2" "LDM_CHUNK=256 LDM_LANES=1" "LDM_LANE_OFFSET_US=0" "LDM_LANE_OFFSET_US=150"; do
  echo "== [$cfg]"; env LDM_DEV=1 $cfg PROBE_PREC=hybrid timeout 200 python tools/mixed_probe.py 2>&1 | grep "LOOP" | tail -2
done
