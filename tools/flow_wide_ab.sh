for r in 1 2; do
  echo "== JU_FLOW_WIDE=0"; JU_FLOW_WIDE=0 python3 tools/flow_layers.py | head -14
  for th in 2 4 6; do echo "== wide, JU_FLOW_TILE=$th"; JU_FLOW_TILE=$th python3 tools/flow_layers.py | head -12; done
  echo "== wide, default tile"; python3 tools/flow_layers.py | head -14
done
