#!/bin/bash
# developer tool, GPU box: the phase profile (tools/tower_phase_map.py) of several developer builds of the tower
# usage: tools/probes/tower_dev_phases.sh <name>...   (build/ab/lib_<name>.so from tools/dev_tower_lib.sh)
for L in "$@"; do echo "== $L"; JU_TEST_HOOKS=1 JU_LIBRARY=build/ab/lib_$L.so python tools/tower_phase_map.py 2>&1 | grep -e diagnostic -e "xcd" ; done
