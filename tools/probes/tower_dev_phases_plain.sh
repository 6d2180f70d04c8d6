#!/bin/bash
# developer tool, GPU box: tools/tower_phases.py (publish / barrier / weight-issue columns) of developer builds
for L in "$@"; do echo "== $L"; JU_TEST_HOOKS=1 JU_LIBRARY=build/ab/lib_$L.so python tools/tower_phases.py 2>&1 | head -3; done
