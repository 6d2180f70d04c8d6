cd $GRAFT_REPO_ROOT
JU_FP8_PROFILE=1 JU_NO_GRAPH=1 timeout 200 python3 bench.py --no-cpu-baseline --steps 60 --warmup 5 --preset ps2-quality --dtype fp8 2>&1 | grep "fp8 conv" | tail -6
