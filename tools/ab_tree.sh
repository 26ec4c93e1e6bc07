set -e
F='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["counters"]["casts"])'
for S in scenes/MirrorBombardment-200M.json scenes/MirrorKnot4k.json; do
B="timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --scene $S"
echo $S
RT_AMD_VERBOSE=1 RT_AMD_TREE=reference $B | python -c "$F" reference-tree
RT_AMD_VERBOSE=1 $B | python -c "$F" default
for SL in 0 1 2 3 4; do RT_AMD_VERBOSE=1 RT_AMD_DEPTH_SLACK=$SL $B | python -c "$F" leaf4-slack$SL; done
for SL in 1 3; do RT_AMD_VERBOSE=1 RT_AMD_LEAF=6 RT_AMD_DEPTH_SLACK=$SL $B | python -c "$F" leaf6-slack$SL; done
for SL in 2 4; do RT_AMD_VERBOSE=1 RT_AMD_LEAF=2 RT_AMD_DEPTH_SLACK=$SL $B | python -c "$F" leaf2-slack$SL; done
done
