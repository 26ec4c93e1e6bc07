# Trace time of a scene for several layouts of the parked photons' ray key (RT_AMD_LEVEL_SORT):
#   bash tools/sweep_level_sort.sh "0x75 0x95" [scene name]
for M in ${1:-0x75 0x95 0xB5 0x74 0x94 0xF5}; do
  echo "== level sort layout $M"
  RT_AMD_LEVEL_SORT=$M timeout -k 10 300 python -u tools/stage_times.py --repeat 2 scenes/${2:-MirrorKnot4k}.json 2>&1 | grep -E "it1" | sed -e "s/setup.*emitters [0-9.]*//" -e "s/gather.*//"
done
