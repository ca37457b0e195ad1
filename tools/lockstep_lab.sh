# pairs/s of B pairs per lock-step group x STREAMS host threads / streams (tools/lockstep_lab.py), 384 pairs each:
#   gpurun -- 'bash tools/lockstep_lab.sh > gpurun_out/r05_lockstep_lab.md'
echo "# tools/lockstep_lab.py B STREAMS 384 (MI355X, round 5, final library): pairs/s of STREAMS host threads, each running lock-step groups of B pairs on its stream; B = 1: rdm_engine_run per pair (the schedule of rounds 1-4)"
echo
for cfg in "1 1" "1 2" "1 4" "2 4" "3 4" "4 1" "4 2" "4 3" "4 4" "4 6" "8 2" "8 4"; do
  python tools/lockstep_lab.py $cfg 384 2>&1 | grep -A1 "pairs/s" | sed 's/^/    /'
done
