#!/bin/bash
# VERDICT r4 next 3, lab: per-kernel-class cost per ORIGINAL pair when the rows of B pairs go through one launch sequence
# (tools/batch_lab.py), and the end-to-end rates of the combinations (streams x B).   gpurun -- 'bash tools/batch_lab.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/batch_lab; rm -rf $O; mkdir -p $O
db() { find "$1" -name "*.db" | head -1; }
{
echo "# rates (original pairs per second; 48 stacked runs after warm-up)"
for cfg in "1 1" "1 4" "2 1" "2 2" "2 4" "4 1" "4 2" "4 4"; do python tools/batch_lab.py $cfg 2>/dev/null | tail -1; done
} > $O/rates.txt
cat $O/rates.txt
for B in 1 4; do
  rocprofv3 --kernel-trace --stats -d $O/t$B -- python tools/batch_lab.py $B 1 24 > $O/t$B.log 2>&1
  python tools/batch_lab_classes.py $(db $O/t$B) $B > $O/classes_B$B.txt
done
paste -d'\n' $O/classes_B1.txt /dev/null > /dev/null
python tools/batch_lab_classes.py $(db $O/t1) 1 $(db $O/t4) 4 > $O/classes.md
cat $O/classes.md
find $O -name "*.db" -delete; find $O -type d -empty -delete
