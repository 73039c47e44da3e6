#!/bin/bash
# HBM-counter calibration on libfgo's access patterns (tools/pmc_calib.hip): two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE;
# --kernel-trace only, as the guide prescribes), then counter / known bytes per pattern.   $1 = output tag under gpurun_out/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -- $GRAFT_REPO_ROOT/tools/pmc_calib > $OUT/known.txt 2> $OUT/f.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -- $GRAFT_REPO_ROOT/tools/pmc_calib > /dev/null 2> $OUT/w.log
python3 $GRAFT_REPO_ROOT/tools/pmc_calib.py $OUT/known.txt $(find $OUT/f -name "*.db" | head -1) $(find $OUT/w -name "*.db" | head -1) > $OUT/calibration.txt 2>&1
cp $OUT/calibration.txt $OUT/../pmc_calibration.txt 2>/dev/null
rm -rf $OUT/f $OUT/w
cat $OUT/calibration.txt
