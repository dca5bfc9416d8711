# round 4: uvolenc from files with both ingest halves on the device (un-filter asynchronous, one event per slot), twice per setting (the host is shared)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4p; mkdir -p $O
D=/tmp/uvol_e2e
rm -rf $D; timeout 1500 python tools/e2e_files.py $D 960 > $O/e2e_960.json 2>> $O/err.log
for R in 1 2; do for A in "" "--host-png-unfilter" "--host-png-unfilter --host-obj-parser" "--batch-frames 240"; do
  rm -rf $D/out; (cd $D && UVOL_TIMING=1 timeout 600 $GRAFT_REPO_ROOT/universal-volumetric_amd/bin/uvolenc project-config.json --batch-frames 120 $A > "$O/e2e_960_run${R}_$(echo $A | tr ' -' '__').txt" 2> "$O/e2e_960_run${R}_$(echo $A | tr ' -' '__')_timing.txt")
done; done
rm -rf $D
grep "frames/s" $O/*.txt
