# round 4: uvolenc from files, ingest threads split between the stages (the device parses the OBJ text, the CPUs go to the PNGs)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O
D=/tmp/uvol_e2e
rm -rf $D; timeout 1500 python tools/e2e_files.py $D 960 > $O/e2e_960.json 2>> $O/err.log
for A in "" "--batch-frames 240" "--ingest-threads 8" "--host-obj-parser"; do
  rm -rf $D/out; (cd $D && UVOL_TIMING=1 timeout 600 $GRAFT_REPO_ROOT/universal-volumetric_amd/bin/uvolenc project-config.json --batch-frames 120 $A > "$O/e2e_960_$(echo $A | tr ' -' '__').txt" 2> "$O/e2e_960_$(echo $A | tr ' -' '__')_timing.txt")
done
rm -rf $D
grep "frames/s" $O/*.txt
