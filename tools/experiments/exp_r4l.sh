# round 4: PNG scanlines un-filtered on the device: GPU parity + uvolenc from files, device against host un-filter
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tex.py tests/test_gpu_cli.py -x -q -k "png_scanlines or uvolenc or shim" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
D=/tmp/uvol_e2e
rm -rf $D; timeout 1500 python tools/e2e_files.py $D 960 > $O/e2e_960.json 2>> $O/err.log
for A in "" "--host-png-unfilter" "--batch-frames 60" "--batch-frames 240"; do
  rm -rf $D/out; (cd $D && UVOL_TIMING=1 timeout 600 $GRAFT_REPO_ROOT/universal-volumetric_amd/bin/uvolenc project-config.json --batch-frames 120 $A > "$O/e2e_960_$(echo $A | tr ' -' '__').txt" 2> "$O/e2e_960_$(echo $A | tr ' -' '__')_timing.txt")
done
rm -rf $D
tail -3 $O/pytest.log; grep "frames/s" $O/*.txt
