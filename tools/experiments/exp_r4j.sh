# round 4: OBJ text parsed on the device: GPU parity tests + uvolenc from files with the device parser against the host parser
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O
echo skip-tests > $O/pytest2.log
D=/tmp/uvol_e2e
rm -rf $D; timeout 1500 python tools/e2e_files.py $D 960 > $O/e2e_960_device_parser.json 2>> $O/err.log
rm -rf $D/out; (cd $D && UVOL_TIMING=1 timeout 600 $GRAFT_REPO_ROOT/universal-volumetric_amd/bin/uvolenc project-config.json --batch-frames 120 > $O/e2e_960_device_parser_run2.txt 2> $O/e2e_960_device_parser_timing.txt)
rm -rf $D/out; (cd $D && UVOL_TIMING=1 timeout 600 $GRAFT_REPO_ROOT/universal-volumetric_amd/bin/uvolenc project-config.json --batch-frames 120 --host-obj-parser > $O/e2e_960_host_parser.txt 2> $O/e2e_960_host_parser_timing.txt)
rm -rf $D/out; (cd $D && timeout 600 $GRAFT_REPO_ROOT/universal-volumetric_amd/bin/uvolenc project-config.json --batch-frames 240 > $O/e2e_960_device_parser_b240.txt 2>> $O/err.log)
rm -rf $D
tail -3 $O/pytest.log; grep "frames/s" $O/*.txt
