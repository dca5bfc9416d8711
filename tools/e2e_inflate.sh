# tools/e2e_inflate.sh <tag> [frames]: uvolenc from files with the PNG inflate on the host (default) and on the device (--device-inflate, two
# texture batch sizes), same inputs, outputs compared; one JSON line per run under gpurun_out/<tag>/
TAG=$1; N=${2:-960}; O=gpurun_out/$TAG; mkdir -p $O; D=/tmp/e2e_inf
run() { name=$1; shift; rm -rf $D/out; UVOL_TIMING=1 python tools/e2e_files.py $D $N "$@" > $O/e2e_$name.json 2> $O/e2e_$name.err; cat $O/e2e_$name.json; }
run host
mv $D/out $D/out_host
run dev480 --device-inflate --tex-batch-frames 480
diff -rq $D/out_host $D/out > $O/diff_dev480.txt 2>&1; echo "diff rc=$? ($(wc -l < $O/diff_dev480.txt) lines)" | tee -a $O/diff_dev480.txt
run dev960 --device-inflate --tex-batch-frames 960

rm -rf $D
