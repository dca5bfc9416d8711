# tools/e2e_pinned.sh <tag> [frames]: uvolenc from files, default against --pinned-text (OBJ text read into page-locked slabs), same inputs
TAG=$1; N=${2:-960}; O=gpurun_out/$TAG; mkdir -p $O; D=/tmp/e2e_pin
run() { name=$1; shift; rm -rf $D/out; UVOL_TIMING=1 python tools/e2e_files.py $D $N "$@" > $O/e2e_$name.json 2> $O/e2e_$name.err; cat $O/e2e_$name.json; }
run default
mv $D/out $D/out_ref
run pinned --pinned-text
diff -rq $D/out_ref $D/out > $O/diff_pinned.txt 2>&1; echo "diff rc=$? ($(wc -l < $O/diff_pinned.txt) lines)" | tee -a $O/diff_pinned.txt
run pinned_run2 --pinned-text
run default_run2
rm -rf $D
