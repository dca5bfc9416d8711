cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
E="--host-pinned --host-enqueued --steps 5 --warmup 1"
bash tools/r6_host.sh r6j "both:X=1:$E" "geo:X=1:$E --only geo" "tex:X=1:$E --only tex" "both_blocking:X=1:--host-pinned --steps 3 --warmup 1" "pageable_blocking:X=1:--steps 3 --warmup 1" "pageable_enq:X=1:--host-enqueued --steps 4 --warmup 1" "both_a4:UVOL_UPLINK_AHEAD=4:$E" "both_steps8:X=1:--host-pinned --host-enqueued --steps 8 --warmup 1"
