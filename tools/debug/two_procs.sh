#!/bin/bash
# experiment: does a second concurrent encode stream raise the aggregate rate?  one process alone, then two at once
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/prof_codec.py --reps 20 --what enc 2>&1 | grep "^enc" | tail -3
echo "-- two processes"
(timeout -k 10 200 python tools/prof_codec.py --reps 20 --what enc 2>&1 | grep "^enc" | tail -3 | sed 's/^/A /') &
pa=$!
(timeout -k 10 200 python tools/prof_codec.py --reps 20 --what enc 2>&1 | grep "^enc" | tail -3 | sed 's/^/B /') &
pb=$!
wait $pa $pb
