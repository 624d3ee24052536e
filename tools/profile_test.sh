#!/bin/bash
# cProfile of one pytest node on the GPU box (where does a slow oracle-bound test spend its time?)
# usage: bash tools/profile_test.sh <node id> <out.txt>
cd "$GRAFT_REPO_ROOT"
python -m cProfile -o /tmp/test.prof -m pytest "$1" -q -p no:cacheprovider > /tmp/test.log 2>&1
tail -2 /tmp/test.log
python -c "import pstats; pstats.Stats('/tmp/test.prof').sort_stats('cumulative').print_stats(70)" > "$2" 2>&1
python -c "import pstats; pstats.Stats('/tmp/test.prof').sort_stats('tottime').print_stats(40)" >> "$2" 2>&1
