#!/bin/bash
# rocprofv3 kernel trace of the default bench command -> per-kernel stats summary + the merged all-queue timeline of the
# last step (tools/stream_timeline.py).  On the GPU box: bash tools/prof_timeline.sh <tag> [ENV=..,ENV2=..] [extra bench flags]
cd "$GRAFT_REPO_ROOT"
T=${1:-prof}; ENVS=${2:-X=1}; EXTRA=${3:-}
O=gpurun_out/$T; mkdir -p "$O"
# shellcheck disable=SC2086
(cd /tmp && export TMPDIR=/tmp && env $(echo "$ENVS" | tr ',' ' ') rocprofv3 --kernel-trace --stats --output-format csv \
   -d "$GRAFT_REPO_ROOT/$O/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-zorder $EXTRA > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1)
python tools/prof_summary.py "$O/prof" 90 > "$O/kernel_stats_summary.txt"
python tools/stream_timeline.py "$O/prof" > "$O/timeline_last_step.txt"
python tools/timeline_sections.py "$O/timeline_last_step.txt" > "$O/timeline_sections.txt"
cp "$(ls $O/prof/*/*kernel_stats.csv | head -1)" "$O/kernel_stats.csv"
gzip -c "$(ls $O/prof/*/*kernel_trace.csv | head -1)" > "$O/kernel_trace.csv.gz"
rm -rf "$O/prof"
tail -1 "$O/prof.log"
