#!/bin/bash
# Alternating A/B runs of the bench step under different environments (one script for every switch; replaces the ten
# one-off ab_r05*.sh of round 5).  On the GPU box:
#   bash tools/ab.sh [-r REPS] [-v VOXELS] [-s STEPS] [-x "extra bench.py flags"] name1:ENV=1,ENV2=x name2: ...
# prints one line per run: name, mean ms per step, p50 / p90 / max of the per-step device times.  A name with nothing after the colon is the default build.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPS=3; V=""; STEPS=20; EXTRA=""
while getopts "r:v:s:x:" o; do case $o in r) REPS=$OPTARG;; v) V=$OPTARG;; s) STEPS=$OPTARG;; x) EXTRA=$OPTARG;; esac; done
shift $((OPTIND - 1))
for rep in $(seq 1 "$REPS"); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    # shellcheck disable=SC2086
    env $(echo "$envs" | tr ',' ' ') timeout 400 python bench.py --steps "$STEPS" --warmup 5 --no-cpu-baseline --no-zorder \
        ${V:+--voxels $V} $EXTRA 2>/dev/null |
      python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('${name}_$rep', round(r['ms_per_step'],2), 'p50', round(c['step_ms_p50'],2), 'p90', round(c['step_ms_p90'],2), 'max', round(c['step_ms_max'],2), 'at', c.get('step_ms_max_at'), flush=True)"
  done
done
