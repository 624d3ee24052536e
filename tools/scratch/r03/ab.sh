# usage: bash tools/scratch/r03/ab.sh "ENV_A" "ENV_B" [reps]  -- alternating runs on the same box
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    env $v python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$v', round(r['ms_per_step'],3))"
  done
done
