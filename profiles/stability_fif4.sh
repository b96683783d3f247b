cd $GRAFT_REPO_ROOT
for cfg in "20 3" "15 3" "40 6" "20 4" "21 3" "20 3"; do set -- $cfg
  for extra in "" "--isolated-frames 0 --present-frames 0"; do
  echo -n "F=4 steps $1 warmup $2 $extra : "
  timeout -k 5 100 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --frames-in-flight 4 $extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'])"
done; done
