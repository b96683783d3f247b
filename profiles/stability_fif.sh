cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5; do for f in 2 3 4; do
  echo -n "run $r F=$f : "
  timeout -k 5 100 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --frames-in-flight $f --isolated-frames 0 --present-frames 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'])"
done; done
