cd $GRAFT_REPO_ROOT
show='import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], "in flight", j["frames_in_flight"], j["value"], "Mrays/s", j["ms_per_step"], "ms/step")'
for n in 4 8; do for f in 4 6 8; do
    timeout -k 5 120 python bench.py --steps 48 --warmup 8 --emulate-rank-of $n --frames-in-flight $f --isolated-frames 0 --present-frames 0 2>/dev/null | python3 -c "$show" "one rank of $n"
done; done
