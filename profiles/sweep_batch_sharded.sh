cd $GRAFT_REPO_ROOT
show='import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], j["value"], "Mrays/s", j["ms_per_step"], "ms/step", j["kernels_ms"])'
for n in 8 4 1; do for b in 64 128 256; do
    RTR_TRACE_BATCH=$b timeout -k 5 120 python bench.py --steps 48 --warmup 8 --emulate-rank-of $n --frames-in-flight 4 2>/dev/null | python3 -c "$show" "one rank of $n batch $b"
done; done
