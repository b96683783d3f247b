#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python bench.py --verify > gpurun_out/r03/bench_default_r03_2.log 2> gpurun_out/r03/bench_default_r03_2.err; echo "default rc=$?"
RTR_BENCH_FORCE_INPROC=1 RTR_MGPU_SELF_EXCHANGE=1 python bench.py --steps 100 > gpurun_out/r03/bench_inproc_one_rank_rccl_r03_2.log 2> gpurun_out/r03/bench_inproc_one_rank_rccl_r03_2.err; echo "inproc rc=$?"
python bench.py --gpus 2 --steps 5 > gpurun_out/r03/bench_gpus2_on_one_gpu_box.log 2>&1; echo "gpus2 rc=$? (expected != 0)"; tail -2 gpurun_out/r03/bench_gpus2_on_one_gpu_box.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_default_r03_2.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['hbm_frac'], d['roofline_secondary'], d['frame_hbm'])
d=json.loads([l for l in open('gpurun_out/r03/bench_inproc_one_rank_rccl_r03_2.log') if l.startswith('{')][-1])
print(d['value'], d['rccl'], d['verify'])
PY
