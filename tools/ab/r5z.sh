cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5z; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench.err ) 2>&1 | tail -3
python -c "
import json; r=json.loads(open('gpurun_out/r5z/bench_driver_like.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('traffic_over_algorithmic'), list(r.get('also_measured',{}).keys()), r['cpu_baseline']['ms_per_step'])"
