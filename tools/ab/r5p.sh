cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O
export TMPDIR=/tmp
for b in 2 4; do
  for at in "" "conv4_slices_at_any_batch=1"; do echo "== B=$b $at"; SF_UNET_ATTRS=$at timeout 300 python tools/unet_time.py $b 2>&1 | grep "sampler path"; done
done | tee $O/unet_ab.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_n1_final.json 2> $O/bench.err
python -c "
import json; r=json.loads(open('gpurun_out/r5p/r05_bench_n1_final.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('traffic'), r['roofline'].get('traffic_over_algorithmic'), r['roofline']['frac_whole_eval'], {k:(v.get('value'), v.get('ms_per_step')) for k,v in r['also_measured'].items()})"
