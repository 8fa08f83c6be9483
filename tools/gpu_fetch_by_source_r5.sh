# r05: where does the fused convs' HBM fetch excess come from?  (1) FETCH_SIZE calibrated on a known nt weight stream, (2) per class, (3) by source
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5fetch}; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/exp/weight_prefetch_chain.hip -o /tmp/wpc 2>/dev/null
cd /tmp
for v in none noweights hot; do
  rm -rf /tmp/cal_$v
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal_$v -- /tmp/wpc $v > $O/cal_$v.log 2>&1
done
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/${1:-r5fetch}/calibration.txt
import csv, glob
res = {}
for v in ("none", "noweights", "hot"):
    vals = []
    for f in glob.glob(f"/tmp/cal_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "k_layer" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]) * 1024.0)
    vals = vals[64:]                       # first replay: cold
    res[v] = sum(vals) / max(len(vals), 1)
    print(f"{v:10s} dispatches {len(vals):4d}  FETCH_SIZE per launch {res[v] / 1e6:8.2f} MB (as counted)")
known = 256 * 72 * 1024
w = res["none"] - res["noweights"]
print(f"weights of a launch: known {known / 1e6:.2f} MB of nt dwordx4 loads (distinct per launch, 302 MB rotation); counted {w / 1e6:.2f} MB -> calibration factor {known / w:.3f}")
print(f"CAL {known / w:.4f}")
PY
CAL=$(grep "^CAL" $O/calibration.txt | awk '{print $2}')
rm -rf /tmp/fp /tmp/fn
SF_FCX_PLAIN=product SF_FCX_EVALS=4 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fp -- python $GRAFT_REPO_ROOT/tools/fconv4_knockout.py 1 > $O/fp.log 2>&1
SF_FCX_PLAIN=1 SF_FCX_EVALS=4 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fn -- python $GRAFT_REPO_ROOT/tools/fconv4_knockout.py 1 > $O/fn.log 2>&1
cd $GRAFT_REPO_ROOT
for d in fp fn; do mkdir -p $O/$d; cp $(find /tmp/$d -name "*counter_collection.csv" | head -1) $O/$d/counter_collection.csv; done
python tools/fetch_by_source.py /tmp/fp /tmp/fn $CAL 2>&1 | grep -v amdgpu.ids | tee $O/r05_unet_fetch_by_source.log
