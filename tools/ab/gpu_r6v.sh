# Round 6: the other configurations' lines at HEAD (configs[2]: 6 input views + EFT feature render in the step; configs[4] share: fp16 operands, full trajectory, 512^2 render_batched; configs[3] share).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6v}
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-also-measured --no-traffic > $O/r06_bench_n1_config2.json 2> $O/c2.err; tail -n 1 $O/r06_bench_n1_config2.json | cut -c1-300
timeout 400 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-also-measured --no-traffic > $O/r06_bench_n1_config4.json 2> $O/c4.err; tail -n 1 $O/r06_bench_n1_config4.json | cut -c1-300
timeout 400 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-also-measured --no-traffic > $O/r06_bench_n1_config3.json 2> $O/c3.err; tail -n 1 $O/r06_bench_n1_config3.json | cut -c1-300
