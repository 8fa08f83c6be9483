cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2d}
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -5 > $O/t_fused.log
python tools/fconv_phases.py > $O/phases.log 2>&1
python tools/unet_profile.py 1 1024 > $O/prof_fused.log 2>&1
HIP_FORCE_DEV_KERNARG=1 python tools/unet_profile.py 1 1024 > $O/prof_fused_devkernarg.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 30 > $GRAFT_REPO_ROOT/$O/rp.log 2>&1
HIP_FORCE_DEV_KERNARG=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp2 -- python $GRAFT_REPO_ROOT/tools/unet_eval_loop.py 1 30 > $GRAFT_REPO_ROOT/$O/rp2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py /tmp/rp1 $O/timeline_fused.txt
python tools/trace_timeline.py /tmp/rp2 $O/timeline_fused_devkernarg.txt
tail -n 3 $O/t_fused.log
cat $O/phases.log
grep "wall\|== B" $O/prof_fused.log $O/prof_fused_devkernarg.log
grep "^# launches" $O/timeline_fused.txt $O/timeline_fused_devkernarg.txt
grep "^# " $O/timeline_fused_devkernarg.txt | head -30
