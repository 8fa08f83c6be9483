cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc1}
mkdir -p $O
export TMPDIR=/tmp
export SF_TIMING_LIB=$GRAFT_REPO_ROOT/sparsefusion_amd/libsparsefusion_hip.so
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d /tmp/p2 -- python $GRAFT_REPO_ROOT/tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d /tmp/p3 -- python $GRAFT_REPO_ROOT/tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d /tmp/p4 -- python $GRAFT_REPO_ROOT/tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 > $O/p4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/p5 -- python $GRAFT_REPO_ROOT/tools/fconv_phases.py unet_32x32_512 unet_8x8_1536 > $O/p5.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_collect.py /tmp/p1 "k_conv_fused<2, 2, 8, 2" /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5 > $O/fconv_32x32_512.json
python tools/pmc_collect.py /tmp/p1 "k_conv_fused<1, 1, 12, 2" /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5 > $O/fconv_8x8_1536.json
tail -3 $O/p1.log
cat $O/fconv_32x32_512.json | python -c "import sys,json; d=json.load(sys.stdin); [print(k, round(v['mean_per_dispatch']), v['dispatches']) for k,v in d.items()]"
echo ---
cat $O/fconv_8x8_1536.json | python -c "import sys,json; d=json.load(sys.stdin); [print(k, round(v['mean_per_dispatch']), v['dispatches']) for k,v in d.items()]"
