mkdir -p gpurun_out/pmc1; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(SQ_[A-Z_0-9]+|TCP_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+|TA_[A-Z_0-9a-z]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|[A-Za-z]+Busy|[A-Za-z]*Util[a-z]*|L2CacheHit|LDSBankConflict|MemUnit[A-Za-z]*)" | sort -u | tr '\n' ' ' > gpurun_out/pmc1/counters.txt
rocprofv3 -L > gpurun_out/pmc1/avail.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc1/p$i -o p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1/p$i.log 2>&1
  tail -1 gpurun_out/pmc1/p$i.log | cut -c1-120
done
ls -R gpurun_out/pmc1 | head -40
