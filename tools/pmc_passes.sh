set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
  case $i in
    1) C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES";;
    2) C="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE";;
    3) C="FETCH_SIZE";;
    4) C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_d/p$i -o p$i --output-format csv -- python3 $R/tools/prof_stage.py 256 3 > $R/gpurun_out/pmc_d_p$i.log 2>&1
  echo pass $i done
done
cd $R
python tools/pmc_collect.py gpurun_out/pmc_d/p1 gpurun_out/pmc_d/p2 gpurun_out/pmc_d/p3 gpurun_out/pmc_d/p4 --how "rocprofv3 --kernel-trace --pmc <set> -- python3 tools/prof_stage.py 256 3; four separate passes (SQ set 1, SQ set 2 + GRBM, FETCH_SIZE alone, WRITE_SIZE + TCC_HIT/MISS); mean of the last 3 dispatches per kernel family; 256 environments of production_sh_40x40_8m_3layers" > gpurun_out/r01d_pmc.json
rm -rf gpurun_out/pmc_d/*/*/*kernel_trace.csv
head -c 1500 gpurun_out/r01d_pmc.json
