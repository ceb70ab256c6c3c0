set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_g/p1 -o p1 --output-format csv -- python3 $R/tools/prof_stage.py 256 3 > $R/gpurun_out/pmc_g_p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $R/gpurun_out/pmc_g/p2 -o p2 --output-format csv -- python3 $R/tools/prof_stage.py 256 3 > $R/gpurun_out/pmc_g_p2.log 2>&1
cd $R
python tools/pmc_collect.py gpurun_out/pmc_g/p1 gpurun_out/pmc_g/p2 > gpurun_out/pmc_g.json
rm -rf gpurun_out/pmc_g/*/*kernel_trace.csv
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_g.json'))['frame_fused']
tiles=345600.0
for k,v in sorted(d.items()):
    if isinstance(v,(int,float)) and v: print(k, round(v/1e6,2),'M', round(v/tiles,1),'per tile')
PY
