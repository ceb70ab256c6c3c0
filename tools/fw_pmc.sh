# usage: tools/fw_pmc.sh <tag> [dbg]   (AOMARL_LIB selects the library build)
set -e
R=$PWD; TAG=$1; DBG=${2:-0}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_$TAG/p$i -o p$i --output-format csv -- python3 $R/tools/fw_pmc.py 256 $DBG > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
tot=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_$TAG/**/*counter_collection.csv",recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_frame_wave" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
    for d in sorted(per)[-3:]:
        for k,v in per[d].items(): tot[k].append(v)
m={k:sum(v)/len(v) for k,v in tot.items()}
print("$TAG dbg=$DBG")
for k in sorted(m): print("  %-28s %.4g" % (k, m[k]))
if "FETCH_SIZE" in m: print("  fetch GB (x2 corr) %.3f  write GB %.4f  L2 hit %.3f" % (m["FETCH_SIZE"]*2048/1e9, m.get("WRITE_SIZE",0)*1024/1e9, m.get("TCC_HIT_sum",0)/max(1,m.get("TCC_HIT_sum",0)+m.get("TCC_MISS_sum",0))))
by_size=None
if "TCC_EA0_RDREQ_sum" in m and "TCC_EA0_RDREQ_128B_sum" in m:
    # read requests of the L2s to the fabric by size: the byte count that needs no calibration
    n32, n64, n128, tot_rq = m["TCC_EA0_RDREQ_32B_sum"], m["TCC_EA0_RDREQ_64B_sum"], m["TCC_EA0_RDREQ_128B_sum"], m["TCC_EA0_RDREQ_sum"]
    by_size = 32.*n32 + 64.*n64 + 128.*n128
    print("  read requests: %.4g total = %.4g x 32 B + %.4g x 64 B + %.4g x 128 B (+ %.4g of other / unclassified) -> %.3f GB;  to DRAM: %.4g requests" %
          (tot_rq, n32, n64, n128, tot_rq - n32 - n64 - n128, by_size / 1e9, m.get("TCC_EA0_RDREQ_DRAM_sum", float("nan"))))
import json
name=open("gpurun_out/fw_kernel_name.txt").read().strip()
if "FETCH_SIZE" in m:
    # MI355X_MICROARCH.md: FETCH_SIZE counts 64-B units but reports in KiB of 32-B requests on gfx950 (x2);
    # WRITE_SIZE in KiB
    fetch=m["FETCH_SIZE"]*2048.; write=m.get("WRITE_SIZE",0)*1024.
    out={"_config":"production_sh_40x40_8m_3layers","_envs":256,"_how":"tools/fw_pmc.sh: rocprofv3 --kernel-trace --pmc, one pass per counter group, mean of the last 3 launches of the kernel; frame kernel alone (tools/fw_pmc.py)",
         "frame_fused":{"kernel":name,"hbm_traffic_bytes_per_launch":fetch+write,"fetch_bytes":fetch,"write_bytes":write,
                        "l2_hit":m.get("TCC_HIT_sum",0)/max(1,m.get("TCC_HIT_sum",0)+m.get("TCC_MISS_sum",0)),
                        "fetch_bytes_by_request_size":by_size,
                        "counters":{k:m[k] for k in sorted(m)}}}
    json.dump(out,open("gpurun_out/$TAG"+"_pmc_frame_kernel.json","w"),indent=1)
if "SQ_WAVE_CYCLES" in m: print("  wait_any/wave_cycles %.3f  active_any %.3f  wait_inst %.3f  valu/tilewave %.1f mfma %.1f lds %.1f vmem %.1f" % (m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"]/335872., m["SQ_INSTS_MFMA"]/335872., m["SQ_INSTS_LDS"]/335872., m["SQ_INSTS_VMEM_RD"]/335872.))
PY
rm -rf gpurun_out/pmc_$TAG
