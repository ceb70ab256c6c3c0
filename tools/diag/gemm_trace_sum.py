import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_gemm_nt_h" in r["Kernel_Name"] or "k_gemm_reduce" in r["Kernel_Name"]]
# groups of 10 GEMM + 10 reduce
i = 0; out = []
g = [r for r in rows if "k_gemm_nt_h" in r["Kernel_Name"]]
rd = [r for r in rows if "k_gemm_reduce" in r["Kernel_Name"]]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k in range(0, len(g), 10):
    grp = g[k:k + 10][3:]; rg = rd[k:k + 10][3:]
    gap = [(int(rg[j]["Start_Timestamp"]) - int(grp[j]["End_Timestamp"])) / 1e3 for j in range(len(grp))]
    print("%-40s grid %s x %s x %s  gemm %.1f us  reduce %.1f us  gap %.1f us" % (grp[0]["Kernel_Name"][:40], grp[0]["Grid_Size_X"], grp[0]["Grid_Size_Y"], grp[0]["Grid_Size_Z"],
          sum(map(d, grp)) / len(grp), sum(map(d, rg)) / len(rg), sum(gap) / len(gap)))
