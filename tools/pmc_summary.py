import csv, glob, collections, sys
pat = sys.argv[1]; kern = sys.argv[2]
for f in sorted(glob.glob(pat)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
            acc["_vgpr"].append(float(r["VGPR_Count"])); acc["_sgpr"].append(float(r["SGPR_Count"]))
    print(" | ".join("%s=%.4g"%(k,sum(v)/len(v)) for k,v in acc.items()))
