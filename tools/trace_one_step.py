import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "pf_stage_queries" in r["Kernel_Name"]]
a,b=idx[len(idx)//2],idx[len(idx)//2+1]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%9.1f +%8.1f q=%s %s grid=%s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Queue_Id"],r["Kernel_Name"].split("(")[0][-40:],r["Grid_Size_X"]))
