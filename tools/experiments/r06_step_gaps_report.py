import csv, glob
rows=[]
for f in glob.glob("/tmp/sg/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-70:], r.get("Queue_Id","?")))
rows.sort()
idx=[i for i,r in enumerate(rows) if "score_kernel_coop" in r[2]]
print(len(rows), len(idx))
a=idx[-6]
t0=rows[a][0]
for r in rows[a:idx[-3]+1]:
    print(round((r[0]-t0)/1e3,1), round((r[1]-r[0])/1e3,1), r[2], "q", r[3])
