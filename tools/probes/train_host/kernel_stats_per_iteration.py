import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
n_it=int(sys.argv[2])
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total per it (ms)", tot/n_it/1e6, "launches per it", sum(int(r['Calls']) for r in rows)/n_it)
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 40]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls'])/n_it:7.1f} {float(r['TotalDurationNs'])/n_it/1e3:8.1f}us {float(r['AverageNs'])/1e3:7.2f}us")
