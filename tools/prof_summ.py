import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total ms/iter", round(tot/7e6,3))
pat=sys.argv[2:] 
for r in rows[:int(24)]:
    n=r['Name']; n=n.replace('void pg::','').replace('pg::','')[:52]
    print("%-52s %4s %7.3f ms/it %5.2f%%"%(n,r['Calls'],int(r['TotalDurationNs'])/7e6,float(r['Percentage'])))
