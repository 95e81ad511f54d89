import csv,sys,glob
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)[0]
    print(d)
    for r in list(csv.DictReader(open(f)))[:10]:
        print(f"  {r['Name'][:58]:58s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f}us")
