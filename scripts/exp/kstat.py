import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.reader(open(f))):
    if 'fast_' in r[0] or 'plan_hot' in r[0]: print(r[0][:40], r[1], r[3])
