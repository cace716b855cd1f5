import csv,glob,sys,collections
d=sys.argv[1]
acc=collections.OrderedDict()
for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        key=(int(r["Dispatch_Id"]), r["Kernel_Name"][:50], r["Grid_Size"])
        acc.setdefault(key,{})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in sorted(acc.items()):
    print(k, {a:int(b) for a,b in v.items()})
