import csv, collections, glob, sys
for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % sys.argv[1]):
    d=collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_model_setup" in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"]); d[int(r["Dispatch_Id"])]["grid"]=float(r["Grid_Size"])
    for k in sorted(d)[-6:]:
        c=d[k]; w=c["grid"]/64
        print(k, "VALU/wave %.0f" % (c["SQ_INSTS_VALU"]/w), "wave_cycles %.0f" % (4*c["SQ_WAVE_CYCLES"]/w), "busy %.1f%%" % (100*c["SQ_ACTIVE_INST_VALU"]/c["SQ_WAVE_CYCLES"]*2))
