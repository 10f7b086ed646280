import csv, glob, collections, sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")[-24:]
            if "dcz" not in r["Kernel_Name"] or "gen_" in r["Kernel_Name"]: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in agg.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    print("== %s  dur %.3f ms" % (k, sum(dur[k])/len(dur[k])/1e6))
    print("   "+"  ".join("%s=%.3g"%(c.replace("SQ_",""),m[c]) for c in sorted(m)))
    if all(k in m for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                            "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")):
        wc=m["SQ_WAVE_CYCLES"]
        print("   wait_any %.0f%%  wait_inst %.0f%%  active %.0f%%  | valu-active/wavecyc %.0f%%  lds-active %.0f%%  bankconf/ldsidx %.0f%%" % (100*m["SQ_WAIT_ANY"]/wc,100*m["SQ_WAIT_INST_ANY"]/wc,100*m["SQ_ACTIVE_INST_ANY"]/wc,100*m["SQ_ACTIVE_INST_VALU"]/wc,100*m["SQ_ACTIVE_INST_LDS"]/wc,100*m["SQ_LDS_BANK_CONFLICT"]/max(1,m["SQ_LDS_IDX_ACTIVE"])))
