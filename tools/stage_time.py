import csv,glob,sys
from collections import defaultdict
root=sys.argv[1]
names={1:"staging",2:"level 0",3:"fill levels",5:"merges",6:"anti-collapse",7:"denormalise",0:"IMDCT + row store"}
prev=0
for k in [1,2,3,5,6,7,0]:
    tot=defaultdict(float); n=defaultdict(int)
    for f in glob.glob(f"{root}/stop{k}/**/*counter_collection.csv",recursive=True):
        per=defaultdict(float); kn={}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in("GRBM_GUI_ACTIVE","SQ_WAVE_CYCLES","SQ_WAIT_ANY"):
                per[(r["Dispatch_Id"],r["Counter_Name"])]+=float(r["Counter_Value"]); kn[r["Dispatch_Id"]]=r["Kernel_Name"]
        for (d,c),v in per.items():
            if kn[d].startswith("og_celt_kernel"): tot[c]+=v; n[c]+=1
    g=tot["GRBM_GUI_ACTIVE"]/n["GRBM_GUI_ACTIVE"]/8
    print(f"{names[k]:20s} cum {g/2400:8.1f} us  stage {(g-prev)/2400:8.1f} us   wave_cycles/frame {tot['SQ_WAVE_CYCLES']/n['SQ_WAVE_CYCLES']/32768:9.0f} wait_any {tot['SQ_WAIT_ANY']/n['SQ_WAIT_ANY']/32768:9.0f}")
    prev=g
