import json,sys
for f in sys.argv[1:]:
    try:
        l=[x for x in open(f) if x.startswith('{"metric')][-1]
        j=json.loads(l); k=j["roofline"]["kernels_ms_per_step"]
        print(f.split('/')[-1], "step", j["ms_per_step"], "k_fm", k.get("k_fm"), "validated", j.get("validated"), {a:round(b,3) for a,b in k.items() if a!="k_fm"})
    except Exception as e: print(f, "ERR", e)
