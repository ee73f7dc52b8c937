import sys, json
for f in sys.argv[1:]:
    lines=[l for l in open(f) if l.startswith("{")]
    # the full JSON line is the one with stages_ms
    for l in lines:
        d=json.loads(l)
        if d.get("stages_ms"):
            st={k:v for k,v in d["stages_ms"].items() if k!="note"}
            print(f, d["ms_per_step"], json.dumps(st))
