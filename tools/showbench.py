import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    st = j["stage_ms_per_step"]
    print(f, "%.3f Gbp/s %.1f ms" % (j["value"], j["ms_per_step"]), " ".join("%s=%.1f" % (k, v) for k, v in st.items() if v > 0.05))
