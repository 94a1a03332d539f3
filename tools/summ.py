import json,sys
for f in sys.argv[1:]:
    try:
        j=json.load(open(f)); r=j["roofline"]
        print(f.split('/')[-1], "Gpkt/s", round(j["value"],3), "ms/step", round(j["ms_per_step"],4), "kernel_ms", round(r["kernel_ms"],4), "GB/s", round(r["achieved"]), "frac", round(r["frac"],3), "stream", round(r["stream_read_GBps"]), "parity", j.get("parity_check"))
    except Exception as e: print(f, "ERR", e)
