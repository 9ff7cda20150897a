import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        nr = d.get("next_rows")
        if not isinstance(nr, dict) or "error" in nr:
            print(nr)
            continue
        for k, r in nr.items():
            print("%-26s %7.3f ms  frac %.3f  valid %s" % (k, r["kernel_ms"], r["frac"], r.get("mask_valid_fraction")))
