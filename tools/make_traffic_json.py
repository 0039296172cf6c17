"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE PMC passes (tools/pmc_final.sh).

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches. The factor 2 is the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section), re-calibrated here on
k_accumulate, whose reads are known exactly (16 B per path + 16 B per pixel, all dwordx4).

    python tools/make_traffic_json.py <pmc dir> <workload> [out.json]
"""
import json, os, re, sys


def counters(md):
    out = {}
    for line in open(md):
        m = re.match(r"\| (k_[a-z_]+)(?:<[^>]*>)? \| ([A-Z_]+) \| ([0-9.e+]+) \| (\d+) \|", line)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
    return out


def main():
    d, workload = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    f, w = counters(os.path.join(d, "fetch.md")), counters(os.path.join(d, "write.md"))
    res = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for k in ("k_trace_closest", "k_trace_shadow", "k_shade", "k_accumulate"):
        if k in f and k in w:
            fs, n = f[k]["FETCH_SIZE"]
            ws, _ = w[k]["WRITE_SIZE"]
            res.setdefault(k, {})[workload] = round((2 * fs + ws) * 1024 / n)
            res[k][workload + "_detail"] = {"FETCH_SIZE_KB_per_launch": round(fs / n, 1), "WRITE_SIZE_KB_per_launch": round(ws / n, 1),
                                           "launches": n, "fetch_correction": 2}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
