"""dev: summary of a NNOP_TEST_ERRLOG file (one JSON line per comparison of the GPU suite, tests/util.py::record_error) ->
per dtype / tensor: count, worst error relative to the tensor's max, worst error / tolerance.   usage: summarize_errlog.py log.jsonl out.json"""
import collections, json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
per = collections.defaultdict(lambda: dict(n=0, worst_rel_to_max=0.0, worst_ratio=0.0))
other = collections.defaultdict(list)
for r in rows:
    if r.get("kind") == "assert_close":
        e = per[f'{r["dt"]}/{r["name"]}']
        e["n"] += 1
        e["worst_rel_to_max"] = max(e["worst_rel_to_max"], r["rel_to_max"])
        e["worst_ratio"] = max(e["worst_ratio"], r["worst_ratio"] / max(r.get("scale", 1.0), 1e-30) * r.get("scale", 1.0))
    else:
        other[r.get("kind", "?")].append({k: v for k, v in r.items() if k != "kind"})
by_dt = collections.defaultdict(float)
for k, e in per.items():
    by_dt[k.split("/")[0]] = max(by_dt[k.split("/")[0]], e["worst_ratio"])
out = dict(source="NNOP_TEST_ERRLOG=... python -m pytest tests -m gpu", comparisons=sum(e["n"] for e in per.values()),
           worst_error_over_tolerance_by_dtype=dict(by_dt), per_tensor=dict(sorted(per.items())), **other)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(dict(comparisons=out["comparisons"], worst=out["worst_error_over_tolerance_by_dtype"]), indent=1))
