"""gpurun_out/pmc_cnn_{pairs}_{FETCH_SIZE,WRITE_SIZE}/ (scripts/pmc_cnn.sh) -> JSON: HBM bytes of one ResNet-FPN call =
sum over the dispatches from the last stem_gather_kernel on of 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB -> bytes."""
import csv, glob, json, sys
out = {"_comment": "HBM / fabric bytes of ONE pope_resnetfpn_forward_f32 call (all its kernels), rocprofv3 --pmc FETCH_SIZE and "
                   "WRITE_SIZE in separate passes (scripts/pmc_cnn.sh), 2 x FETCH_SIZE + WRITE_SIZE in KiB -> bytes; f16x3"}
for pairs in (3, 24):
    tot, per = {}, {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{sys.argv[1]}/pmc_cnn_{pairs}_{c}/**/*counter_collection.csv", recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        first = max(i for i, r in enumerate(rows) if "stem_gather" in r["Kernel_Name"])
        tot[c] = sum(float(r["Counter_Value"]) for r in rows[first:])
        tot["dispatches"] = len(rows) - first
        for k, r in enumerate(rows[first:]):
            per.setdefault(k, {"kernel": r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:48]})[c] = float(r["Counter_Value"])
    out[f"resnet_fpn_{2 * pairs}_images"] = {"bytes": int(round((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024)),
                                            "fetch_kib": tot["FETCH_SIZE"], "write_kib": tot["WRITE_SIZE"], "dispatches": tot["dispatches"],
                                            "per_dispatch_mb": [[v["kernel"], round(2 * v["FETCH_SIZE"] * 1024 / 1e6, 1), round(v["WRITE_SIZE"] * 1024 / 1e6, 1)] for _, v in sorted(per.items())]}
print(json.dumps(out, indent=1))
