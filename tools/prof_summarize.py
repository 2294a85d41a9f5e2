"""Condense rocprofv3 output directories (under gpurun_out/) into the summaries kept in profiles/.

usage: python tools/prof_summarize.py <round-tag>
  expects gpurun_out/prof_stats_<MODE>/ (--kernel-trace --stats), gpurun_out/prof_fetch_<MODE>/ and
  gpurun_out/prof_write_<MODE>/ (--pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes), MODE in CFS, PSGCFS.
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "cfs_solve_fused_kernel"
ALGO = 585120


def find(d, suffix):
    g = glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*" + suffix), recursive=True)
    return g[0] if g else None


def counter_mean(d, name):
    f = find(d, "counter_collection.csv")
    if not f:
        return None, 0
    vals = {}
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name:
            vals.setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(vals.values())
    return (sum(v) / len(v) if v else None), len(v)


def main(tag):
    out = {"round": tag, "kernel": KERNEL, "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python bench.py "
           "--mode <MODE> --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-other-mode (separate passes; --streams 1 = serial launches)",
           "correction": "MI355X_MICROARCH.md HBM section: counters are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide "
                         "coalesced reads -> read bytes = 2*FETCH_SIZE*1024 (upper estimate: most loads of this kernel are 8 B/lane); "
                         "WRITE_SIZE*1024 is exact"}
    for mode in ("CFS", "PSGCFS"):
        st = find("prof_stats_" + mode, "kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(ROOT, "profiles", f"{tag}_bench_{mode.lower()}_kernel_stats.csv"))
            for r in csv.DictReader(open(st)):
                if KERNEL in r["Name"]:
                    out[f"kernel_avg_ns_{mode}_default_streams"] = float(r["AverageNs"])
                    out[f"kernel_calls_{mode}"] = int(r["Calls"])
        fk, n = counter_mean("prof_fetch_" + mode, "FETCH_SIZE")
        wk, _ = counter_mean("prof_write_" + mode, "WRITE_SIZE")
        if fk is not None and wk is not None:
            out[f"launches_{mode}"] = n
            out[f"FETCH_SIZE_KB_per_launch_mean_{mode}"] = fk
            out[f"WRITE_SIZE_KB_per_launch_mean_{mode}"] = wk
            out[f"hbm_bytes_per_launch_{mode}"] = 2 * fk * 1024 + wk * 1024
            out[f"hbm_bytes_per_launch_{mode}_uncorrected"] = (fk + wk) * 1024
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    shutil.copy(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), os.path.join(ROOT, "profiles", "pmc_latest.json"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
