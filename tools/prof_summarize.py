"""Condense the rocprofv3 output of tools/profile_round.sh (under gpurun_out/) into the summaries kept in profiles/.

usage: python tools/prof_summarize.py <round-tag>
  gpurun_out/prof_stats_<MODE>/   --kernel-trace --stats          -> profiles/<tag>_bench_<mode>_kernel_stats.csv
  gpurun_out/prof_pmc<k>_<MODE>/  one --pmc group each (k = 1..7) -> profiles/<tag>_pmc.json (+ pmc_latest.json, read by bench.py)
                                                                     profiles/<tag>_pmc_instmix.json
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "cfs_solve_fused_kernel"
GROUPS = 7


def find(d, suffix):
    g = glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*" + suffix), recursive=True)
    return max(g, key=os.path.getmtime) if g else None      # gpurun merges into gpurun_out/: older passes may still lie there


def counters(d):
    """{counter: mean per launch of the fused kernel} of one pass"""
    f = find(d, "counter_collection.csv")
    if not f:
        return {}, 0
    acc = {}
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {k: sum(v.values()) / len(v) for k, v in acc.items()}
    n = max((len(v) for v in acc.values()), default=0)
    return out, n


def counters_of(d, kernel):
    """{counter: (mean per launch, launches)} of kernel `kernel` in one pass"""
    f = find(d, "counter_collection.csv")
    if not f:
        return {}
    acc = {}
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in acc.items()}


def main(tag):
    pmc = {"round": tag, "kernel": KERNEL,
           "command": "rocprofv3 --pmc <group> --output-format csv -- python3 bench.py --mode <MODE> --steps 5 --warmup 2 --blocks 1 --streams 1 "
                      "--no-cpu-baseline --no-other-mode (tools/profile_round.sh: one pass per group, no trace domains; --streams 1 = serial launches)",
           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of "
                         "wide coalesced reads -> read bytes = 2*FETCH_SIZE*1024 (upper estimate: most loads of this kernel are 8 B/lane); "
                         "WRITE_SIZE*1024 is exact"}
    mix = {"round": tag, "kernel": KERNEL + " (tier w2s, identity-Hessian instantiation, for PSGCFS; w2m for CFS)", "command": pmc["command"],
           "units": "means per launch of the fused kernel; SQ_INSTS_* are wave-level instructions summed over the chip; SQ_WAVE_CYCLES / "
                    "SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles", "per_launch": {}}
    for mode in ("PSGCFS", "CFS"):
        st = find("prof_stats_" + mode, "kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(ROOT, "profiles", f"{tag}_bench_{mode.lower()}_kernel_stats.csv"))
            for r in csv.DictReader(open(st)):
                if KERNEL in r["Name"]:
                    pmc[f"kernel_avg_ns_{mode}_default_streams"] = float(r["AverageNs"])
                    pmc[f"kernel_calls_{mode}"] = int(r["Calls"])
        allc = {}
        for k in range(1, GROUPS + 1):
            c, n = counters(f"prof_pmc{k}_{mode}")
            allc.update(c)
            if k == 1:
                pmc[f"launches_{mode}"] = n
        if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
            fk, wk = allc.pop("FETCH_SIZE"), allc.pop("WRITE_SIZE")
            pmc[f"FETCH_SIZE_KB_per_launch_mean_{mode}"] = fk
            pmc[f"WRITE_SIZE_KB_per_launch_mean_{mode}"] = wk
            pmc[f"hbm_bytes_per_launch_{mode}"] = 2 * fk * 1024 + wk * 1024
            pmc[f"hbm_bytes_per_launch_{mode}_uncorrected"] = (fk + wk) * 1024
        if allc:
            mix["per_launch"][mode] = allc
            g = lambda k: allc.get(k, 0.0)  # noqa: E731
            flop = 64 * (2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64"))
            pmc[f"fp64_valu_flops_per_launch_{mode}"] = flop
            d = {"fp64_flop_per_launch_upper (64 lanes per wave instruction)": flop}
            f64 = g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64")
            if g("SQ_INSTS_VALU"):
                d["fp64_share_of_valu_instructions"] = f64 / g("SQ_INSTS_VALU")
            if g("SQ_WAVE_CYCLES"):
                d["wave_time_waiting (SQ_WAIT_ANY/SQ_WAVE_CYCLES)"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
                d["wave_time_issuing (SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES)"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
                d["wave_time_issue_stalled (SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES)"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
            if g("SQ_LDS_IDX_ACTIVE"):
                d["lds_bank_conflict_share (SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE)"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
            if g("TCC_HIT_sum") + g("TCC_MISS_sum"):
                d["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
            mix["derived_" + mode] = d
    for nm, dst in (("prof_stats_rrt", f"{tag}_rrt_bench_kernel_stats.csv"), ("prof_stats_mesh", f"{tag}_mesh_config5_reference_map_kernel_stats.csv")):
        st = find(nm, "kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(ROOT, "profiles", dst))
    # the one MFMA site of the path: cfs_batched_gemv_kernel (-H^-1 ff before a CFS solve, QQ * logged u after it)
    gm = counters_of("prof_pmc3_CFS", "cfs_batched_gemv_kernel")
    if gm:
        mops, n = gm.get("SQ_INSTS_VALU_MFMA_MOPS_F64", (0.0, 0))
        fma, _ = gm.get("SQ_INSTS_VALU_FMA_F64", (0.0, 0))
        mix["mfma_site"] = {"kernel": "cfs_batched_gemv_kernel (v_mfma_f64_16x16x4_f64)", "launches_counted": n,
                            "SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch": mops, "SQ_INSTS_VALU_FMA_F64_per_launch": fma,
                            "SQ_INSTS_VALU_MFMA_MOPS_F64_of_the_fused_kernel": mix["per_launch"].get("CFS", {}).get("SQ_INSTS_VALU_MFMA_MOPS_F64"),
                            "note": "MFMA is nominal on this path: the two batched products (nn x nn times nn x B) are the only GEMM-shaped work; "
                                    "constraint rows are structured and never form a dense matrix (DESIGN.md section 4.2)"}
    json.dump(pmc, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    shutil.copy(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), os.path.join(ROOT, "profiles", "pmc_latest.json"))
    json.dump(mix, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_instmix.json"), "w"), indent=1)
    print(json.dumps(pmc, indent=1))
    print(json.dumps({k: v for k, v in mix.items() if k.startswith("derived")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
