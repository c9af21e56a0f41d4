"""Per-kernel-family hardware counters on MI355X (north_star: "rocprof HBM GB/s and MFMA-busy counters per kernel family").

    python tools/prof_counters.py --out profiles/r02_kernel_counters.json [--only 1,2,20] [--passes mfma,lds,fetch,write]

Runs tools/prof_targets.py under `rocprofv3 --pmc <set> --kernel-trace -f csv` once per counter set (separate passes, as the
MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides --kernel-trace), attributes every
dispatch to the marker-delimited group it belongs to and writes, per group and kernel: launches, average duration, MFMA-busy
fraction, LDS bank-conflict rate, wait breakdown, HBM read / write bytes and the achieved TFLOP/s / GB/s against the peaks.
gfx950 corrections (MI355X_MICROARCH.md): FETCH_SIZE reports half the bytes of wide coalesced reads (doubled here; the raw value
is kept), counters in KB.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs (32 per 32x32x16 bf16 MFMA per
SIMD); MFMA-busy = that / (1024 x kernel duration x effective clock), with the clock taken from GRBM_GUI_ACTIVE / duration."""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PASSES = {
    "mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
            "SQ_WAVE_CYCLES"],
    # instruction mix: dynamic instruction counts per class and the cycles the vector ALU / LDS / scalar issue was busy (per wave-cycle)
    "valu": ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAVES",
             "SQ_WAVE_CYCLES"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
}
NSIMD = 1024
PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def short(name):
    m = re.match(r"_Z(\d+)", name)
    if m:                                              # mangled (llvm-cxxfilt does not know the __bf16 mangling DF16b): name + raw template tail
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        tail = name[m.end() + n:]
        targs = re.match(r"I(.*?)E(v|Pv|i)", tail)
        return base + ("<" + targs.group(1).replace("DF16b", "bf16,").replace("Lb0E", "0,").replace("Lb1E", "1,").replace("Li", "").rstrip(",") + ">"
                       if targs else "")
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


def build_marker(hipcc=None):
    """tools/libprofmarker.so (marker kernel whose grid size tags the next group of launches), compiled HERE, in the un-profiled
    driver process and with the profiler's preload variables stripped - never from tools/prof_targets.py, which runs under rocprofv3
    with the GPU already initialised."""
    src, lib = os.path.join(HERE, "prof_marker.hip"), os.path.join(HERE, "libprofmarker.so")
    if os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(src):
        return lib
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",) and not k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "HSA_TOOLS"))}
    subprocess.check_call([hipcc or os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", lib, src], env=env)
    return lib


def run_pass(pname, counters, outdir, env):
    d = os.path.join(outdir, pname)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "-f", "csv", "-d", d, "-o", "r", "--",
           sys.executable, os.path.join(HERE, "prof_targets.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
    tags = None
    for line in r.stdout.splitlines():
        if line.startswith("PROF_TAGS "):
            tags = json.loads(line[len("PROF_TAGS "):])
    if r.returncode != 0 or tags is None:
        sys.stderr.write("pass %s failed (rc %d)\n%s\n%s\n" % (pname, r.returncode, r.stdout[-1500:], r.stderr[-3000:]))
        return None, None
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    return rows, tags


def aggregate(rows):
    """-> {(tag, kernel): {"n": launches, "dur_ns": total, counter: total}} in dispatch order, markers delimit the groups."""
    by_disp = {}
    for r in rows:
        did = int(r["Dispatch_Id"])
        e = by_disp.setdefault(did, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                     "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out, tag = {}, None
    for did in sorted(by_disp):
        e = by_disp[did]
        if "prof_marker_kernel" in e["name"]:
            tag = e["grid"] // 64
            if tag == 0x3fff:
                tag = None
            continue
        if tag is None:
            continue
        k = (tag, short(e["name"]))
        a = out.setdefault(k, {"n": 0, "dur_ns": 0})
        a["n"] += 1
        a["dur_ns"] += e["dur"]
        for c, v in e["c"].items():
            a[c] = a.get(c, 0.0) + v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_kernel_counters.json"))
    ap.add_argument("--build-marker", action="store_true", help="only compile tools/libprofmarker.so (no GPU needed) and exit")
    ap.add_argument("--only", default="")
    ap.add_argument("--passes", default="mfma,lds,fetch,write")
    ap.add_argument("--scratch", default=os.path.join(ROOT, "gpurun_out", "pmc_r04"))
    args = ap.parse_args()
    build_marker()                                           # before any rocprofv3 pass starts the target
    if args.build_marker:
        return
    env = dict(os.environ, PROF_ONLY=args.only, TMPDIR="/tmp")
    merged, tags = {}, {}
    for p in args.passes.split(","):
        rows, t = run_pass(p, PASSES[p], args.scratch, env)
        if rows is None:
            continue
        tags = t
        for k, a in aggregate(rows).items():
            m = merged.setdefault(k, {})
            for c, v in a.items():
                if c in ("n", "dur_ns"):
                    m.setdefault(p + "." + c, v)
                else:
                    m[c] = v / a["n"]                       # per-launch average
            m["n"] = a["n"]
            m.setdefault("dur_list", []).append(a["dur_ns"] / a["n"])
    report = []
    for (tag, kern), m in sorted(merged.items()):
        label, flop, nbytes = tags.get(str(tag), ("?", 0, 0))
        dur = min(m.pop("dur_list"))                         # profiled passes run at slightly different clocks; keep the fastest
        e = {"tag": tag, "group": label, "kernel": kern, "launches": m["n"], "avg_us": round(dur / 1e3, 2)}
        if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            clk_cycles = m["GRBM_GUI_ACTIVE"] / 8.0
            dur_mfma = max(m.get("mfma.dur_ns", dur * m["n"]) / m["n"], 1)              # ns per launch in the pass that read the counters
            ghz = clk_cycles / dur_mfma
            # GRBM_GUI_ACTIVE keeps counting around a launch (command processor, cache flushes): for launches shorter than ~100 us the
            # clock derived from it came out at 3-15 GHz in round 3, and an MFMA-busy figure normalised by it is meaningless (verdict
            # r03, weak #10).  Those launches - and any whose derived clock is implausible - are normalised by the kernel-trace duration
            # at the NOMINAL 2.4 GHz instead: a lower bound of the pipe's busy fraction (the chip never clocks above nominal), flagged.
            if dur_mfma < 100e3 or ghz > 2.5:
                e["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (NSIMD * dur_mfma * 2.4), 4)
                e["mfma_busy_basis"] = "kernel-trace duration x nominal 2.4 GHz (lower bound; GRBM-derived clock %.1f GHz rejected)" % ghz
            else:
                e["eff_clock_ghz"] = round(ghz, 3)
                e["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (NSIMD * clk_cycles), 4)
                e["mfma_busy_basis"] = "GRBM_GUI_ACTIVE / 8 (active cycles of the launch)"
        if "SQ_LDS_IDX_ACTIVE" in m:
            e["lds_conflict_rate"] = round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 4)
            wc = max(m.get("SQ_WAVE_CYCLES", 0), 1)
            e["wait_any_frac"] = round(m["SQ_WAIT_ANY"] / wc, 4)
            e["wait_inst_any_frac"] = round(m["SQ_WAIT_INST_ANY"] / wc, 4)
            e["wait_inst_lds_frac"] = round(m["SQ_WAIT_INST_LDS"] / wc, 4)
            e["active_inst_frac"] = round(m["SQ_ACTIVE_INST_ANY"] / wc, 4)
        if "SQ_INSTS_VALU" in m:
            nw = max(m.get("SQ_WAVES", 0), 1)
            e["insts_per_wave"] = {"valu": round(m["SQ_INSTS_VALU"] / nw, 1), "mfma": round(m.get("SQ_INSTS_MFMA", 0) / nw, 1),
                                   "salu": round(m.get("SQ_INSTS_SALU", 0) / nw, 1), "lds": round(m.get("SQ_INSTS_LDS", 0) / nw, 1)}
            wc = max(m.get("SQ_WAVE_CYCLES", 0), 1)
            e["valu_active_frac"] = round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4)
            e["lds_active_frac"] = round(m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 4)
        if "FETCH_SIZE" in m:
            e["fetch_kb_raw"] = round(m["FETCH_SIZE"], 1)
            e["hbm_read_mb"] = round(2 * m["FETCH_SIZE"] * 1024 / 1e6, 2)
        if "WRITE_SIZE" in m:
            e["hbm_write_mb"] = round(m["WRITE_SIZE"] * 1024 / 1e6, 2)
        if "hbm_read_mb" in e and "hbm_write_mb" in e:
            e["hbm_gbs"] = round((e["hbm_read_mb"] + e["hbm_write_mb"]) * 1e6 / (dur * 1e-9) / 1e9, 1)
            e["hbm_frac_of_8tbs"] = round(e["hbm_gbs"] / PEAK_GBS, 4)
        # algorithmic figures only make sense for the group's main kernel (the longest one); filled below
        e["_flop"], e["_bytes"] = flop, nbytes
        report.append(e)
    # main kernel of each group = largest avg duration
    for tag in set(e["tag"] for e in report):
        grp = [e for e in report if e["tag"] == tag]
        main_k = max(grp, key=lambda e: e["avg_us"])
        for e in grp:
            flop, nbytes = e.pop("_flop"), e.pop("_bytes")
            if e is main_k:
                tot_us = sum(g["avg_us"] * g["launches"] for g in grp) / max(main_k["launches"], 1)
                e["group_total_us"] = round(tot_us, 2)
                if flop:
                    e["algorithmic_tflops"] = round(flop / (tot_us * 1e-6) / 1e12, 1)
                    e["frac_of_bf16_peak"] = round(e["algorithmic_tflops"] / PEAK_TF, 4)
                if nbytes:
                    e["algorithmic_mb"] = round(nbytes / 1e6, 1)
                    e["algorithmic_gbs"] = round(nbytes / (tot_us * 1e-6) / 1e9, 1)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"command": "python tools/prof_counters.py (rocprofv3 --pmc <set> --kernel-trace, one pass per set, target tools/prof_targets.py)",
               "passes": {p: PASSES[p] for p in args.passes.split(",")}, "peaks": {"bf16_tflops": PEAK_TF, "hbm_gbs": PEAK_GBS},
               "kernels": report}, open(args.out, "w"), indent=1)
    for e in report:
        print(json.dumps(e))


if __name__ == "__main__":
    main()
