"""Per-kernel register / LDS / scratch table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; runs without a GPU).
    python tools/kernel_resources.py sar-ssl_amd/csrc/attention.hip [name-filter]"""
import os, re, subprocess, sys, tempfile

def main():
    src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sar-ssl_amd", "csrc"))
    import build
    with tempfile.TemporaryDirectory() as d:
        cmd = [build._hipcc()] + build.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "x.o")]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m: continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1); cur[k.strip()] = v.strip()
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
    except Exception:
        dem = [r["name"] for r in rows]
    print("%-92s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "LDS", "occ"))
    for r, n in zip(rows, dem):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if flt and flt not in n: continue
        print("%-92s %5s %5s %7s %6s %4s" % (n[:92], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                                            r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))

if __name__ == "__main__":
    main()
