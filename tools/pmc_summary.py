#!/usr/bin/env python3
"""Summarise tools/pmc.sh output into one text file (per-kernel mean of every counter + derived figures) and, next to it, a .json
with the same numbers per kernel (bench.py reads the instruction counts and the HBM traffic of its roofline block from that file)."""
import collections, csv, glob, json, os, sys

def csrc_sha256():
    """sha256 over the kernel sources this pass was taken from (bench.py compares it with the sources it runs: `pmc_stale`)."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "mitsuba3-camera_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def main(d, out):
    lines = []
    stats = glob.glob(os.path.join(d, "stats", "*", "*_kernel_stats.csv"))
    if stats:
        lines.append("# rocprofv3 --kernel-trace --stats (top kernels)")
        for i, r in enumerate(csv.DictReader(open(stats[0]))):
            if i < 6:
                lines.append("%-60s calls=%s avg_ms=%.4f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
    agg = {}
    by_grid = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    meta = {}; metas = {}
    out_json = {"source": os.path.basename(os.path.normpath(d)), "csrc_sha256": csrc_sha256(), "kernels": {}}
    if stats:
        for r in csv.DictReader(open(stats[0])):
            out_json["kernels"].setdefault(r["Name"].split("(")[0][-40:], {})["avg_ns"] = float(r["AverageNs"])
    for f in sorted(glob.glob(os.path.join(d, "pass*", "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            if not ("k_render" in k or "k_prb" in k or "k_splat" in k or "k_ptracer" in k or "k_pt_" in k):
                continue
            by_grid[k][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            metas[(k, int(r["Grid_Size"]))] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["SGPR_Count"])
    # A kernel is launched with several grid sizes (the lens leg renders at 256 spp and, inside an update, at 1 spp): the counters of the LARGEST
    # launches are the ones bench.py scales by lanes -- averaging over all of them mixed a 16 384-thread launch into a 4 194 304-thread one.
    for k, grids in by_grid.items():
        g = max(grids)
        agg[k] = grids[g]; meta[k] = metas[(k, g)]
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        lines.append("")
        lines.append("# kernel %s  grid=%s wg=%s lds=%s scratch=%s vgpr=%s sgpr=%s" % ((k,) + meta[k]))
        kj = out_json["kernels"].setdefault(k, {})
        kj.update({"grid": int(meta[k][0]), "workgroup": int(meta[k][1]), "lds": int(meta[k][2]), "scratch": int(meta[k][3]), "vgpr": int(meta[k][4]), "sgpr": int(meta[k][5])})
        kj["counters"] = {n: m[n] for n in sorted(m)}
        if "FETCH_SIZE" in m:       # KiB; gfx950 tallies wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM): x2 is the upper bound
            kj["hbm_read_bytes"] = [m["FETCH_SIZE"] * 1024, 2 * m["FETCH_SIZE"] * 1024]
        if "WRITE_SIZE" in m:
            kj["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        for n in sorted(m):
            lines.append("%-28s %.6g" % (n, m[n]))
        w = m.get("SQ_WAVES")
        if w:
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VALU_TRANS_F32"):
                if n in m:
                    lines.append("per-wave %-22s %.1f" % (n, m[n] / w))
        if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
            lines.append("VALU lane utilisation        %.3f" % (m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"])))
        if "FETCH_SIZE" in m:   # KiB units; gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2 upper bound
            lines.append("HBM read bytes (FETCH_SIZE*1024, x2 corrected) %.4g .. %.4g" % (m["FETCH_SIZE"] * 1024, 2 * m["FETCH_SIZE"] * 1024))
        if "WRITE_SIZE" in m:
            lines.append("HBM write bytes (WRITE_SIZE*1024)              %.4g" % (m["WRITE_SIZE"] * 1024))
        if "GRBM_GUI_ACTIVE" in m:
            lines.append("GRBM_GUI_ACTIVE/8 (cycles)   %.4g" % (m["GRBM_GUI_ACTIVE"] / 8))
    open(out, "w").write("\n".join(lines) + "\n")
    json.dump(out_json, open(os.path.splitext(out)[0] + ".json", "w"), indent=1)
    print("\n".join(lines))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
