"""Turn the rocprofv3 output of scripts/collect_profiles.sh (under gpurun_out/) into the small tracked files
under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc_{FETCH,WRITE}_SIZE_summary.csv, <tag>_spmv_traffic.json.
usage: python scripts/summarize_profiles.py r01"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def newest(pattern):
    files = glob.glob(os.path.join(OUT, pattern))
    return max(files, key=os.path.getmtime) if files else None


def pmc_summary(counter):
    f = newest("pmc_%s/*/*counter_collection.csv" % counter)
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    rows = sorted(((k, v[0], v[1] / v[0]) for k, v in acc.items()), key=lambda t: -t[1] * t[2])
    with open(os.path.join(PROF, "%s_pmc_%s_summary.csv" % (tag, counter)), "w") as g:
        g.write("kernel,launches,avg_%s_KB\n" % counter)
        for k, n, avg in rows[:25]:
            g.write('"%s",%d,%.1f\n' % (k, n, avg))
    return {k: (n, avg) for k, n, avg in rows}


stats = newest("prof_stats/*/*kernel_stats.csv")
shutil.copy(stats, os.path.join(PROF, "%s_kernel_stats_bjacobi_ilu0.csv" % tag))
fetch, write = pmc_summary("FETCH_SIZE"), pmc_summary("WRITE_SIZE")


def traffic(prefix):
    nf = [(n, a) for k, (n, a) in fetch.items() if k.startswith(prefix)]
    nw = [(n, a) for k, (n, a) in write.items() if k.startswith(prefix)]
    f = sum(n * a for n, a in nf) / sum(n for n, a in nf)
    w = sum(n * a for n, a in nw) / sum(n for n, a in nw)
    return f, w, sum(n for n, a in nf)


spmv_prefix = "void isph::k_sell_spmv16" if any(k.startswith("void isph::k_sell_spmv16") for k in fetch) else "void isph::k_sell_spmv"
sf, sw, sn = traffic(spmv_prefix)
jf, jw, jn = traffic("void isph::k_ilu_solve_stream")
old = json.load(open(os.path.join(PROF, "%s_spmv_traffic.json" % tag))) if os.path.exists(os.path.join(PROF, "%s_spmv_traffic.json" % tag)) else {}
doc = {
    "kernel": spmv_prefix.replace("void isph::", "") + "<8,*>",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (scripts/collect_profiles.sh)",
    "workload": old.get("workload", "3D TGV 100^3 advect"),
    "fetch_size_kb_raw": sf, "write_size_kb": sw,
    "fetch_correction": "x2 (gfx950: FETCH_SIZE reports 1/2 of a wide coalesced streaming read)",
    "traffic_bytes_per_launch": (2 * sf + sw) * 1024, "launches": sn,
    "nrow": old.get("nrow", 1000000), "nnz": old.get("nnz", 103845090),
    "ilu_solve_stream_traffic_bytes_per_launch": (2 * jf + jw) * 1024, "ilu_solve_stream_launches": jn,
}
json.dump(doc, open(os.path.join(PROF, "%s_spmv_traffic.json" % tag), "w"), indent=1)
print(json.dumps(doc, indent=1))
