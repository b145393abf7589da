"""Regenerates the TGV rows of reference_known_answers.json from the two text tables the
reference keeps (sph-script/conv-taylor-green-vortex-2d-rev{390,230}.txt).  Runs only where
/root/reference exists (the build container); the JSON it writes is data: expected outputs."""
import json
import os
import re

REF = "/root/reference/IMPLICIT-SPH/sph-script/"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse(fn):
    out, kernel, N = {}, None, None
    for line in open(fn).read().splitlines():
        line = line.strip()
        if line in ("Wendland", "Quintic"):
            kernel = line.lower()
            out[kernel] = {}
        m = re.match(r"N = (\d+)", line)
        if m:
            N = m.group(1)
        m = re.match(r"time step:(\d+), time:([\d.e+-]+)", line)
        if m:
            out[kernel][N] = {"step": int(m.group(1)), "time": float(m.group(2))}
        m = re.match(r"(pressure|velocity) l2 error \(norm\): ([\d.e+-]+) \(([\d.e+-]+)\)", line)
        if m:
            k = "p" if m.group(1) == "pressure" else "u"
            out[kernel][N][k + "_err"] = float(m.group(2))
            out[kernel][N][k + "_norm"] = float(m.group(3))
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "reference_known_answers.json")
    g = json.load(open(path))
    r390 = parse(REF + "conv-taylor-green-vortex-2d-rev390.txt")
    r230 = parse(REF + "conv-taylor-green-vortex-2d-rev230.txt")
    keep = lambda d: {k: v for k, v in d.items() if int(k) <= 128}      # N = 256, 512 are not run by the tests
    g["conv_taylor_green_vortex_2d_rev390"]["rows"] = keep(r390["wendland"])
    g["conv_taylor_green_vortex_2d_rev390_quintic"]["rows"] = keep(r390["quintic"])
    g["conv_taylor_green_vortex_2d_rev230"]["rows"] = keep(r230["wendland"])
    g["conv_taylor_green_vortex_2d_rev230_quintic"]["rows"] = keep(r230["quintic"])
    json.dump(g, open(path, "w"), indent=1)
