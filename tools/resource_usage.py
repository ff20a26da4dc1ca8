"""Per-kernel resource usage of the built library, from the compiler's remarks
(3d-reconstruction-triangulation_amd/build/resource_usage.txt, written by `make libtdr_amd.so` / `make asm`).
usage: python tools/resource_usage.py [substring ...]      (default: the LM kernels and the C3 matrix kernel)
Importable: parse(path) -> {demangled kernel name: {"vgprs", "agprs", "sgpr_spill", "scratch", "occupancy", "lds"}}"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "3d-reconstruction-triangulation_amd", "build", "resource_usage.txt")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.splitlines()
        return [o.replace("tdr_amd::(anonymous namespace)::", "") for o in out]
    except (OSError, subprocess.CalledProcessError):
        return names


def parse(path=DEFAULT):
    txt = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names, recs = [], []
    for b in blocks:
        name = b.split()[0]

        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", b)
            return int(m.group(1)) if m else None
        names.append(name)
        recs.append({"vgprs": g("VGPRs"), "agprs": g("AGPRs"), "sgpr_spill": g("SGPRs Spill"), "vgpr_spill": g("VGPRs Spill"),
                     "scratch": g("ScratchSize [bytes/lane]"), "occupancy": g("Occupancy [waves/SIMD]"),
                     "lds": g("LDS Size [bytes/block]")})
    out = {}
    for n, r in zip(demangle(names), recs):
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)  # drop the argument list
        out[n] = r
    return out


if __name__ == "__main__":
    pats = sys.argv[1:] or ["ray_kernel<", "ray_coop_kernel<", "matrix_kernel<8, false, 0, true, true, 64", "matrix_i16xk"]
    for name, r in parse().items():
        if any(p in name for p in pats):
            print(f"{name:95s} V {r['vgprs']:3d} A {r['agprs'] or 0:3d} scratch {r['scratch']:4d} occ {r['occupancy']} sgpr-spill {r['sgpr_spill']}")
