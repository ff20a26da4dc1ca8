"""How much of the spread of the drop-in call's time across boxes is NUMA placement?  Finds the NUMA node of the
visible GPU (PCI bus id -> sysfs), then runs tools/host_path_rate.py (C3 shape, default pipeline mode) in child
processes whose CPU affinity is (a) unrestricted, (b) the GPU's node, (c) the other node.
usage: python tools/host_numa_ab.py [n_cam] [n_sys]"""
import glob, json, os, subprocess, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_cam = sys.argv[1] if len(sys.argv) > 1 else "8"
n_sys = sys.argv[2] if len(sys.argv) > 2 else "10000000"
p = torch.cuda.get_device_properties(0)
bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())


def cpus(path):
    out = set()
    for part in open(path).read().strip().split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


nodes = {int(os.path.basename(d)[4:]): cpus(d + "/cpulist") for d in glob.glob("/sys/devices/system/node/node[0-9]*")}
allowed = os.sched_getaffinity(0)
print(json.dumps({"gpu": bdf, "gpu_numa_node": node, "nodes": {k: len(v) for k, v in nodes.items()},
                  "allowed_cpus": len(allowed)}), flush=True)
cases = [("unrestricted", allowed), ("gpu_node", nodes[node] & allowed)]
cases += [(f"node{k}", v & allowed) for k, v in sorted(nodes.items()) if k != node and v & allowed]
for label, mask in cases:
    for mode in ("3", "2"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_path_rate.py"), n_cam, n_sys, mode, "0"],
                           capture_output=True, text=True, timeout=280,
                           preexec_fn=lambda m=mask: os.sched_setaffinity(0, m))
        line = [l for l in r.stdout.splitlines() if l.startswith("{'mode'")]
        print(label, "mode", mode, line[-1] if line else r.stderr[-400:], flush=True)
