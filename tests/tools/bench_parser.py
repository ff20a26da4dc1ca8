"""Detection ingest at full size: a synthetic 1 M-frame x 8-camera recording in the reference's
on-disk format (float-formatted fields, missing frame numbers, empty frames), the parallel parser
(host/DetectionsParser.h) against a reference-style sequential reader, results compared field by field.
usage: python tests/tools/bench_parser.py [n_frames] [n_cam]  -> one JSON line per thread count (CPU only)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "3d-reconstruction-triangulation_amd", "host_io_check")
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_cam = int(sys.argv[2]) if len(sys.argv) > 2 else 8
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    subprocess.run([EXE, "gencsv", d, str(n_cam), str(n_frames), "7"], check=True, capture_output=True)
    for threads in (1, 2, 4, 8, 16):
        if threads > (os.cpu_count() or 1):
            break
        r = json.loads(subprocess.run([EXE, "parsebench", d, str(threads), "3"], check=True, capture_output=True,
                                      text=True).stdout)
        r["host_cpus"] = os.cpu_count()
        print(json.dumps(r), flush=True)
        assert r["equal"]
