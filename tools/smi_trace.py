#!/usr/bin/env python3
"""Side process: samples `rocm-smi -c -P -t --csv` every `period` seconds into a CSV until the file <out>.stop appears (or
`max_s` elapse).  Never touches the GPU through HIP (it only runs rocm-smi), so it can be started before the measured
command and outlive it.  usage: smi_trace.py <out.csv> [period_s=1.0] [max_s=1200]"""
import csv
import io
import os
import subprocess
import sys
import time

out = sys.argv[1]
period = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
max_s = float(sys.argv[3]) if len(sys.argv) > 3 else 1200.0
t0 = time.time()
header = None
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    while time.time() - t0 < max_s and not os.path.exists(out + ".stop"):
        ts = time.time() - t0
        try:
            txt = subprocess.run(["rocm-smi", "-c", "-P", "-t", "--csv"], capture_output=True, text=True, timeout=10).stdout
            rows = list(csv.reader(io.StringIO(txt.strip())))
            rows = [r for r in rows if r]
            if len(rows) >= 2:
                if header is None:
                    header = ["t_s"] + rows[0]
                    w.writerow(header)
                w.writerow([f"{ts:.2f}"] + rows[1])
                f.flush()
        except Exception as e:  # noqa: BLE001
            w.writerow([f"{ts:.2f}", "error", str(e)])
        time.sleep(max(0.0, period - ((time.time() - t0) - ts)))
