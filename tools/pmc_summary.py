#!/usr/bin/env python3
"""Summarise rocprofv3 outputs for profiles/: kernel-stats by kernel family and the HBM traffic
of the conv kernels from the FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, as the
MI355X guide prescribes; gfx950 correction: FETCH_SIZE x2 for wide coalesced reads; unit KB).

usage: pmc_summary.py [--all] <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv> [sq counter_collection.csv [l2 counter_collection.csv]]

--all: the per-instantiation tables cover EVERY kernel that holds at least 0.3 % of the traced time (the training step's BatchNorm /
weight-gradient / optimizer kernels, the scoring kernels), not only the conv kernels.

The optional fourth file is an SQ pass (GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU / _LDS): per conv instantiation, matrix-core busy cycles
as a fraction of the kernel's GPU-active cycles x CUs x 4 SIMDs -> "mfma_util", and how the waves' cycles split between issuing,
waiting on memory / barriers and issue stalls.

The optional fifth file is an L2 pass (TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum, round 6): per instantiation the L2 hit rate
TCC_HIT / (TCC_HIT + TCC_MISS) (the guide's formula; atomics count as misses) and the L1 -> L2 read requests per launch -- with 64 bytes per
request (an assumption: the counter's unit is the request, not the byte) an estimate of the bytes per clock the kernel pulls from L2."""
import collections
import csv
import json
import sys


def demangle(n):
    """_Z22bn_apply_fwd_p2_kernelPKf... -> bn_apply_fwd_p2_kernel ; _Z23bn_bwd_apply2_p2_kernelILi3EEv... -> bn_bwd_apply2_p2_kernel<3> (rocprofv3 leaves
    some kernel names mangled; the Itanium prefix is all that is needed here)"""
    import re

    m = re.match(r"_Z(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1))
    base = n[m.end(): m.end() + ln]
    rest = n[m.end() + ln:]
    t = re.match(r"I((?:L[a-z]-?\d+E)+)E", rest)
    if t:
        base += "<" + ", ".join(re.findall(r"L[a-z](-?\d+)E", t.group(1))) + ">"
    return base


def fam(n):
    n = demangle(n)
    return ("conv_block_p2_kernel" if "conv_block_p2" in n else "conv_bneck_p2_kernel" if "conv_bneck_p2" in n else
            "conv_stem_p2_kernel" if "conv_stem_p2" in n else "conv_fuse_up_p2_kernel" if "conv_fuse_up_p2" in n else
            "conv_p2_kernel" if "conv_p2" in n else
            "conv_block_kernel" if "conv_block" in n else "conv_split_kernel" if "conv_split" in n else
            "conv_bf3_kernel" if "conv_bf3" in n else "conv_mfma_kernel" if "conv_mfma" in n else
            "conv_stem_kernel" if "conv_stem" in n else n.split("(")[0][:48])


def inst(n):
    """kernel name with its template arguments, without the parameter list"""
    n = demangle(n).split("(")[0]
    return n[5:] if n.startswith("void ") else n


def agg(path, counter, key=fam):
    d = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        g = d[key(r["Kernel_Name"])]
        g[0] += 1
        g[1] += float(r["Counter_Value"])
        g[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


argv = [a for a in sys.argv[1:] if a != "--all"]
ALL = "--all" in sys.argv[1:]
stats, fetch, write = argv[0:3]
sq = argv[3] if len(argv) > 3 else None
l2 = argv[4] if len(argv) > 4 else None
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
big = {inst(r["Name"]) for r in rows if float(r["TotalDurationNs"]) >= 0.003 * tot}


def wanted(n):
    return "conv_" in n or (ALL and n in big)


k = {}
for r in rows:
    a = k.setdefault(fam(r["Name"]), [0, 0.0])
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"])
out = {"kernel_stats": [dict(kernel=n, calls=c, avg_us=round(t / c / 1e3, 2), total_ms=round(t / 1e6, 3),
                             pct=round(100 * t / tot, 2)) for n, (c, t) in sorted(k.items(), key=lambda kv: -kv[1][1])[:24 if ALL else 10]]}
out["traced_kernel_ms_total"] = round(tot / 1e6, 3)
f, w = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
out["hbm_traffic_per_launch"] = []
for n in ("conv_p2_kernel", "conv_block_p2_kernel", "conv_bneck_p2_kernel", "conv_stem_p2_kernel", "conv_fuse_up_p2_kernel", "conv_split_kernel", "conv_block_kernel", "conv_bf3_kernel", "conv_mfma_kernel", "conv_stem_kernel"):
    if n in f and n in w:
        nf, fs, tf = f[n]
        nw, ws, _ = w[n]
        rd, wr = 2.0 * fs / nf * 1024, ws / nw * 1024
        out["hbm_traffic_per_launch"].append(dict(kernel=n, launches=nf, avg_us_under_pmc=round(tf / nf / 1e3, 2),
                                                  fetch_bytes_corrected=round(rd), write_bytes=round(wr),
                                                  total_bytes=round(rd + wr),
                                                  GBps=round((rd + wr) / (tf / nf), 1)))
# the same per template instantiation of the conv kernels (bench.py picks the 3x3 stride-1 ones)
fi, wi = agg(fetch, "FETCH_SIZE", inst), agg(write, "WRITE_SIZE", inst)
out["launch_counts"] = {n.split("<")[0]: 0 for n in fi}  # launches per kernel (template arguments folded) in the FETCH_SIZE pass
for n in fi:
    out["launch_counts"][n.split("<")[0]] += fi[n][0]
out["hbm_traffic_by_instantiation"] = []
for n in sorted(fi, key=lambda n: -fi[n][2]):
    if not wanted(n) or n not in wi:
        continue
    nf, fs, tf = fi[n]
    nw, ws, _ = wi[n]
    rd, wr = 2.0 * fs / nf * 1024, ws / nw * 1024
    out["hbm_traffic_by_instantiation"].append(dict(kernel=n, launches=nf, avg_us_under_pmc=round(tf / nf / 1e3, 2),
                                                    fetch_bytes_corrected=round(rd), write_bytes=round(wr),
                                                    total_bytes=round(rd + wr), GBps=round((rd + wr) / (tf / nf), 1)))
if sq:
    # SQ pass: one row per (dispatch, counter); sums over the XCDs / SEs are already folded by rocprofv3
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(sq)):
        k_ = inst(r["Kernel_Name"])
        if not wanted(k_):
            continue
        per[k_][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k_] += 1
            per[k_]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out["sq_by_instantiation"] = []
    for k_, c in sorted(per.items(), key=lambda kv: -kv[1]["_ns"]):
        n_ = max(cnt[k_], 1)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        wave = c.get("SQ_WAVE_CYCLES", 0.0)
        row = dict(kernel=k_, launches=n_, avg_us_under_pmc=round(c["_ns"] / n_ / 1e3, 2))
        if gui:
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (256 CUs x 4 can be busy at once);
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (937 786 "cycles" for a 52.6 us kernel = 8 x 2.23 GHz)
            row["mfma_util"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 256 * 4), 4)
        if wave:
            for name, key in (("issuing", "SQ_ACTIVE_INST_ANY"), ("waiting_mem_or_barrier", "SQ_WAIT_ANY"),
                              ("issue_stalled", "SQ_WAIT_INST_ANY"), ("valu", "SQ_ACTIVE_INST_VALU"), ("lds", "SQ_ACTIVE_INST_LDS")):
                if key in c:
                    row["wave_cycles_" + name] = round(c[key] / wave, 4)
        out["sq_by_instantiation"].append(row)
if l2:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(l2)):
        k_ = inst(r["Kernel_Name"])
        if not wanted(k_):
            continue
        per[k_][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum":
            cnt[k_] += 1
            per[k_]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out["l2_by_instantiation"] = []
    for k_, c in sorted(per.items(), key=lambda kv: -kv[1]["_ns"]):
        n_ = max(cnt[k_], 1)
        hit, miss, rd = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0), c.get("TCP_TCC_READ_REQ_sum", 0.0)
        us = c["_ns"] / n_ / 1e3
        row = dict(kernel=k_, launches=n_, avg_us_under_pmc=round(us, 2), l2_hit_rate=round(hit / (hit + miss), 4) if hit + miss else None,
                   tcc_hit_per_launch=round(hit / n_), tcc_miss_per_launch=round(miss / n_), tcp_tcc_read_req_per_launch=round(rd / n_))
        if rd and us:
            row["l2_read_GBps_at_64B_per_request"] = round(rd / n_ * 64.0 / (us * 1e3), 1)
        out["l2_by_instantiation"].append(row)
print(json.dumps(out, indent=1))
