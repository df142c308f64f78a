#!/usr/bin/env python3
"""Summarise rocprofv3 output (gpurun_out/<round>/{stats,fetch,write,mfma}, produced by tools/collect_profile.sh) into
profiles/: kernel-time table, per-launch HBM traffic of the dominant kernels (FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is
doubled as the MI355X guide prescribes for gfx950 -- it tallies 128-B requests at 64 B) and the matrix-core utilisation per
kernel (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x the kernel's cycles, GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r01"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
os.makedirs("profiles", exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)   # a re-collected leg leaves two files: take the newest
    return f[-1] if f else None


stats = one("stats/*kernel_stats.csv") or one("stats/*/*kernel_stats.csv")
lines = []
if stats:
    rows = list(csv.DictReader(open(stats)))
    lines.append(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-inference   ({tag}; bf16, per-GPU batch 128)")
    lines.append(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:36]:
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
        lines.append(f"{name:70s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
    # the same kernel name covers launches of different shapes (the 128-channel weight gradient: layer 3 vs layers 5 / 6; the persistent
    # convolution <0>: forward and data gradients): average duration per (kernel, grid) from the kernel trace of the same run
    tr = one("stats/*kernel_trace.csv") or one("stats/*/*kernel_trace.csv")
    if tr:
        by = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(tr)):
            nm = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").split("(")[0]
            if not any(t in nm for t in ("wgrad_bf16_kernel", "conv_bf16_ws")):
                continue
            grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else str(r.get("Grid_Size", "?"))
            a = by[(nm, grid)]
            a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a[1] += 1
        lines.append("")
        lines.append("# by launch shape (kernel trace of the same run): kernel | grid (work-items) | calls | avg_us")
        for (nm, grid), (t, n) in sorted(by.items()):
            lines.append(f"{nm[:70]:70s} {grid:>16s} {n:6d} {t / n / 1e3:10.1f}")
    open(f"profiles/{tag}_kernel_stats.txt", "w").write("\n".join(lines) + "\n")

traffic = {}
for kind, pat, mult in (("fetch", "fetch/*counter_collection.csv", 2.0), ("write", "write/*counter_collection.csv", 1.0)):
    f = one(pat) or one(pat.replace("/", "/*/", 1))
    if not f:
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][0] += float(r["Counter_Value"]) * 1024.0 * mult
        agg[k][1] += 1
    for k, (tot, n) in agg.items():
        traffic.setdefault(k, {})[kind + "_bytes_per_launch"] = tot / n
        traffic[k]["launches_" + kind] = n
out = {}
for k, v in traffic.items():
    if "fetch_bytes_per_launch" in v and "write_bytes_per_launch" in v:
        v["hbm_bytes_per_launch"] = v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]
    key = k.replace("void ", "").replace(", ", ",")
    out[key] = v
json.dump(out, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
# every kernel that moves more than 1 MB per launch (round 3's table was cut at 12 rows and lost the dominant kernel)
top = [kv for kv in sorted(out.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0)) if kv[1].get("hbm_bytes_per_launch", 0) > 1e6]
with open(f"profiles/{tag}_hbm_traffic.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference  ({tag})\n")
    f.write("# bytes per launch; FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section)\n")
    for k, v in top:
        f.write(f"{k:70s} fetch {v.get('fetch_bytes_per_launch',0)/1e6:10.1f} MB  write {v.get('write_bytes_per_launch',0)/1e6:10.1f} MB  n={v.get('launches_fetch')}\n")
# ---- matrix-core utilisation ------------------------------------------------------------------------------------------
mf = one("mfma/*counter_collection.csv") or one("mfma/*/*counter_collection.csv")
if mf:
    agg = defaultdict(lambda: defaultdict(float))
    nl = defaultdict(int)
    # (the clock a kernel held = GRBM_GUI_ACTIVE / 8 XCDs / its duration in the same run's kernel trace)
    dur = {}
    mtr = one("mfma/*kernel_trace.csv") or one("mfma/*/*kernel_trace.csv")
    if mtr:
        for r in csv.DictReader(open(mtr)):
            dur[r.get("Dispatch_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for r in csv.DictReader(open(mf)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            nl[k] += 1
            if r.get("Dispatch_Id") in dur:
                agg[k]["__ns"] += dur[r.get("Dispatch_Id")]
                agg[k]["__gui"] += float(r["Counter_Value"])
    rows = []
    for k, c in agg.items():
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0 or not k:
            continue
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        rows.append((c["GRBM_GUI_ACTIVE"], k, nl[k], cyc / nl[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), c["SQ_WAIT_ANY"] / wc,
                     c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, (c["__gui"] / 8.0 / c["__ns"] * 1e3) if c.get("__ns") else 0.0))
    rows.sort(reverse=True)
    with open(f"profiles/{tag}_mfma_util.txt", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE "
                f"--output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference  ({tag})\n")
        f.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): fraction of the kernel's cycles in which a SIMD's matrix pipe is busy;\n")
        f.write("# wait_any / wait_inst / active = shares of SQ_WAVE_CYCLES (parked at s_waitcnt or a barrier / issue stall / issuing)\n")
        f.write("# MHz = GRBM_GUI_ACTIVE / 8 / the launch's duration in the same run's kernel trace: the shader clock the kernel held (under the counters)\n")
        f.write(f"{'kernel':66s} {'launches':>8s} {'cycles/launch':>14s} {'mfma_busy':>10s} {'wait_any':>9s} {'wait_inst':>10s} {'active':>7s} {'MHz':>6s}\n")
        for r in rows[:30]:
            f.write(f"{r[1][:66]:66s} {r[2]:8d} {r[3]:14.0f} {r[4]:10.3f} {r[5]:9.2f} {r[6]:10.2f} {r[7]:7.2f} {r[8]:6.0f}\n")
    print(open(f"profiles/{tag}_mfma_util.txt").read())
# ---- LDS conflicts and instruction mix (tools/collect_counters_extra.sh) ---------------------------------------------------------
def per_kernel(pattern):
    f = one(pattern) or one(pattern.replace("/", "/*/", 1))
    agg, nl = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    if f:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nl[k][r["Counter_Name"]] += 1
    return agg, nl


lds, lds_n = per_kernel("lds/*counter_collection.csv")
ins, ins_n = per_kernel("insts/*counter_collection.csv")
if lds or ins:
    keys = sorted(set(lds) | set(ins), key=lambda k: -ins.get(k, {}).get("SQ_INSTS_MFMA", 0.0) - ins.get(k, {}).get("SQ_INSTS_VALU", 0.0) * 1e-3)
    with open(f"profiles/{tag}_lds_insts.txt", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc LDSBankConflict LdsUtil | SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES (separate passes) "
                f"--output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference  ({tag})\n")
        f.write("# per launch (mean): LDS bank-conflict cycles in % of the LDS-active cycles, LDS utilisation in %, wavefront instructions by kind in thousands\n")
        f.write(f"{'kernel':62s} {'launches':>8s} {'bank_conf%':>10s} {'lds_util%':>10s} {'k_mfma':>9s} {'k_lds':>9s} {'k_valu':>9s} {'k_salu':>9s} {'k_vmem':>9s} {'lds/mfma':>9s}\n")
        for k in keys[:26]:
            n = max(ins_n.get(k, {}).get("SQ_INSTS_VALU", 0), lds_n.get(k, {}).get("LDSBankConflict", 0), 1)
            g = lambda d, dn, c: (d.get(k, {}).get(c, 0.0) / max(dn.get(k, {}).get(c, 0), 1))
            mf, ld = g(ins, ins_n, "SQ_INSTS_MFMA"), g(ins, ins_n, "SQ_INSTS_LDS")
            f.write(f"{k[:62]:62s} {n:8d} {g(lds, lds_n, 'LDSBankConflict'):10.2f} {g(lds, lds_n, 'LdsUtil'):10.2f} {mf/1e3:9.0f} {ld/1e3:9.0f} "
                    f"{g(ins, ins_n, 'SQ_INSTS_VALU')/1e3:9.0f} {g(ins, ins_n, 'SQ_INSTS_SALU')/1e3:9.0f} {g(ins, ins_n, 'SQ_INSTS_VMEM')/1e3:9.0f} {ld / mf if mf else 0:9.2f}\n")
    print(open(f"profiles/{tag}_lds_insts.txt").read())

act, act_n = per_kernel("active/*counter_collection.csv")
if act:
    rows = []
    for k, c in act.items():
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0 or not k:
            continue
        simd = cyc * 1024.0
        nl_ = max(act_n[k]["GRBM_GUI_ACTIVE"], 1)
        rows.append((c["GRBM_GUI_ACTIVE"], k, nl_, cyc / nl_, 4 * c["SQ_ACTIVE_INST_VALU"] / simd, 4 * c["SQ_ACTIVE_INST_LDS"] / simd,
                     4 * c["SQ_ACTIVE_INST_VMEM"] / simd, 4 * c["SQ_ACTIVE_INST_SCA"] / simd))
    rows.sort(reverse=True)
    with open(f"profiles/{tag}_issue_util.txt", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE "
                f"--output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-inference  ({tag})\n")
        f.write("# share of a SIMD's cycles with an instruction of that kind executing: SQ_ACTIVE_INST_* (counted in units of 4 cycles -- a wave64 vector\n")
        f.write("# instruction occupies the 16-lane SIMD for 4) x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).  valu near 1 = bound by vector-ALU issue;\n")
        f.write("# the scalar column uses the same normalisation (one scalar unit serves the CU's four SIMDs: read it as relative between kernels)\n")
        f.write(f"{'kernel':62s} {'launches':>8s} {'cycles/launch':>14s} {'valu':>6s} {'lds':>6s} {'vmem':>6s} {'scalar':>7s}\n")
        for r in rows[:26]:
            f.write(f"{r[1][:62]:62s} {r[2]:8d} {r[3]:14.0f} {r[4]:6.2f} {r[5]:6.2f} {r[6]:6.2f} {r[7]:7.2f}\n")
    print(open(f"profiles/{tag}_issue_util.txt").read())

# ---- inference path (tools/infer_profile.py) -----------------------------------------------------------------------------
def stats_table(pattern, header, n=24):
    f = one(pattern) or one(pattern.replace("/", "/*/", 1))
    if not f:
        return []
    out = [header, f"{'kernel':78s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
    for r in list(csv.DictReader(open(f)))[:n]:
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:78]
        out.append(f"{name:78s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
    return out


inf = []
for leg, args in (("infer_e2e", "e2e 5"), ("infer_post", "post 10")):
    t = stats_table(f"{leg}/*kernel_stats.csv",
                    f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/infer_profile.py {args}   ({tag}; batch 256, bf16 eval forward, 772x1032)")
    if t:
        log = os.path.join(src, leg + ".log")
        js = [ln.strip() for ln in open(log) if ln.startswith("{")] if os.path.exists(log) else []
        inf += t + (["# HIP-event times of the same run: " + js[-1]] if js else []) + [""]
if inf:
    open(f"profiles/{tag}_infer_kernel_stats.txt", "w").write("\n".join(inf))
    print("\n".join(inf))
itraffic = {}
for kind, pat, mult in (("fetch", "infer_fetch/*counter_collection.csv", 2.0), ("write", "infer_write/*counter_collection.csv", 1.0)):
    f = one(pat) or one(pat.replace("/", "/*/", 1))
    if not f:
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][0] += float(r["Counter_Value"]) * 1024.0 * mult
        agg[k][1] += 1
    for k, (tot, n) in agg.items():
        itraffic.setdefault(k, {})[kind] = tot / n
        itraffic[k]["n"] = n
if itraffic:
    with open(f"profiles/{tag}_infer_hbm_traffic.txt", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 tools/infer_profile.py post 3  ({tag})\n")
        f.write("# bytes per launch at batch 256; FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section); algorithmic: 154 MB read of the\n")
        f.write("# [256, 12, 97, 129] fp32 head output per pass (+ 154 MB written by the separate decode)\n")
        for k, v in sorted(itraffic.items(), key=lambda kv: -(kv[1].get("fetch", 0) + kv[1].get("write", 0)))[:10]:
            f.write(f"{k:70s} fetch {v.get('fetch',0)/1e6:10.1f} MB  write {v.get('write',0)/1e6:10.1f} MB  n={v.get('n')}\n")
    print(open(f"profiles/{tag}_infer_hbm_traffic.txt").read())
    json.dump(itraffic, open("profiles/traffic_infer.json", "w"), indent=1, sort_keys=True)
print(open(f"profiles/{tag}_kernel_stats.txt").read()[:3200])
print(open(f"profiles/{tag}_hbm_traffic.txt").read())
