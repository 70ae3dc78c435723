#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (written by tools/profile_gpu.sh) into profiles/<name>_*.{csv,json,md}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def counters(sub, kernel_filter):
    out = collections.defaultdict(list)
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        return {}
    for r in csv.DictReader(open(files[0])):
        if kernel_filter in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in out.items()}


stats_file = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copyfile(stats_file, os.path.join(dst, "%s_kernel_stats.csv" % name))
rows = list(csv.DictReader(open(stats_file)))
sp = [r for r in rows if "selfplay" in r["Name"]][0]
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
shutil.copyfile(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % name))

summary = {"tag": tag, "selfplay_kernel": {"calls": int(sp["Calls"]), "avg_ns": float(sp["AverageNs"]),
                                             "min_ns": float(sp["MinNs"]), "max_ns": float(sp["MaxNs"])}}
pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_grbm"):
    pmc.update(counters(sub, "selfplay"))
summary["pmc_per_launch"] = pmc
waves_steps_for_traffic = bench["config"]["games_per_gpu"] * bench["config"]["moves_per_launch"]
calib = {}
for sub, key in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if files:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            calib["%s:%s" % (k, c)] = sum(v) / len(v)
summary["calibration_raw_KB"] = calib
GiB = float(1 << 30)
fac = {}
if "calib_read_u32:FETCH_SIZE" in calib:
    fac["fetch_bytes_per_counted_KB_u32_reads"] = GiB / calib["calib_read_u32:FETCH_SIZE"]
if "calib_write_u8_rows:WRITE_SIZE" in calib:
    fac["write_bytes_per_counted_KB_u8_rows"] = (GiB // 180 * 180) / calib["calib_write_u8_rows:WRITE_SIZE"]
if "calib_write_u8_rows192:WRITE_SIZE" in calib:
    fac["write_bytes_per_counted_KB_u8_rows192"] = ((GiB // 192 // 2) * 2 * 192) / calib["calib_write_u8_rows192:WRITE_SIZE"]
if "calib_write_u64_rows192:WRITE_SIZE" in calib:
    fac["write_bytes_per_counted_KB_u64_rows192"] = ((GiB // 192 // 2) * 2 * 184) / calib["calib_write_u64_rows192:WRITE_SIZE"]
if "calib_write_u32:WRITE_SIZE" in calib:
    fac["write_bytes_per_counted_KB_u32"] = GiB / calib["calib_write_u32:WRITE_SIZE"]
summary["calibration_factors"] = fac
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    f = pmc["FETCH_SIZE"]["mean"] * fac.get("fetch_bytes_per_counted_KB_u32_reads", 1024.0)
    pitch = bench["config"].get("mask_row_pitch_bytes", 180)
    payload = pitch + 4 + 4 + 1 + 4 + (24 if bench["config"].get("mask_bits_stream", True) else 0)     # mask row, action, reward, done, compact record (+ mask bits)
    wkey = "write_bytes_per_counted_KB_u8_rows192" if pitch >= 192 and "write_bytes_per_counted_KB_u8_rows192" in fac else "write_bytes_per_counted_KB_u8_rows"
    if pitch >= 192 and "write_bytes_per_counted_KB_u64_rows192" in fac:      # round 3: one 8-byte store per lane, 184 bytes per padded row
        wkey = "write_bytes_per_counted_KB_u64_rows192"
        payload = 184 + 4 + 4 + 1 + 4 + (24 if bench["config"].get("mask_bits_stream", True) else 0)
    # WRITE_SIZE counts whole 32-byte sectors and is exact for stores (MI355X_MICROARCH.md, HBM / rocprofv3 section): 1024 bytes per
    # counted KB.  The calibration kernels confirm it -- `calib_write_u64_rows192` stores 184 payload bytes into every 192-byte row and
    # the counter reports the 192: payload / counter = 981.3 B per counted KB is the PAYLOAD fraction (184 / 192), not an HBM-byte
    # correction (round 3 multiplied by it and understated the write traffic by 4 %).
    w = pmc["WRITE_SIZE"]["mean"] * 1024.0
    summary["hbm_bytes_per_launch"] = {"fetch": f, "write": w, "total": f + w,
                                        "write_bytes_per_counted_KB": 1024.0,
                                        "payload_per_counted_KB_in_calibration": fac.get(wkey),
                                        "payload_bytes_written_per_move": payload,
                                        "write_amplification": w / waves_steps_for_traffic / payload,
                                        "note": "FETCH_SIZE x the read calibration of this path's access widths; WRITE_SIZE x 1024 (exact for stores)"}
    json.dump({"bytes_per_launch": f + w, "fetch": f, "write": w, "source": "profiles/%s_summary.json" % name,
               "games": bench["config"]["games_per_gpu"], "moves_per_launch": bench["config"]["moves_per_launch"],
               "bytes_per_move": (f + w) / waves_steps_for_traffic, "write_bytes_per_move": w / waves_steps_for_traffic,
               "payload_bytes_written_per_move": payload, "write_amplification": w / waves_steps_for_traffic / payload,
               "kernel": bench["roofline"].get("kernel", "?"),
               "launch": "%s, %d games x %d moves" % (bench["roofline"].get("kernel", "?"), bench["config"]["games_per_gpu"], bench["config"]["moves_per_launch"])}, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
waves_steps = bench["config"]["games_per_gpu"] * bench["config"]["moves_per_launch"]
if "SQ_INSTS_VALU" in pmc:
    summary["per_wave_step"] = {k: pmc[k]["mean"] / waves_steps for k in pmc if k.startswith("SQ_")}
# ---- the issue side ("VALU utilisation / occupancy alongside the HBM fraction"): the kernel is bound by the rate at which a
# SIMD issues this instruction mix, so the checkable figure is cycles per wave-instruction per SIMD against the rate the same kernel
# reaches when the SIMDs are saturated with waves (the 8192-game run of tools/games_sweep.sh: twice the waves, same instructions per move)
if "SQ_INSTS_VALU" in pmc and "SQ_INSTS_SALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    wave_instr = pmc["SQ_INSTS_VALU"]["mean"] + pmc["SQ_INSTS_SALU"]["mean"]
    cycles = pmc["GRBM_GUI_ACTIVE"]["mean"] / 8.0                     # summed over the 8 XCDs
    N_SIMD = 1024
    issue = {"instr_per_game_move": wave_instr / waves_steps, "valu_per_game_move": pmc["SQ_INSTS_VALU"]["mean"] / waves_steps,
             "salu_per_game_move": pmc["SQ_INSTS_SALU"]["mean"] / waves_steps, "wave_instr_per_launch": wave_instr,
             "cycles_per_launch": cycles, "clock_ghz": cycles / float(sp["AverageNs"]),
             "waves_per_simd": pmc["SQ_WAVES"]["mean"] / N_SIMD if "SQ_WAVES" in pmc else None,
             "cycles_per_instr_per_simd": cycles * N_SIMD / wave_instr,
             "wait_any_share_of_wave_cycles": (pmc["SQ_WAIT_ANY"]["mean"] / pmc["SQ_WAVE_CYCLES"]["mean"]) if "SQ_WAIT_ANY" in pmc and "SQ_WAVE_CYCLES" in pmc else None,
             "source": "profiles/%s_summary.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ..., GRBM_GUI_ACTIVE in its own pass)" % name}
    sweep = os.path.join(dst, "%s_games_sweep.txt" % name)
    if os.path.exists(sweep):
        vals = {}
        for ln in open(sweep):
            if ln.startswith("games") and "G env steps/s" in ln:
                vals[int(ln.split()[1].rstrip(":"))] = float(ln.split()[2])
        if 4096 in vals and 8192 in vals:
            # same instructions per game-move, twice the waves per SIMD: cycles per instruction scale with 1 / throughput
            issue["saturation_cycles_per_instr_per_simd"] = issue["cycles_per_instr_per_simd"] * vals[4096] / vals[8192]
            issue["saturation_source"] = "profiles/%s_games_sweep.txt: %.3f G env steps/s at 4096 games, %.3f G at 8192 (four waves per SIMD)" % (name, vals[4096], vals[8192])
            issue["frac"] = issue["saturation_cycles_per_instr_per_simd"] / issue["cycles_per_instr_per_simd"]
    summary["issue"] = issue
    json.dump(issue, open(os.path.join(dst, "issue_rate.json"), "w"), indent=1)
summary["bench"] = {k: bench[k] for k in ("value", "ms_per_step", "roofline", "cpu_baseline") if k in bench}
json.dump(summary, open(os.path.join(dst, "%s_summary.json" % name), "w"), indent=1)
print(json.dumps(summary, indent=1))
