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
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_grbm", "pmc_sq3"):
    for k_, v_ in counters(sub, "selfplay").items():
        pmc.setdefault(k_, v_)          # (SQ_INSTS_VALU is collected in two passes: the first one counts)
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
# ---- the issue side ("VALU utilisation / occupancy alongside the HBM fraction").  The kernel is bound by instruction issue, so
# the roofline that says something about it is the VALU pipe's: how many of the SIMDs' cycles the kernel's vector instructions NEED on this
# hardware, against the cycles the launch took.
#   hw_frac = sum over instruction classes (count x pipe cycles per wave64 instruction on a SIMD-32) / (1024 SIMDs x cycles per launch)
# Counts: the PMC class counters of this run (pmc_sq3: SQ_INSTS_VALU_INT64 / _ADD_F64 / _MUL_F64 / _FMA_F64 / _CVT / _TRANS_F64) and, for the
# one priced class the counters do not separate (quarter-rate 32-bit multiplies), the static opcode mix of the kernel's move loop
# (profiles/<name>_isa_mix.json, tools/isa_stats.py --mix).  Costs: MI355X_MICROARCH.md (a wave64 VALU instruction issues over 2 cycles on a
# SIMD-32; fp64 vector peak = half the f32 peak) and profiles/round3_pattern_cost.txt (tools/pattern_cost.hip on this GPU: what a pattern
# costs once two waves share the pipe).
COSTS = {
    "default": (2.0, "MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles (32 lanes / cycle x 2)"),
    "int64": (8.0, "quarter rate; measured 7.6: `v_lshrrev_b64; v_or` 19.22 cycles per link with two waves per SIMD = 2 x (7.6 + 2) (round3_pattern_cost.txt)"),
    "f64": (4.0, "half rate (fp64 vector peak 78.6 TFLOP/s = half of 157.3); measured <= 4.3: v_add_f64 8.56 per link at two waves per SIMD"),
    "cvt": (4.7, "measured: `v_cvt_f64_u32; v_cvt_u32_f64` 18.71 cycles per link at two waves per SIMD = 2 x 2 x 4.7 (round3_pattern_cost.txt)"),
    "trans_f64": (8.0, "v_rcp_f64 (the statistics' percentage at an episode end): transcendental rate"),
    "mul32": (8.0, "v_mul_hi / v_mul_lo _u32: quarter rate; measured 7.5: `v_mul_hi_u32; v_or` 19.04 per link at two waves per SIMD"),
}
N_SIMD, N_CU = 1024, 256
if "SQ_INSTS_VALU" in pmc and "SQ_INSTS_SALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    valu, salu = pmc["SQ_INSTS_VALU"]["mean"], pmc["SQ_INSTS_SALU"]["mean"]
    wave_instr = valu + salu
    cycles = pmc["GRBM_GUI_ACTIVE"]["mean"] / 8.0                     # summed over the 8 XCDs
    issue = {"instr_per_game_move": wave_instr / waves_steps, "valu_per_game_move": valu / waves_steps,
             "salu_per_game_move": salu / waves_steps, "wave_instr_per_launch": wave_instr,
             "cycles_per_launch": cycles, "clock_ghz": cycles / float(sp["AverageNs"]),
             "waves_per_simd": pmc["SQ_WAVES"]["mean"] / N_SIMD if "SQ_WAVES" in pmc else None,
             "cycles_per_instr_per_simd": cycles * N_SIMD / wave_instr,
             "source": "profiles/%s_summary.json (rocprofv3 --pmc passes of `bench.py --steps 20 --warmup 5`: SQ_INSTS_*, the class counters, "
                       "SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE in its own pass)" % name}
    g_ = lambda k: pmc[k]["mean"] if k in pmc else None
    if g_("SQ_INSTS_VALU_INT64") is not None:
        mixf = os.path.join(dst, "%s_isa_mix.json" % name)
        mul_share = json.load(open(mixf))["quarter_rate_int32_multiply_share_of_valu"] if os.path.exists(mixf) else 0.0
        cls = {"int64": g_("SQ_INSTS_VALU_INT64"),
               "f64": (g_("SQ_INSTS_VALU_ADD_F64") or 0.0) + (g_("SQ_INSTS_VALU_MUL_F64") or 0.0) + (g_("SQ_INSTS_VALU_FMA_F64") or 0.0),
               "cvt": g_("SQ_INSTS_VALU_CVT") or 0.0, "trans_f64": g_("SQ_INSTS_VALU_TRANS_F64") or 0.0, "mul32": valu * mul_share}
        cls["default"] = valu - sum(cls.values())
        pipe = {k: cls[k] * COSTS[k][0] for k in cls}
        need = sum(pipe.values())
        issue["hw"] = {
            "frac": need / (N_SIMD * cycles),
            "frac_if_every_valu_instruction_were_full_rate": 2.0 * valu / (N_SIMD * cycles),
            "valu_pipe_cycles_needed_per_launch": need, "simd_cycles_per_launch": N_SIMD * cycles,
            "mean_pipe_cycles_per_valu_instruction": need / valu,
            "classes_per_game_move": {k: cls[k] / waves_steps for k in cls},
            "pipe_cycles_per_game_move": {k: pipe[k] / waves_steps for k in pipe},
            "int32_counter_per_game_move": (g_("SQ_INSTS_VALU_INT32") or 0.0) / waves_steps,
            "costs": {k: {"cycles": COSTS[k][0], "source": COSTS[k][1]} for k in COSTS},
            "mul32_share_source": "profiles/%s_isa_mix.json (static opcode mix of the move loop, tools/isa_stats.py --mix)" % name,
            # the scalar pipe is the other issue resource: one scalar ALU per CU serves its four SIMDs, ~1.08 cycles per instruction
            # (profiles/round2_issue_model.txt: 64 independent s_ instructions, eight waves per SIMD, 4.3 cycles per instruction per SIMD)
            "scalar_pipe_frac": salu * 1.08 / (N_CU * cycles),
            "definition": "VALU pipe cycles the kernel's instructions need on a SIMD-32 (class counts x cycles per wave64 instruction) / "
                          "(1024 SIMDs x GRBM_GUI_ACTIVE cycles of the launch): 1.0 = every SIMD issues a vector instruction whenever it can"}
        issue["hw_frac"] = issue["hw"]["frac"]
        # what this occupancy allows: a move is ONE dependent chain, and a dependent vector instruction costs 4.42 cycles per instruction and
        # SIMD with two waves per SIMD (profiles/round2_issue_model.txt, "valu x64 dependent", 2 w/SIMD; 2.92 for independent instructions)
        issue["hw"]["ceiling_of_a_dependent_chain_at_two_waves_per_simd"] = issue["hw"]["mean_pipe_cycles_per_valu_instruction"] / 4.423
        issue["hw"]["ceiling_source"] = ("profiles/round2_issue_model.txt (tools/issue_model.hip on this GPU): 64 DEPENDENT v_ instructions, two waves per SIMD: "
                                         "4.423 cycles per instruction per SIMD; this kernel: %.2f per VALU instruction" % (N_SIMD * cycles / valu))
    if "SQ_ACTIVE_INST_VALU" in pmc and "SQ_WAVE_CYCLES" in pmc:
        wc = pmc["SQ_WAVE_CYCLES"]["mean"]
        issue["wave_time_shares"] = {"active_valu": pmc["SQ_ACTIVE_INST_VALU"]["mean"] / wc,
                                     "active_scalar": pmc["SQ_ACTIVE_INST_SCA"]["mean"] / wc if "SQ_ACTIVE_INST_SCA" in pmc else None,
                                     "active_any": pmc["SQ_ACTIVE_INST_ANY"]["mean"] / wc if "SQ_ACTIVE_INST_ANY" in pmc else None,
                                     "wait_any": pmc["SQ_WAIT_ANY"]["mean"] / wc if "SQ_WAIT_ANY" in pmc else None,
                                     "wait_inst_any": pmc["SQ_WAIT_INST_ANY"]["mean"] / wc if "SQ_WAIT_INST_ANY" in pmc else None,
                                     "note": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES etc. (quad-cycles both): the share of a wave's resident time it "
                                             "spends on vector instructions, parked at s_waitcnt, or stalled at issue"}
        issue["active_inst_valu_quad_cycles_per_valu_instruction"] = pmc["SQ_ACTIVE_INST_VALU"]["mean"] / valu
        issue["wait_any_share_of_wave_cycles"] = issue["wave_time_shares"]["wait_any"]
    if "SQ_ACTIVE_INST_VALU" in pmc and "SQ_BUSY_CYCLES" in pmc:
        issue["active_inst_valu_per_busy_cycle"] = pmc["SQ_ACTIVE_INST_VALU"]["mean"] / pmc["SQ_BUSY_CYCLES"]["mean"]
        # the same pipe share from the hardware's own activity counter: a wave is "active in a VALU instruction" for one quad-cycle (4 cycles)
        # per instruction (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00), of which the SIMD-32 pipe is occupied for 2 -- so
        # SQ_ACTIVE_INST_VALU x 2 / (SIMDs x cycles) is the full-rate pipe share and must agree with hw.frac_if_every_valu_instruction_were_full_rate
        issue["valu_pipe_share_from_SQ_ACTIVE_INST_VALU"] = pmc["SQ_ACTIVE_INST_VALU"]["mean"] * 2.0 / (N_SIMD * cycles)
    sweep = os.path.join(dst, "%s_games_sweep.txt" % name)
    if os.path.exists(sweep):
        vals = {}
        for ln in open(sweep):
            if ln.startswith("games") and "G env steps/s" in ln:
                vals[int(ln.split()[1].rstrip(":"))] = float(ln.split()[2])
        if 4096 in vals and 8192 in vals:
            # same instructions per game-move, twice the waves per SIMD: cycles per instruction scale with 1 / throughput.  This is the
            # kernel against ITSELF at higher occupancy -- how much of the gap to hw_frac = 1 more waves would close -- not a hardware bound
            issue["occupancy_frac"] = vals[4096] / vals[8192]
            issue["occupancy_source"] = "profiles/%s_games_sweep.txt: %.3f G env steps/s at 4096 games, %.3f G on an 8192-game grid (oversubscribed 2x at the same two RESIDENT waves per SIMD: backfill and a shorter tail, not higher occupancy)" % (name, vals[4096], vals[8192])
    summary["issue"] = issue
    json.dump(issue, open(os.path.join(dst, "issue_rate.json"), "w"), indent=1)
summary["bench"] = {k: bench[k] for k in ("value", "ms_per_step", "roofline", "cpu_baseline") if k in bench}
json.dump(summary, open(os.path.join(dst, "%s_summary.json" % name), "w"), indent=1)
print(json.dumps(summary, indent=1))
