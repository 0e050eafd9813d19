"""Reduce the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/gpu_round.sh) to HBM bytes per launch for the KG kernels.

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB (rocprofv3 derived counters); on gfx950 FETCH_SIZE reports 1/2 of
the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md section HBM), so the read side is doubled -- an upper
estimate for kernels whose reads are not wide streams."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            key = "kg_mc_lane_kernel" if "kg_mc_lane_kernel" in name else "kg_mc_block_kernel" if "kg_mc_block_kernel" in name else "kg_mc_stream_kernel" if "kg_mc_stream_kernel" in name else "kg_mc_kernel" if "kg_mc_kernel" in name else (
                "cov_build_kernel" if ("cov_build_kernel" in name or "cov_build_value_kernel" in name) else None)
            if key == "cov_build_kernel" and int(row["Grid_Size"]) < 1_000_000:
                continue  # only the N x (E M) gradient-tail build, not the small state builds
            if key:
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and "End_Timestamp" in row:
                    acc[key]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    fetch = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
    write = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
    out[k] = {"fetch_size_raw_kb": fetch, "write_size_raw_kb": write,
              "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "launches": len(c.get("FETCH_SIZE", []))}
    # FP64 wave-instruction counters of the same kernel, when a pass collected them (bench.py: roofline.executed_frac):
    # executed flop per launch = 64 lanes x (2 FMA + ADD + MUL)
    for ctr in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU"):
        if ctr in c:
            out[k][ctr] = sum(c[ctr]) / len(c[ctr])
    # r4: effective clock and VALU busy of the kernel (bench.py roofline.in_kernel).  GRBM_GUI_ACTIVE comes back summed over the 8
    # XCDs of the chip (18e9 "cycles" per second otherwise); SQ_* are sums over the SIMDs in quad-cycles (MI355X_MICROARCH.md)
    NUM_XCD, NUM_CU = 8, 256
    mean = lambda v: sum(v) / len(v)
    if c.get("GRBM_GUI_ACTIVE") and c.get("_dur_ns"):
        # the LAST launch of each pass (the first carries first-touch allocation stalls)
        gui, dur = c["GRBM_GUI_ACTIVE"][-1] / NUM_XCD, c["_dur_ns"][-1]
        out[k]["effective_clock_ghz"] = gui / dur
        out[k]["kernel_ms_under_pmc"] = dur / 1e6
        if c.get("SQ_ACTIVE_INST_VALU"):
            out[k]["valu_busy"] = c["SQ_ACTIVE_INST_VALU"][-1] / NUM_CU / gui
        if c.get("SQ_INST_CYCLES_SALU"):
            out[k]["salu_busy"] = c["SQ_INST_CYCLES_SALU"][-1] / NUM_CU / gui
        if c.get("SQ_INSTS_VALU"):
            out[k]["valu_insts_per_launch"] = mean(c["SQ_INSTS_VALU"])
            if c.get("SQ_ACTIVE_INST_VALU"):
                out[k]["valu_quad_cycles_per_inst"] = c["SQ_ACTIVE_INST_VALU"][-1] / c["SQ_INSTS_VALU"][-1]
        if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES"):
            out[k]["wait_inst_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"][-1] / c["SQ_WAVE_CYCLES"][-1]
    if "SQ_INSTS_VALU_FMA_F64" in out[k]:
        out[k]["executed_fp64_flop_per_launch"] = 64.0 * (2.0 * out[k]["SQ_INSTS_VALU_FMA_F64"] + out[k].get("SQ_INSTS_VALU_ADD_F64", 0.0)
                                                          + out[k].get("SQ_INSTS_VALU_MUL_F64", 0.0))
print(json.dumps(out))
