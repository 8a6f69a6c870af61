#!/usr/bin/env python3
"""Calibration of the SQ VALU counters (measurement infrastructure): what do SQ_INSTS_VALU and SQ_ACTIVE_INST_VALU read for
kernels whose VALU occupancy is KNOWN?

    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d OUT -- python3 tools/valu_calib.py
    python3 tools/valu_calib.py --report OUT      # per microbenchmark: cycles the counter charges per instruction, busy fraction

tools/ubench.hip's k_valu<OP> is a loop of one VALU instruction, nothing else: with 8 waves per SIMD the pipe is saturated
(its measured issue rate is the chip's peak for that instruction class), with 1 wave per SIMD it is mostly idle.  A busy
formula that is right must read ~1.0 for the former whatever the instruction class (2-cycle VOP2 forms, 4-cycle VOP3
integer forms) and the known fraction for the latter; it is then applied to k_verify_a / k_scan_probe (bench.py,
roofline.valu_busy)."""
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

OPS = {1: "v_xor_b32 (VOP2, full rate)", 0: "v_alignbit_b32 (VOP3, half rate)", 8: "v_bfe_u32 (VOP3, half rate)", 24: "v_add_u32 (VOP2)"}


def run():
    import ubench
    lib = ubench.load()
    out = {}
    for op, name in OPS.items():
        for w in (8, 2, 1):
            r = lib.ub_valu(op, w, 40000)
            out["k_valu<%d> %d waves/SIMD" % (op, w)] = {"name": name, "wave_insts_per_sec_per_simd": r, "cycles_per_inst_at_2.4GHz": 2.4e9 / r}
    print(json.dumps(out, indent=1))


def report(d):
    rows = {}
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "k_valu" not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Dispatch_Id"]), int(r["Grid_Size"]))
                rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print("%-16s %9s %14s %16s %14s | %s" % ("kernel", "grid", "INSTS_VALU", "ACTIVE_INST_VALU", "BUSY_CU_CYCLES",
                                              "ACTIVE/INSTS  ACTIVE*4/(4*BUSY_CU)  INSTS*2/(4*BUSY_CU) [2 cycles per instruction]"))
    out = []
    for (k, disp, grid), c in sorted(rows.items(), key=lambda kv: kv[0][1]):
        iv, av, bc = c.get("SQ_INSTS_VALU", 0), c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_BUSY_CU_CYCLES", 0)
        if not iv or not bc:
            continue
        line = {"kernel": k, "grid": grid, "SQ_INSTS_VALU": iv, "SQ_ACTIVE_INST_VALU": av, "SQ_BUSY_CU_CYCLES": bc,
                "active_per_inst": av / iv, "busy_active": av * 4 / (4 * bc), "busy_insts_x2": iv * 2 / (4 * bc)}
        out.append(line)
        print("%-16s %9d %14.4g %16.4g %14.4g | %8.3f %14.3f %18.3f" % (k, grid, iv, av, bc, line["active_per_inst"], line["busy_active"], line["busy_insts_x2"]))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        res = report(sys.argv[2])
        if len(sys.argv) > 3:
            with open(sys.argv[3], "w") as f:
                json.dump(res, f, indent=1)
    else:
        run()
