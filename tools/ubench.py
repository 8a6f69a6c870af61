#!/usr/bin/env python3
"""Run tools/ubench.hip on the GPU and print the measured rates as JSON (measurement infrastructure).

    python tools/ubench.py > profiles/r02_ubench.json

valu: wave64 integer instructions per second per SIMD, and the cycles each one takes at 2.4 GHz (the chip's
maximum clock; the clock held under load is lower, so the true cycle count is a little below the number printed);
lds / l2: bytes per second, whole chip."""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ubench.hip")
LIB = os.path.join(HERE, "libpo_ubench.so")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950",
                               "-fPIC", "-shared", "-o", LIB, SRC])
    return LIB


def load():
    lib = ctypes.CDLL(build())
    lib.ub_valu.restype = ctypes.c_double
    lib.ub_valu.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ub_lds.restype = ctypes.c_double
    lib.ub_lds.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.ub_l2.restype = ctypes.c_double
    lib.ub_l2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
    lib.ub_rand_lines.restype = ctypes.c_double
    lib.ub_rand_lines.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return lib


def measure_lines(quick=False):
    """Random 64-byte lines per second (whole chip) by table size, lanes per line, loads in flight per lane and waves per
    SIMD: what a hash-table probe costs on this memory system wherever the table lives (VERDICT r3 #3a)."""
    lib = load()
    it = 200 if quick else 1000
    out = {}
    for mb in (4, 64, 512, 2048):
        row = {}
        for lpl in (1, 4):
            for depth in (4, 8):
                for w in ((8,) if quick else (2, 4, 8)):
                    r = lib.ub_rand_lines(mb << 20, w, it, lpl, depth)
                    row["lanes_per_line_%d_depth_%d_waves_%d" % (lpl, depth, w)] = round(r / 1e9, 2)
        row["best_G_lines_per_s_1_lane_per_line"] = max(v for k, v in row.items() if k.startswith("lanes_per_line_1"))
        row["best_G_lines_per_s_4_lanes_per_line"] = max(v for k, v in row.items() if k.startswith("lanes_per_line_4"))
        out["%d_MB" % mb] = row
    return out


def measure(quick=False):
    lib = load()
    it = 20000 if quick else 100000
    out = {"valu": {}, "lds_random_b64_TBps": {}, "l2_hit_dwordx4_TBps": {}}
    ops = ("v_alignbit_b32 v,v,v", "v_xor_b32", "v_lshrrev_b32", "v_alignbit_b32 v,v,imm", "v_alignbit_b32 v,v,s",
           "v_or3_b32 v,v,v", "v_and_or_b32 v,v,v", "v_lshl_or_b32 v,imm,v", "v_bfe_u32 v,imm,imm", "v_add3_u32 v,v,v",
           "v_perm_b32 v,v,v", "v_bitop3_b32 v,v,v", "v_and_b32 v,s", "v_or3_b32 v,v,s", "v_mad_u32_u24 v,v,v",
           "v_mov_b32_dpp row_shr:1", "v_mov_b32_dpp wave_shr:1", "v_add_u32_dpp row_shr:1", "v_add_u32_dpp wave_shl:1",
           "v_min_u32", "v_cndmask_b32 vcc", "v_cndmask_b32_e64 sgpr mask", "v_addc_co_u32 vcc", "v_cmp_ne_u32 + v_xor_b32 (2 instr)",
           "v_add_u32", "v_sub_u32", "v_and_b32 v,v", "v_or_b32 v,v", "v_max_u32", "v_lshlrev_b32", "v_mov_b32", "v_min_i32")
    for op, name in enumerate(ops):
        out["valu"][name] = {}
        for w in ((1, 2, 4, 8) if op < 3 else (2, 8)):
            r = lib.ub_valu(op, w, it)
            out["valu"][name]["%d_waves_per_simd" % w] = {"wave_insts_per_sec_per_simd": r, "cycles_per_inst_at_2.4GHz": 2.4e9 / r}
    for w in (1, 2, 4, 8):
        out["lds_random_b64_TBps"]["%d_waves_per_simd" % w] = lib.ub_lds(w, it // 10) / 1e12
    for w in (2, 4, 8):
        out["l2_hit_dwordx4_TBps"]["%d_waves_per_simd" % w] = lib.ub_l2(w, it // 20, 2 << 20) / 1e12
    out["l2_miss_64MB_per_xcd_TBps"] = lib.ub_l2(8, it // 40, 64 << 20) / 1e12
    best = max(v["wave_insts_per_sec_per_simd"] for d in out["valu"].values() for v in d.values())
    out["random_64B_lines_G_per_s"] = measure_lines(quick)
    out["peaks"] = {"valu_wave_insts_per_sec_per_simd": best,
                    "valu_wave_insts_per_sec_chip": best * 1024,
                    "lds_random_b64_TBps": max(out["lds_random_b64_TBps"].values()),
                    "l2_hit_TBps": max(out["l2_hit_dwordx4_TBps"].values()),
                    "random_lines_G_per_s": {k: v["best_G_lines_per_s_1_lane_per_line"] for k, v in out["random_64B_lines_G_per_s"].items()}}
    return out


if __name__ == "__main__":
    if "--build" in sys.argv:
        print(build(force=True))
    elif "--lines" in sys.argv:
        print(json.dumps(measure_lines("--quick" in sys.argv), indent=1))
    else:
        print(json.dumps(measure("--quick" in sys.argv), indent=1))
