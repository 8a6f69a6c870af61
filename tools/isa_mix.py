#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel, per basic block (measurement infrastructure).

    python3 tools/isa_mix.py 'k_verify_a<2, false, true, false>'   [--blocks]

Compiles the library's device code to assembly and classifies every vector-ALU instruction of the named kernel by the
issue cost MEASURED on the box (tools/ubench.hip, tools/valu_calib.py): 2.15 cycles for VOP1 / VOP2 encodings (`_e32`) and
v_bitop3_b32, 4.05 cycles for VOP3-encoded integer ops (`_e64` forms, v_alignbit_b32, v_bfe, v_or3, v_add3, v_perm,
v_mad_u32_u24, v_lshl_or, v_and_or, v_lshl_add, v_cndmask_e64, v_cmp_e64, DPP/SDWA forms counted as their base op),
8.1 cycles for 64-bit shifts / multiplies.  With SQ_INSTS_VALU and SQ_BUSY_CU_CYCLES of a launch that gives
    valu_busy = SQ_INSTS_VALU x (mean cycles per instruction of the mix) / (4 SIMDs x SQ_BUSY_CU_CYCLES)
-- the static mix of the loop blocks stands in for the dynamic one (the loop is where the launch spends its instructions)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "phasm_amd", "csrc", "c_api.hip")

HALF_MNEMONICS = ("v_alignbit_b32", "v_alignbyte_b32", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_or3_b32", "v_and_or_b32", "v_lshl_or_b32",
                  "v_add3_u32", "v_perm_b32", "v_mad_u32_u24", "v_mad_i32_i24", "v_lshl_add_u32", "v_add_lshl_u32", "v_xad_u32",
                  "v_min3_u32", "v_max3_u32", "v_med3_u32", "v_readlane_b32", "v_writelane_b32", "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32",
                  "v_mul_lo_u32", "v_mul_hi_u32", "v_sad_u32")
QUARTER_MNEMONICS = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64")
FULL_VOP3 = ("v_bitop3_b32",)
COST = {"full": 2.15, "half": 4.05, "quarter": 8.1}


def demangle_all(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def classify(mn):
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    if base in QUARTER_MNEMONICS:
        return "quarter"
    if base in FULL_VOP3:
        return "full"
    if base in HALF_MNEMONICS or mn.endswith("_e64") or mn.endswith("_e64_dpp"):
        return "half"
    return "full"


def main():
    want = sys.argv[1]
    show_blocks = "--blocks" in sys.argv
    asm = "/tmp/phasm_isa_mix.s"
    if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(os.path.dirname(SRC), "kernels.hip.h"))):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                               "-S", "-o", asm, SRC], stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = demangle_all([n for _, n in starts])
    hit = [(i, n) for (i, _), n in zip(starts, names) if want.replace(" ", "") in n.replace(" ", "").replace("void", "").replace("po::", "")]
    if not hit:
        sys.exit("kernel not found: %s" % want)
    i0 = hit[0][0]
    i1 = next(i for i in range(i0 + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], {"label": "entry", "full": 0, "half": 0, "quarter": 0, "salu": 0, "vmem": 0, "lds": 0, "branch": 0, "half_ops": {}}
    for l in lines[i0 + 1:i1 + 1]:
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", l.strip()):
                blocks.append(cur)
                cur = {"label": l.strip().rstrip(":"), "full": 0, "half": 0, "quarter": 0, "salu": 0, "vmem": 0, "lds": 0, "branch": 0, "half_ops": {}}
            continue
        mn = t.split()[0]
        if mn.startswith("v_") and not mn.startswith("v_mfma"):
            c = classify(mn)
            cur[c] += 1
            if c != "full":
                cur["half_ops"][mn] = cur["half_ops"].get(mn, 0) + 1
        elif mn.startswith("s_cbranch") or mn == "s_branch":
            cur["branch"] += 1
        elif mn.startswith("s_"):
            cur["salu"] += 1
        elif mn.startswith("global_") or mn.startswith("flat_") or mn.startswith("buffer_") or mn.startswith("scratch_"):
            cur["vmem"] += 1
        elif mn.startswith("ds_"):
            cur["lds"] += 1
    blocks.append(cur)
    tot = {k: sum(b[k] for b in blocks) for k in ("full", "half", "quarter", "salu", "vmem", "lds", "branch")}
    nv = tot["full"] + tot["half"] + tot["quarter"]
    mean = (tot["full"] * COST["full"] + tot["half"] * COST["half"] + tot["quarter"] * COST["quarter"]) / max(nv, 1)
    print("kernel %s: %d blocks, VALU %d (full-rate %d, half-rate %d, quarter-rate %d) SALU %d branch %d VMEM %d LDS %d" % (
        hit[0][1][:90], len(blocks), nv, tot["full"], tot["half"], tot["quarter"], tot["salu"], tot["branch"], tot["vmem"], tot["lds"]))
    print("static mean issue cost %.3f cycles per VALU instruction" % mean)
    ops = {}
    for b in blocks:
        for k, v in b["half_ops"].items():
            ops[k] = ops.get(k, 0) + v
    print("not full rate:", ", ".join("%s x%d" % kv for kv in sorted(ops.items(), key=lambda kv: -kv[1])))
    if show_blocks:
        for b in blocks:
            n = b["full"] + b["half"] + b["quarter"]
            if n:
                print("  %-12s valu %3d (half %3d quarter %2d)  salu %3d  br %d  vmem %2d  lds %2d" % (b["label"], n, b["half"], b["quarter"], b["salu"], b["branch"], b["vmem"], b["lds"]))
    return mean


if __name__ == "__main__":
    main()
