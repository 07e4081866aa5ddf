#!/usr/bin/env python3
"""tools/valu_mix.py -- mean VALU issue cost (SIMD cycles per wave64 instruction) of every libygzf kernel -> profiles/valu_issue_cost.json.

Each kernel's gfx950 ISA (hipcc --cuda-device-only -S of the product sources, same flags as the build) is classified with the issue rates
measured by tools/micro/valu_peak.hip on the MI355X (profiles/micro/r02_valu_issue_rates.txt):
    2 cycles  v_add/sub/subrev_u32, v_and/or/xor/not_b32, v_lshrrev_b32, v_mov_b32, v_bitop3_b32, v_add/sub/mul/fma_f32 (and their carry forms)
    4 cycles  everything else that was measured (v_lerp_u8, v_alignbyte, v_perm, v_dot4, v_sad, v_pk_*, v_or3, v_and_or, v_lshl_or,
              v_lshl_add, v_add3, v_mul_u32_u24 / v_mad_u32_u24 / v_mul_lo / v_mul_hi, v_cmp*, v_cndmask, DPP / SDWA forms, v_mbcnt,
              v_lshlrev_b32, v_min/max*, v_bfe/bfi, v_cvt*) and, as the conservative default, anything not measured
    8 cycles  transcendental and double-precision forms (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos, *_f64) -- not measured, taken
              at a quarter of the full rate
Instructions inside loops dominate the dynamic mix: every enclosing loop (a backward branch in the ISA) multiplies an instruction's weight
by 10.  The result is an ESTIMATE of the dynamic mix, good to a few tenths of a cycle; bench.py multiplies it with the measured dynamic
instruction count (SQ_INSTS_VALU) for roofline_valu."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb_ygz_slam_amd", "csrc")
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_mov_b32", "v_bitop3_b32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32",
        "v_xnor_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
SLOW = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_|_f64")


def cost(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    if op.endswith(("_dpp", "_sdwa")):
        return 4
    if SLOW.search(base):
        return 8
    return 2 if base in FULL else 4


def kernels(asm):
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            cur = out.setdefault(name, [])
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur.append(("L", m.group(1)))
            continue
        m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)\s*(.*)", line)
        if m:
            cur.append(("I", m.group(1), m.group(2)))
    return out


def analyse(items):
    pos = {it[1]: i for i, it in enumerate(items) if it[0] == "L"}
    depth = [0] * len(items)
    for i, it in enumerate(items):
        if it[0] == "I" and it[1].startswith(("s_cbranch", "s_branch")):
            tgt = it[2].strip().split()[0] if it[2].strip() else ""
            if tgt in pos and pos[tgt] < i:
                for j in range(pos[tgt], i + 1):
                    depth[j] += 1
    tot_w = tot_c = flat_n = flat_c = 0.0
    hist = {}
    for i, it in enumerate(items):
        if it[0] != "I" or not it[1].startswith("v_"):
            continue
        w = 10.0 ** min(depth[i], 3)
        c = cost(it[1])
        tot_w += w; tot_c += w * c; flat_n += 1; flat_c += c
        hist[c] = hist.get(c, 0) + w
    if not tot_w:
        return None
    return {"cycles_per_inst": round(tot_c / tot_w, 2), "cycles_per_inst_static": round(flat_c / flat_n, 2), "static_valu_insts": int(flat_n),
            "weighted_share": {str(k): round(v / tot_w, 3) for k, v in sorted(hist.items())}}


def short(mangled):
    m = re.match(r"_ZN4ygzf(\d+)", mangled)
    n = mangled[m.end():m.end() + int(m.group(1))] if m else mangled
    return "k_pyr_resize" if n.startswith("k_pyr_resize") else n


def main():
    res = {}
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith("_kernels.hip")):
        with tempfile.TemporaryDirectory() as td:
            s = os.path.join(td, "k.s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                                   "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", s],
                                  stderr=subprocess.DEVNULL)
            for name, items in kernels(open(s).read()).items():
                a = analyse(items)
                if a is None:
                    continue
                k = short(name)
                if k not in res or a["static_valu_insts"] > res[k]["static_valu_insts"]:     # template variants: keep the largest body
                    res[k] = a
    res["_doc"] = "SIMD issue cycles per wave64 VALU instruction, loop-weighted ISA mix x measured issue rates; see tools/valu_mix.py"
    json.dump(res, open(os.path.join(ROOT, "profiles", "valu_issue_cost.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items()):
        if k != "_doc":
            print("%-24s %.2f cycles/inst (static %.2f, %d VALU)  %s" % (k, v["cycles_per_inst"], v["cycles_per_inst_static"], v["static_valu_insts"], v["weighted_share"]))


if __name__ == "__main__":
    main()
