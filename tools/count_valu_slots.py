#!/usr/bin/env python3
"""Counts the VALU issue slots per path-step of the shipped pricing kernels from their gfx950 ISA.

The in-register path has no HBM traffic and no matrix work: its roofline is the VALU issue rate.
One "slot" = one full-rate wave64 VALU instruction = 2 cycles on a SIMD-32 (v_fma_f32, v_xor_b32,
v_bitop3_b32 ...).  Instructions that issue slower are weighted by their measured cost
(tools/ubench_valu.hip on MI355X -> profiles/r01_valu_issue_costs.json):
    2 cycles  (1 slot)   f32 fma/mul/add, 32-bit add/sub/and/or/xor/bitop3, v_mov_b32 — with every source in a vector
                         register, an inline constant or a literal; the SAME instruction reading a scalar register
                         issues in 4 cycles (tools/ubench_bank.hip -> profiles/r02_operand_costs.txt), and is counted so
    4 cycles  (2 slots)  every fp64 arithmetic op, 32-bit integer multiplies incl. v_mad_u64_u32, LEFT shifts
                         (v_lshlrev_b32, v_lshl_add/or; the right shifts v_lshrrev_b32 / v_ashrrev_i32 are full rate),
                         v_bfe, v_perm, v_and_or, v_add3, SDWA forms, 64-bit shifts, conversions, compares, v_cndmask,
                         v_mov_b64, packed f32
    8 cycles  (4 slots)  f32 transcendentals (exp, log, sin, cos, sqrt, rcp, rsq)
   16 cycles  (8 slots)  v_rcp_f64 / v_rsq_f64 / v_sqrt_f64
The tool compiles csrc/price_f64.hip, csrc/price_f32.hip and csrc/store.hip to assembly with the build's flags, finds the step
loop of each kernel (the innermost loop that contains the Philox multiplies), and writes
profiles/valu_slots.json (read by bench.py) and profiles/isa_slots.md (the per-instruction table).
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monte-carlo-project-cuda_amd", "csrc")

HALF_PREFIX = ("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64", "v_mad_u32", "v_mul_u32", "v_mul_i32", "v_cvt",
               "v_cndmask", "v_cmp", "v_lshl", "v_alignbit", "v_add3", "v_bfe", "v_bfi",
               "v_mov_b64", "v_and_or", "v_or3", "v_readlane", "v_readfirstlane", "v_ldexp", "v_frexp", "v_fract",
               "v_rndne", "v_trunc", "v_floor", "v_ceil", "v_add_co", "v_addc", "v_sub_co", "v_subb", "v_subrev_co",
               "v_subbrev", "v_pk_", "v_perm", "v_mbcnt", "v_max_f64", "v_min_f64")


SGPR_OPERAND = re.compile(r"^(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|ttmp\d+)$")


def reads_sgpr(line: str) -> bool:
    """True when a source operand of the instruction is a scalar register (the first operand is the destination)."""
    parts = line.split(None, 1)
    if len(parts) < 2:
        return False
    ops = [o.strip() for o in parts[1].split(",")]
    return any(SGPR_OPERAND.match(re.sub(r"^[-|]+|[|]+$", "", o.split()[0])) for o in ops[1:] if o)


def line_cycles(line: str) -> int:
    op = line.split()[0]
    c = cycles(op)
    if c == 2 and reads_sgpr(line):
        return 4
    return c


def cycles(op: str) -> int:
    if not op.startswith("v_"):
        return 0
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return 16
    if re.match(r"v_(exp|log|sin|cos|sqrt|rcp|rsq)_f32", op):
        return 8
    if "_f64" in op or "_b64" in op or "_i64" in op or "_u64" in op or op.endswith("_sdwa") or op.startswith(HALF_PREFIX):
        return 4
    return 2


def measured_costs():
    """Issue cost per instruction as MEASURED on MI355X by tools/ubench_valu.hip (profiles/r01_valu_issue_costs.json:
    cycles per wave64 instruction per SIMD, 8 waves per SIMD).  Instructions the microbenchmark did not cover fall
    back to the class weight of cycles()."""
    path = os.path.join(ROOT, "profiles", "r01_valu_issue_costs.json")
    table = {}
    try:
        with open(path) as f:
            for row in json.load(f)["rows"]:
                table[row["inst"].split("(")[0]] = row["cycles_per_wave_inst_per_simd"]
    except (OSError, ValueError, KeyError):
        return None
    alias = {"v_fmac_f64": "v_fma_f64", "v_fmac_f32": "v_fma_f32", "v_fmamk_f32": "v_fma_f32", "v_fmaak_f32": "v_fma_f32",
             "v_sub_u32": "v_add_u32", "v_and_b32": "v_xor_b32", "v_or_b32": "v_xor_b32", "v_mov_b32": "v_xor_b32",
             "v_lshrrev_b32": "v_lshlrev_b32", "v_ashrrev_i32": "v_lshlrev_b32", "v_bfe_u32": "v_lshlrev_b32",
             "v_cvt_f64_u32": "v_cvt_f32_u32", "v_cvt_f64_i32": "v_cvt_f32_u32", "v_cvt_i32_f64": "v_cvt_f32_u32",
             "v_pk_mul_f32": "v_pk_fma_f32", "v_cmp_le_i32": "v_cmp_gt_f32", "v_cmp_gt_f64": "v_cmp_gt_f64",
             "v_sin_f32": "v_sin_f32", "v_cos_f32": "v_cos_f32"}

    def cost(op: str) -> float:
        base = re.sub(r"_e(32|64)$", "", op)
        base = alias.get(base, base)
        if base in table:
            return table[base]
        return float(cycles(op))
    return cost


def build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mcamd_build", os.path.join(ROOT, "monte-carlo-project-cuda_amd", "build.py"))
    bmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bmod)
    return bmod


def asm_of(src: str, hipcc: str, flags) -> str:
    """Assembly of one kernel source, compiled with the library's own compiler and flags (build.py)."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc, *flags, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out])
        return open(out).read()


def step_loop(asm: str, symbol: str, which: str = "shortest"):
    # `symbol` is the mangled name up to and including the template arguments; the parameter list may follow
    m = re.search(r"^%s\w*:.*?\.Lfunc_end" % re.escape(symbol), asm, re.S | re.M)
    if not m:
        raise SystemExit(f"kernel {symbol} not found")
    lines = [l.strip() for l in m.group(0).split("\n")]
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(lines):
        mm = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"s_branch (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            # basic blocks the source marks as rare (mc_device.hpp below_barrier: the exact barrier test, taken by a
            # wavefront a few times per ten thousand steps) are not part of the per-step count.  A block runs from a
            # label or a "; %bb.N:" comment to the next one.
            seg = lines[labels[mm.group(1)]:i + 1]
            starts = [k for k, x in enumerate(seg) if re.match(r"^\.LBB\d+_\d+:", x) or x.startswith("; %bb.")] + [len(seg)]
            body = []
            for b0, b1 in zip(starts[:-1], starts[1:]):
                blk = seg[b0:b1]
                if not any("MCAMD_RARE_BLOCK" in x for x in blk):
                    body.extend(blk)
            ins = [x for x in body if x and not x.startswith((".", ";")) and not x.endswith(":")]
            ops = [x.split()[0] for x in ins]
            # the step loop carries a whole Philox4x32-10 (20 32x32->64 multiplies, a few hoisted): other loops of a
            # kernel may hold a stray multiply (64-bit index arithmetic) and must not be mistaken for it
            if ops.count("v_mad_u64_u32") + ops.count("v_mul_hi_u32") >= 12:
                loops.append(ins)
    if not loops:
        raise SystemExit(f"no Philox loop in {symbol}")
    n_mul = lambda ins: sum(1 for x in ins if x.split()[0] in ("v_mad_u64_u32", "v_mul_hi_u32"))
    if which == "fewest_mul":    # the loop whose first Philox round is half scalar (a wave-uniform block index)
        return min(loops, key=lambda ins: (n_mul(ins), len(ins)))
    if which == "most_mul":      # the loop with a per-lane block index; innermost = shortest among those
        top = max(n_mul(ins) for ins in loops if n_mul(ins) <= 24)
        return min((ins for ins in loops if n_mul(ins) == top), key=len)
    return min(loops, key=len)


KERNELS = {
    # key: (source, mangled kernel, path-steps advanced per loop iteration)
    # window-less in-register pricing: the default loop sums the log-returns (LOGSPACE = true); the product form
    # (MCAMD_FLAG_PRODUCT_FORM) multiplies the running price every step
    # (the pair-sum loop walks kPairSumPaths = 2 paths per thread: 2 x 2 / 2 x 4 path-steps per iteration)
    "price_f64": ("price_f64.hip", "_ZN5mcamd12price_kernelIdLb0ELb1ELi0EEE", 4),
    "price_f64_product": ("price_f64.hip", "_ZN5mcamd12price_kernelIdLb0ELb0ELi0EEE", 2),
    "price_f32": ("price_f32.hip", "_ZN5mcamd12price_kernelIfLb0ELb1ELi0EEE", 8),
    "price_f32_product": ("price_f32.hip", "_ZN5mcamd12price_kernelIfLb0ELb0ELi0EEE", 4),
    "store_f32": ("store.hip", "_ZN5mcamd12store_kernelIfLb0ELi0ELb1EEE", 16),
    # nested MC inner stage, fp64, barrier window (BASELINE configs[3]): St is evaluated at every step for the count
    # two step loops since the lane compaction (csrc/nmc_compact.hpp): batches of fresh paths (block index uniform) and
    # batches of resumed ones (block index per lane: the first Philox round loses its scalar half)
    # (the default loops carry ln(St/S0): LOGSPACE = true; *_product = MCAMD_FLAG_PRODUCT_FORM)
    "nmc_wave_f64_window": ("nmc.hip", "_ZN5mcamd15nmc_wave_kernelIdLb1ELi0ELb1EEE", 2, "fewest_mul"),
    "nmc_wave_f64_window_resumed": ("nmc.hip", "_ZN5mcamd15nmc_wave_kernelIdLb1ELi0ELb1EEE", 2, "most_mul"),
    "nmc_wave_f64_window_product": ("nmc.hip", "_ZN5mcamd15nmc_wave_kernelIdLb1ELi0ELb0EEE", 2, "fewest_mul"),
    "price_f64_window": ("price_f64.hip", "_ZN5mcamd12price_kernelIdLb1ELb1ELi0EEE", 2),
    "price_f64_window_product": ("price_f64.hip", "_ZN5mcamd12price_kernelIdLb1ELb0ELi0EEE", 2),
}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--hipcc", default=None, help="compiler (default: the one build.py finds)")
    ap.add_argument("--extra-flag", action="append", default=[], help="extra compile flag of a variant build (repeatable)")
    args = ap.parse_args()
    bmod = build_module()
    hipcc = args.hipcc or bmod.hipcc()
    extra = tuple(args.extra_flag)
    flags = [f for f in bmod._flags(extra) if f != "-fPIC"]
    cache, result, md = {}, {}, ["# VALU issue slots per path-step (gfx950 ISA of the shipped inner loops)\n",
                                 "Generated by tools/count_valu_slots.py; weights from profiles/r01_valu_issue_costs.json.\n"]
    cost = measured_costs()
    for key, spec_ in KERNELS.items():
        src, sym, steps = spec_[:3]
        asm = cache[src] if src in cache else cache.setdefault(src, asm_of(src, hipcc, flags))
        lines_ = step_loop(asm, sym, spec_[3] if len(spec_) > 3 else "shortest")
        # an instruction is keyed by its opcode, plus "(sgpr)" when a full-rate opcode reads a scalar register
        keyed = [(l.split()[0] + (" (sgpr src)" if cycles(l.split()[0]) == 2 and reads_sgpr(l) else ""), line_cycles(l))
                 for l in lines_]
        ins = [k for k, _ in keyed]
        cost_of = dict(keyed)
        cnt = collections.Counter(ins)
        cyc = sum(cost_of[op] * n for op, n in cnt.items())
        valu = sum(n for op, n in cnt.items() if op.startswith("v_"))
        slots = cyc / 2.0 / steps
        result[key] = round(slots, 2)
        result[key + "_detail"] = {"loop_instructions": len(ins), "valu_instructions": valu, "issue_cycles_per_iteration": cyc,
                                   "path_steps_per_iteration": steps}
        if cost is not None:   # the same count priced with the per-instruction costs measured on the chip
            result[key + "_detail"]["measured_cost_cycles_per_iteration"] = round(
                sum((cost(op.split()[0]) + (2.0 if op.endswith("(sgpr src)") else 0.0)) * n
                    for op, n in cnt.items() if op.startswith("v_")), 1)
        md.append(f"\n## {key}: `{sym}`\n")
        md.append(f"{len(ins)} instructions per loop iteration ({valu} VALU), {steps} path-steps per iteration, "
                  f"{cyc} issue cycles -> **{slots:.1f} slots per path-step**"
                  + (f" (priced with the issue costs measured on MI355X, r01_valu_issue_costs.json: "
                     f"{result[key + '_detail']['measured_cost_cycles_per_iteration']} cycles per iteration)" if cost else "") + "\n")
        md.append("| count | instruction | cycles each |\n|---:|---|---:|")
        for op, n in sorted(cnt.items(), key=lambda kv: (-cost_of[kv[0]] * kv[1], kv[0])):
            md.append(f"| {n} | `{op}` | {cost_of[op]} |")
    # the id of the sources these counts were taken from (== mcamd_build_id() of a library built from them)
    result["build_id"] = bmod.build_id(extra)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "valu_slots.json"), "w") as f:
        json.dump(result, f, indent=1)
    with open(os.path.join(ROOT, "profiles", "isa_slots.md"), "w") as f:
        f.write("\n".join(md) + "\n")
    print(json.dumps({k: v for k, v in result.items() if not k.endswith("_detail")}))


if __name__ == "__main__":
    sys.exit(main())
