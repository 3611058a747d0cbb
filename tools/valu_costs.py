#!/usr/bin/env python3
"""profiles/r05_valu_probe.json (tools/valu_probe.hip on an MI355X) -> profiles/r05_valu_costs.json: what one wave64 vector
instruction of each SQ_INSTS_VALU_* class costs a SIMD, in cycles, at four waves per SIMD (how the time-parallel kernels run).
tools/collect_profile.py prices a launch's instruction census with it (bench.py: roofline.valu, valu_frac).

Two price lists. `pure`: the class alone in an independent stream -- the least an instruction of the class can cost, so the
fraction it gives is a LOWER bound on how much of the issue capacity the kernel uses. `in_mix`: what the class costs among
4-cycle instructions -- the probe's alternating streams: a 2-cycle VOP1/VOP2 instruction pairs only with another one, behind
an f64 / DPP / conversion it costs 3.7 -- which is what the kernels' streams look like."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_valu_probe.json")
rows = {r["inst"]: r["cycles_per_inst_per_simd"] for r in json.load(open(src))["rows"] if r["waves_per_simd"] == 4}


def c(name):
    return rows[name] if name in rows else rows[[k for k in rows if k.startswith(name)][0]]


def avg(*names):
    return sum(c(n) for n in names) / len(names)


alt_f64_u32 = c("alternating v_add_f64 / v_add_u32")
simple_in_mix = 2 * alt_f64_u32 - c("v_add_f64")  # what the v_add_u32 of the alternating stream costs
pure = {
    "ADD_F64": c("v_add_f64"), "MUL_F64": c("v_mul_f64"), "FMA_F64": c("v_fma_f64"), "TRANS_F64": 2 * c("v_rcp_f32"),
    "ADD_F32": avg("v_add_f32", "v_sub_f32_e32"), "MUL_F32": c("v_mul_f32"),
    "FMA_F32": avg("v_fma_f32", "v_fmac_f32_e32", "v_pk_fma_f32"), "TRANS_F32": c("v_rcp_f32"),
    "CVT": avg("v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f64_u32", "v_cvt_i32_f64", "v_cvt_i32_f32", "v_cvt_f32_i32"),
    # integer: adds / subs / shifts / logic at the 2-cycle rate, v_lshl_add_u32 / v_mul_lo_u32 at the 4-cycle one; the
    # closed-form kernel's hot path has five of the first kind per one of the second (disassembly census)
    "INT32": (5 * avg("v_add_u32", "v_lshrrev_b32_e32", "v_and_b32") + c("v_lshl_add_u32")) / 6,
    "INT64": c("v_lshl_add_u64"),
    # what SQ_INSTS_VALU counts beyond the classes above: moves (DPP and plain), compares, selects, lane reads. In the
    # hot path: 5.3 DPP moves, 1 compare and 7 plain moves / selects per 13.3 (same census)
    "OTHER": (5.3 * c("v_mov_b32_dpp wave_shr:1") + 1.0 * c("v_cmp_eq_u32_e64") + 3.5 * c("v_mov_b32") +
              3.5 * c("v_cndmask_b32_e64")) / 13.3,
}
in_mix = dict(pure)
for k in ("ADD_F32", "MUL_F32"):
    in_mix[k] = simple_in_mix
in_mix["INT32"] = (5 * simple_in_mix + c("v_lshl_add_u32")) / 6
in_mix["OTHER"] = (5.3 * c("v_mov_b32_dpp wave_shr:1") + 1.0 * c("v_cmp_eq_u32_e64") + 3.5 * simple_in_mix +
                   3.5 * c("v_cndmask_b32_e64")) / 13.3
out = {"source": "profiles/r05_valu_probe.json (tools/valu_probe.hip, MI355X, 256 workgroups x 1024 threads = 4 waves per SIMD, "
                 "independent instructions over 16 register streams; cycles = wall time x the clock measured by s_memtime / s_memrealtime)",
       "unit": "cycles per wave64 instruction per SIMD",
       "pure": {k: round(v, 3) for k, v in pure.items()}, "in_mix": {k: round(v, 3) for k, v in in_mix.items()},
       "flat_in_mix": round(c("hot-path mix (16"), 3),
       "note": "pure: the class alone (a lower bound on its cost: the VALU fraction priced with it is a lower bound); in_mix: "
               "2-cycle VOP1/VOP2 instructions priced as they cost next to a 4-cycle one (alternating streams of the probe: "
               f"v_add_f64 / v_add_u32 {alt_f64_u32:.2f} per instruction); flat_in_mix: the probe's 36-instruction hot-path mix, "
               "per instruction, for kernels without a class census"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_valu_costs.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
