#!/usr/bin/env python3
"""Per-construct instruction census of one kernel of the library, from its ISA.

    python tools/isa_census.py build                       # device code object with line tables -> /tmp/sau_census/kg.elf
    python tools/isa_census.py lines  'fast_kernel<8, 2, false, false>'           # static counts per fast_voice source line
    python tools/isa_census.py census 'fast_kernel<8, 2, false, false>' RULES.json OUT.json

The hardware counters (SQ_INSTS_VALU_*) split a launch's vector instructions into arithmetic classes and leave
"everything else" -- moves, DPP moves, lane reads, compares, selects -- as one bucket. This tool gives that bucket names.
Every instruction of the kernel is attributed, through the line table and the inline stack (llvm-symbolizer --inlines),
to (a) the statement of fast_voice it belongs to and (b) the helper it was inlined from (wave_incl_scan64_dpp, lookback64,
ras_sample, fk_poly, ...). All branches of the row-group loop are wave-uniform (decoded steps are scalar data), and a
group's rows are unrolled, so an instruction on a workload's path executes exactly once per (row group, step that takes
its branch): the dynamic count of a construct per row group is the sum of the static counts of its statements times how
many of the voice's steps take that statement -- RULES.json says that per source line range for one workload (a voice's
decoded steps are known: tests/tools/gpu_decoded_steps.py prints them). The total is checked against the launch's measured
SQ_INSTS_VALU (tools/collect_profile.py) -- the census is only as good as that agreement, which is printed and stored."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = "/tmp/sau_census"
LLVM = "/opt/rocm/lib/llvm/bin/"
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-fvisibility=hidden",
         "-fvisibility-inlines-hidden", "-gline-tables-only", "--cuda-device-only"]


def build(defines=()):
    os.makedirs(WORK, exist_ok=True)
    src = os.path.join(ROOT, "saugns_amd", "csrc", "hip_backend.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-D" + d for d in defines] + ["-c", src, "-o", WORK + "/kg.co"])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + WORK + "/kg.co",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + WORK + "/kg.elf"])
    print(WORK + "/kg.elf")


def classify(op):
    """instruction class as the hardware counters see it + a finer name for the vector ones"""
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime", "s_atomic")):
        return "SMEM", op
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier", "s_waitcnt", "s_nop", "s_sleep",
                      "s_sethalt", "s_setprio", "s_getreg", "s_setreg")):
        return "CTRL", op
    if op.startswith("s_"):
        return "SALU", op
    if op.startswith("ds_"):
        return "LDS", op
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM", op
    if not op.startswith("v_"):
        return "OTHER", op
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    dpp = op.endswith("_dpp")
    f64 = "_f64" in base and not base.startswith("v_cvt")
    if base.startswith("v_cvt") or base in ("v_rndne_f32", "v_trunc_f32", "v_floor_f32", "v_ceil_f32", "v_fract_f32", "v_ldexp_f64",
                                            "v_frexp_mant_f32", "v_frexp_exp_i32_f32", "v_ldexp_f32", "v_rndne_f64", "v_floor_f64"):
        fine = "cvt/round"
    elif f64:
        fine = "f64 arithmetic"
    elif base.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        fine = "transcendental"
    elif base.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        fine = "lane read/write (v_readlane, v_writelane)"
    elif base.startswith("v_mov") and dpp:
        fine = "DPP move (v_mov_b32_dpp)"
    elif dpp:
        fine = "DPP fused into an ALU instruction"
    elif base.startswith(("v_mov", "v_accvgpr")):
        fine = "move (v_mov_b32/b64)"
    elif base.startswith("v_cmp"):
        fine = "compare"
    elif base.startswith("v_cndmask"):
        fine = "select (v_cndmask)"
    elif base.startswith(("v_mbcnt", "v_bfe", "v_bfi", "v_perm", "v_alignbit", "v_bitop3", "v_and", "v_or", "v_xor", "v_not", "v_lshl", "v_lshr",
                          "v_ashr")):
        fine = "bit / shift"
    elif re.search(r"_(u32|i32|u64|i64|co_u32|u16|i16|u24|i24)$", base) or base.startswith(("v_mad_u", "v_mad_i", "v_mul_lo", "v_mul_hi",
                                                                                             "v_lshl_add", "v_add3", "v_sub_co", "v_add_co",
                                                                                             "v_addc", "v_subb", "v_min_u", "v_max_u", "v_min_i",
                                                                                             "v_max_i")):
        fine = "integer arithmetic"
    elif "_f32" in base or "_f16" in base:
        fine = "f32 arithmetic"
    else:
        fine = "other vector"
    return "VALU", fine


def lambda_names():
    """the lambdas of fast_voice by the source lines of their bodies (read off the source: `auto name = [&]`)"""
    names = {}
    src = open(os.path.join(ROOT, "saugns_amd", "csrc", "k_fast_group.h")).read().splitlines()
    cur, depth = None, 0
    for n, ln in enumerate(src, 1):
        m = re.match(r"\s*auto (\w+) = \[&\]", ln)
        if m and cur is None:
            cur, depth = m.group(1), 0
        if cur:
            names[n] = cur
            depth += ln.count("{") - ln.count("}")
            if depth <= 0 and ("};" in ln or ln.rstrip().endswith("};")):
                cur = None
    return names


LAMBDAS = lambda_names()


def kernel_range(name):
    out = subprocess.run(["nm", "-C", "--print-size", "--defined-only", WORK + "/kg.elf"], capture_output=True, text=True).stdout
    for ln in out.splitlines():
        p = ln.split(None, 3)
        if len(p) == 4 and p[2] in "Tt" and (p[3].startswith(("void sauhip::" + name + "(", "sauhip::" + name + "(")) or p[3] == name):
            return int(p[0], 16), int(p[1], 16)
    raise SystemExit("no kernel named %s in the code object" % name)


def load(name):
    lo, size = kernel_range(name)
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--start-address=%#x" % lo,
                          "--stop-address=%#x" % (lo + size), WORK + "/kg.elf"], capture_output=True, text=True).stdout
    ins = []
    rng = os.environ.get("CENSUS_RANGE")  # "lo:hi", byte offsets within the kernel: one copy of a group body that is there twice
    for ln in dis.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", ln)
        if m:
            a = int(m.group(3), 16)
            if rng and not (int(rng.split(":")[0], 0) <= a - lo < int(rng.split(":")[1], 0)):
                continue
            ins.append({"addr": a, "op": m.group(1), "args": m.group(2)})
    sym = subprocess.run([LLVM + "llvm-symbolizer", "--obj=" + WORK + "/kg.elf", "--inlines", "--functions=short", "--output-style=JSON"],
                         input="\n".join("%#x" % i["addr"] for i in ins), capture_output=True, text=True).stdout
    for i, ln in zip(ins, sym.splitlines()):
        fr = json.loads(ln)["Symbol"]
        stack = [(os.path.basename(f["FileName"]), f["Line"], re.sub(r"<.*", "", f["FunctionName"])) for f in fr]
        i["stack"] = stack
        # the statement of fast_voice this instruction belongs to (the outermost frame in fast_voice), the lines of the lambdas
        # defined in fast_voice on the way down (prev_of, lead32, freq_at ...), and the helper chain below them
        fvs = [k for k, f in enumerate(stack) if f[0] == "k_fast_group.h" and f[2] in ("fast_voice", "operator()")]
        if not fvs:
            i["fv_line"] = None
            i["inner"] = []
            i["helpers"] = [f[2] for f in reversed(stack[:-1]) if f[2] != "operator()"]
        else:
            i["fv_line"] = stack[fvs[-1]][1]
            i["inner"] = [stack[k][1] for k in reversed(fvs[:-1])]
            i["helpers"] = [LAMBDAS.get(stack[k][1], "lambda@%d" % stack[k][1]) for k in reversed(fvs[:-1])] + \
                           [f[2] for f in reversed(stack[:fvs[0]]) if f[2] != "operator()"]
        i["line0"] = "%s:%d" % (stack[0][0], stack[0][1])
    # instructions without a line (spill code, copies the register allocator made): the statement of the instruction before
    prev = None
    for i in ins:
        if i["stack"][0][1] == 0 and prev is not None:
            i["fv_line"], i["inner"], i["helpers"] = prev["fv_line"], prev["inner"], prev["helpers"]
            i["noline"] = True
        else:
            prev = i
    return ins


def cmd_lines(name):
    ins = load(name)
    per = collections.defaultdict(lambda: collections.Counter())
    for i in ins:
        cls, fine = classify(i["op"])
        key = i["fv_line"] if i["fv_line"] is not None else -1
        per[key][cls] += 1
        if cls == "VALU":
            per[key]["helper:" + (">".join(i["helpers"]) or "-")] += 1
    for k in sorted(per):
        c = per[k]
        hs = ", ".join("%s %d" % (h[7:], n) for h, n in c.most_common() if h.startswith("helper:"))
        print("%5d  VALU %5d SALU %5d LDS %4d SMEM %4d VMEM %4d CTRL %4d | %s" % (k, c["VALU"], c["SALU"], c["LDS"], c["SMEM"], c["VMEM"], c["CTRL"], hs))


def cmd_census(name, rules_file, out_file):
    """RULES.json: {"workload": ..., "rows": T, "ops_per_voice": n, "groups_per_launch": g (optional, for the check against PMC),
    "measured": {"SQ_INSTS_VALU": per launch (wave-instructions), ...} (optional),
    "rules": [[first_line, last_line, times_per_group, "why"], ...]}  -- lines of k_fast_group.h (the row group's evaluation in fast_voice);
    a line no rule names executes 0 times (cold: segment edges, hold resolution, other step kinds)."""
    rules = json.load(open(rules_file))
    ins = load(name)
    mult = {}
    for lo, hi, n, _why in rules["rules"]:
        for ln in range(lo, hi + 1):
            mult[ln] = n
    cold_inner = set()
    for lo, hi, _why in rules.get("cold_inner", []):
        cold_inner.update(range(lo, hi + 1))
    helper_cold = set()
    for fn, lo, hi, _why in rules.get("helper_cold", []):
        helper_cold.update((fn, ln) for ln in range(lo, hi + 1))
    # device functions the kernel really calls (not inlined): their instructions x calls per row group, by their own lines
    for x in rules.get("called_functions", []):
        hot = set()
        for lo, hi in x["hot_lines"]:
            hot.update(range(lo, hi + 1))
        for i in load(x["symbol"]):
            i["fv_line"] = "call"
            i["inner"] = []
            i["helpers"] = [x["as"]] + i["helpers"]
            i["times"] = x["times"] if all(f[1] in hot or f[1] == 0 for f in i["stack"] if f[0] == x["file"]) else 0
            ins.append(i)
    tot = collections.Counter()
    by_fine = collections.Counter()
    by_construct = collections.Counter()
    by_construct_fine = collections.defaultdict(collections.Counter)
    for i in ins:
        n = i["times"] if "times" in i else mult.get(i["fv_line"], 0)
        if any(ln in cold_inner for ln in i["inner"]) or any((f[0], f[1]) in helper_cold for f in i["stack"]):
            n = 0
        if not n:
            continue
        cls, fine = classify(i["op"])
        tot[cls] += n
        if cls == "VALU":
            by_fine[fine] += n
            c = i["helpers"][0] if i["helpers"] else "fast_voice (step interpreter, row loops)"
            if len(i["helpers"]) > 1 and i["helpers"][0] in ("lookback32", "lookback64", "ras_sample"):
                c = i["helpers"][0] + " > " + i["helpers"][1]
            by_construct[c] += n
            by_construct_fine[c][fine] += n
    T, nops = rules["rows"], rules["ops_per_voice"]
    per_opsample = lambda x: x / (T * nops)  # a group-step instruction serves 64 lanes x T rows; per operator-sample lane-instructions = count x 64 / (64 T nops)
    arith = ("f64 arithmetic", "f32 arithmetic", "integer arithmetic", "cvt/round", "transcendental")
    out = {
        "kernel": name, "workload": rules["workload"], "rows_per_group": T, "operators_per_voice": nops,
        "per_row_group": dict(tot),
        "valu_lane_instructions_per_operator_sample": round(per_opsample(tot["VALU"]), 2),
        "valu_by_kind_per_operator_sample": {k: round(per_opsample(v), 2) for k, v in by_fine.most_common()},
        "valu_non_arithmetic_frac": round(sum(v for k, v in by_fine.items() if k not in arith) / max(1, tot["VALU"]), 3),
        "valu_by_construct_per_operator_sample": {k: {"total": round(per_opsample(v), 2),
                                                      **{f: round(per_opsample(n), 2) for f, n in by_construct_fine[k].most_common()}}
                                                  for k, v in by_construct.most_common()},
        "method": "static ISA counts per fast_voice statement (line table + inline stacks) x steps of the voice that take the statement; "
                  "all branches in the row-group loop are wave-uniform and rows are unrolled",
    }
    if "measured" in rules and "groups_per_launch" in rules:
        g = rules["groups_per_launch"]
        chk = {}
        for k, cls in (("SQ_INSTS_VALU", "VALU"), ("SQ_INSTS_SALU", "SALU"), ("SQ_INSTS_LDS", "LDS"), ("SQ_INSTS_SMEM", "SMEM")):
            if k in rules["measured"]:
                chk[k] = {"census": tot[cls] * g, "measured": rules["measured"][k], "ratio": round(tot[cls] * g / rules["measured"][k], 3)}
        out["check_against_pmc"] = chk
    json.dump(out, open(out_file, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "lines":
        cmd_lines(sys.argv[2])
    elif sys.argv[1] == "census":
        cmd_census(sys.argv[2], sys.argv[3], sys.argv[4])
