"""Turn one round's rocprofv3 outputs under gpurun_out/ into the summaries kept in profiles/:
    python tools/collect_profile.py r01e r01_e
expects gpurun_out/prof_<tag>/<tag>_kernel_stats.csv, gpurun_out/bench_<tag>.json and the PMC
passes gpurun_out/pmc_<tag>_{fetch,write,sq,inst}/*_counter_collection.csv."""
import collections
import csv
import json
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
shutil.copy(f"gpurun_out/prof_{tag}/{tag}_kernel_stats.csv", f"profiles/{name}_bench_kernel_stats.csv")
bench_line = open(f"gpurun_out/bench_{tag}.json").read().strip().splitlines()[-1]
open(f"profiles/{name}_bench.json", "w").write(bench_line + "\n")
bench = json.loads(bench_line)
frames_per_step = bench["config"]["frames_per_step"]
out = {"command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu",
       "workload": {"name": "config3", "voices": bench["config"]["voices"], "operators": bench["config"]["operators"],
                    "frames_per_step": frames_per_step},
       # bench.py reports `traffic` from this file only while the kernel sources are the ones profiled
       "kernel_source_sha": bench["roofline"].get("kernel_source_sha"),
       "note": "separate passes per counter group; FETCH_SIZE/WRITE_SIZE are KB per dispatch, averaged over "
               "dispatches; gfx950 correction: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section; confirmed "
               "here: mix_kernel reads voices x frames f32 and FETCH_SIZE reports half of it), WRITE_SIZE "
               "exact (fast_kernel writes the same bytes of voice rows)", "kernels": {}}
import os
COSTS = json.load(open("profiles/r05_valu_costs.json"))
CLASSES = ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "INT32", "INT64")


# what SQ_INSTS_VALU counts beyond the arithmetic classes (moves, DPP moves, lane reads / writes, compares,
# selects) used to be priced as ONE bucket at 3.72 cycles; the ISA census (tools/isa_census.py -> profiles/census/) names its
# parts per kernel, and the probe (profiles/r05_valu_probe.json) has a price for each: v_mov_b32 2.3, v_mov_b32_dpp 4.4,
# v_readlane / v_writelane 4.25, v_cmp 4.25, v_cndmask on a scalar mask 4.25.
OTHER_PRICES = {"move (v_mov_b32/b64)": 2.3, "DPP move (v_mov_b32_dpp)": 4.4, "lane read/write (v_readlane, v_writelane)": 4.25,
                "compare": 4.25, "select (v_cndmask)": 4.25, "other vector": 4.25}
CENSUS_OF = {"fast_kernel<12, 0, false, true, false, true>": "profiles/census/r06_config3_census.json",
             "fast_kernel<12, 0, false, true, false, false>": "profiles/census/r06_config3_census.json",  # (the launch of the edge groups: the same code with the masks)
             "fast_kernel<8, 2, false, true, false, false>": "profiles/census/r06_fmbank_census.json",
             "fast_kernel<8, 2, false, false, false, false>": "profiles/census/r06_lookback_census.json",
             "duo_kernel": "profiles/census/r06_lookback_census.json"}  # (its look-back waves' code; the closed-form waves' is fast_kernel<8, 0>'s)


def other_price(kernel):
    """cycles per instruction of the kernel's non-arithmetic vector instructions, from its census (None: no census)"""
    for k, f in CENSUS_OF.items():
        if k in kernel and os.path.exists(f):
            kinds = json.load(open(f))["valu_by_kind_per_operator_sample"]
            n = sum(x for q, x in kinds.items() if q in OTHER_PRICES)
            if n > 0:
                return sum(x * OTHER_PRICES[q] for q, x in kinds.items() if q in OTHER_PRICES) / n, f
    return None, None


def valu_side(v, kernel=""):
    """the VALU side of the roofline from a kernel's counters. With the class census (SQ_INSTS_VALU_*:
    tools/pmc_classes.sh / the `cls` passes of tools/profile_round.sh) every class is priced with the issue cost
    tools/valu_probe.hip measured (profiles/r05_valu_costs.json), what SQ_INSTS_VALU counts beyond the classes -- moves
    incl. DPP, compares, selects, lane reads -- as OTHER; without it, every instruction at the probe's in-mix average.
    Against 1024 SIMDs x launch cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs: MI355X_MICROARCH.md)."""
    if not ("SQ_INSTS_VALU" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0):
        return None
    cycles = v["GRBM_GUI_ACTIVE"] / 8
    n = v["SQ_INSTS_VALU"]
    out = {"insts_per_launch": n, "launch_cycles": cycles, "simds": 1024, "simd_cycles_per_launch": 1024 * cycles,
           "cycles_per_inst_per_simd": 1024 * cycles / n if n else None,
           "salu_per_valu": (v.get("SQ_INSTS_SALU", 0) / n) if n else None, "costs": "profiles/r05_valu_costs.json"}
    if all(("SQ_INSTS_VALU_" + c) in v for c in CLASSES):
        census = {c: v["SQ_INSTS_VALU_" + c] for c in CLASSES}
        census["OTHER"] = max(0.0, n - sum(census.values()))
        out["census_frac"] = {c: round(x / n, 4) for c, x in census.items()} if n else None
        op, src = other_price(kernel)
        if op is not None:
            out["other_price_cycles"] = round(op, 3)
            out["other_price_from"] = src
        for name in ("pure", "in_mix"):
            w = sum(x * (op if (c == "OTHER" and op is not None) else COSTS[name][c]) for c, x in census.items())
            out["weighted_cycles_" + name] = w
            out["frac_" + name] = w / (1024 * cycles)
        # the figure on the bench line: priced with every class's best case, i.e. the least the kernel can be using
        out["weighted_cycles_per_launch"] = out["weighted_cycles_pure"]
        out["frac"] = out["frac_pure"]
        out["formula"] = ("sum over SQ_INSTS_VALU_* classes (+ the rest of SQ_INSTS_VALU as OTHER) of instructions x the class's cost "
                          "as a stream of its own (a lower bound on what the kernel uses; frac_in_mix: 2-cycle instructions priced as "
                          "they cost between 4-cycle ones) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
    else:
        out["weighted_cycles_per_launch"] = n * COSTS["flat_in_mix"]
        out["frac"] = n * COSTS["flat_in_mix"] / (1024 * cycles)
        out["formula"] = "SQ_INSTS_VALU x the probe's hot-path mix average (no class census for this kernel) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"
    if "SQ_ACTIVE_INST_VALU" in v:  # (quad-cycles summed over waves: the share of SIMD time with a vector instruction in execution)
        out["sq_active_inst_valu_frac"] = 4 * v["SQ_ACTIVE_INST_VALU"] / (1024 * cycles)
    return out


for d, f in (("fetch", "f"), ("write", "w"), ("sq", "s"), ("inst", "i"), ("grbm", "g"), ("cls_a", "a"), ("cls_b", "b")):
    src = f"gpurun_out/pmc_{tag}_{d}/{f}_counter_collection.csv"
    if not os.path.exists(src):
        continue
    shutil.copy(src, f"profiles/{name}_pmc_{d}_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        if "sauhip" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            o = out["kernels"].setdefault(k, {})
            o.update(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), scratch=int(r["Scratch_Size"]),
                     workgroup=int(r["Workgroup_Size"]), grid=int(r["Grid_Size"]))
    for k, v in agg.items():
        for c, x in v.items():
            out["kernels"][k][c] = sum(x) / len(x)
for k, v in out["kernels"].items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch_corrected"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    vs = valu_side(v, k)
    if vs:
        v["valu"] = vs
json.dump(out, open(f"profiles/{name}_pmc_summary.json", "w"), indent=1)
# the other workloads' bench lines and kernel statistics, when the round measured them
if os.path.exists(f"gpurun_out/bench_{tag}_full.json"):
    open(f"profiles/{name}_bench_full.json", "w").write(open(f"gpurun_out/bench_{tag}_full.json").read().strip().splitlines()[-1] + "\n")
for wl in ("c5", "c4"):
    b = f"gpurun_out/bench_{tag}_{wl}.json"
    if os.path.exists(b):
        line = open(b).read().strip().splitlines()[-1]
        open(f"profiles/{name}_{wl}_bench.json", "w").write(line + "\n")
    st = f"gpurun_out/prof_{tag}_{wl}/{tag}_{wl}_kernel_stats.csv"
    if os.path.exists(st):
        shutil.copy(st, f"profiles/{name}_{wl}_kernel_stats.csv")
# configs 5 and 4: per-kernel HBM bytes and instruction counts (averages per launch), and the bytes of one step --
# the profiled command renders twice (the first step, whose SHA-256 is checked, and one timed step)
for wl, wname in (("c5", "config5"), ("c4", "config4"), ("fmbank", "fm"), ("config2", "config2")):
    ks = {}
    for d, f in ((wl + "_fetch", "m"), (wl + "_write", "m"), (wl + "_inst", "i"), (wl + "_cls_a", "a"), (wl + "_cls_b", "b"), (wl + "_grbm", "g")):
        src = f"gpurun_out/pmc_{tag}_{d}/{f}_counter_collection.csv"
        if not os.path.exists(src):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(src)):
            if "sauhip" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, x in v.items():
                ks.setdefault(k, {})[c] = sum(x) / len(x)
                ks[k]["launches"] = len(x)
    if not ks:
        continue
    total = 0.0
    for k, v in ks.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch_corrected"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            total += v["hbm_bytes_per_launch_corrected"] * v["launches"]
        vs = valu_side(v, k)
        if vs:
            v["valu"] = vs
    json.dump({"command": f"rocprofv3 --pmc <C> --kernel-trace -- python3 bench.py --workload {wname} --steps 1 --warmup 0 --no-cpu",
               "workload": {"name": wname, "renders_profiled": 2},
               "kernel_source_sha": bench["roofline"].get("kernel_source_sha"),
               "note": "averages per launch over the run; zero-work launches of a pass nobody needs are included in that "
                       "pass's average; hbm_bytes_per_step_corrected = sum over the kernels of bytes per launch x launches, "
                       "over the two renders of the command",
               "hbm_bytes_per_step_corrected": total / 2 if total else None,
               "kernels": ks}, open(f"profiles/{name}_{wl}_pmc_summary.json", "w"), indent=1)
# running-sum workloads: the sweep's lines, kernel statistics, instruction counts per launch of the single-pass build
if os.path.exists(f"gpurun_out/sweep_{tag}_fm.txt"):
    shutil.copy(f"gpurun_out/sweep_{tag}_fm.txt", f"profiles/{name}_fm_sweep.txt")
if os.path.exists(f"gpurun_out/prof_{tag}_fm/{tag}_fm_kernel_stats.csv"):
    shutil.copy(f"gpurun_out/prof_{tag}_fm/{tag}_fm_kernel_stats.csv", f"profiles/{name}_fm_kernel_stats.csv")
src = f"gpurun_out/pmc_{tag}_fm_inst/i_counter_collection.csv"
if os.path.exists(src):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        if "sauhip" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    fm = {k: dict({c: sum(x) / len(x) for c, x in v.items()}, launches=len(next(iter(v.values())))) for k, v in agg.items()}
    json.dump({"command": "rocprofv3 --pmc <C> --kernel-trace -- python3 tests/tools/gpu_sweep.py c3f",
               "note": "1024 voices x 4 operators with carrier FM, then 1024 x 2 with a carrier glide; 44100-frame steps; "
                       "averages per launch over both", "kernels": fm},
              open(f"profiles/{name}_fm_pmc_summary.json", "w"), indent=1)
fk = max((v for n, v in out["kernels"].items() if "fast_kernel" in n), key=lambda v: v.get("GRBM_GUI_ACTIVE", 0))  # (the dominant launch)
rows = bench["config"]["operators"] * frames_per_step / 60  # 64-lane rows incl. lead-in, per launch
print({a: (round(b / rows, 2) if isinstance(b, float) and b > 1e6 else b) for a, b in fk.items()})
