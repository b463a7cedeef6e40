#!/usr/bin/env python3
"""Turns gpurun_out/prof (written by tools/collect_profiles.sh on the GPU box) into the committed evidence:

    profiles/rNN_bench_f32_kernel_stats.md   rocprofv3 --kernel-trace --stats summary of the default bench.py run
    profiles/rNN_pmc_search_round.json       HBM bytes per cz_search_round launch (FETCH_SIZE / WRITE_SIZE passes)
    profiles/rNN_pmc_nn.json                 MFMA utilisation + HBM bytes of the hand-written network kernels

    python tools/summarize_profiles.py [--round 1] [--src gpurun_out/prof]
Counter units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
reported in KiB; FETCH_SIZE counts a wide coalesced read at half its bytes (doubled here); separate --pmc passes.
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEARCH = ("k_noise", "k_sim", "k_advance")
ROUND_LAUNCHES = 4            # k_sim(BACKUP), k_advance, k_noise, k_sim(SELECT)   (round 2: five, k_noise twice)
ROUND_NAMES = ["k_sim", "k_advance", "k_noise", "k_sim"]
NN = ("k_tower_pairs", "k_tower_plain2", "k_tower_plain", "k_tower", "k_resblock_ip4_c8", "k_resblock_ip_c8", "k_resblock_ip", "k_resblock_c8", "k_resblock_pipe", "k_resblock", "k_input_conv", "k_conv3x3",
      "k_split_bias_act", "k_bias_act", "k_fc_tile", "k_policy_normalize", "k_head_convs")
TOWER = ("k_tower_pairs", "k_tower", "k_resblock_ip4_c8", "k_resblock_c8", "k_resblock_pipe", "k_resblock")       # the residual tower's launches


def _targs(name, base):
    """the template arguments of kernel `base` in a kernel name: 'k_tower<true, 1>(...)' -> ['true', '1']; rocprofv3 leaves names
    with _Float16 parameters MANGLED (its demangler does not know DF16_): '7k_towerILb1ELi1EEEv...' -> ['true', '1'] too."""
    tag = f"{len(base)}{base}I"
    i = name.find(tag)
    if name.startswith("_Z") and i >= 0:
        j, out = i + len(tag), []
        while j < len(name) and name[j] != "E":
            if name.startswith("Lb", j):
                out.append("true" if name[j + 2] == "1" else "false"); j += 4
            elif name.startswith("Li", j):
                k = name.index("E", j)
                out.append(name[j + 2:k].replace("n", "-")); j = k + 1
            elif name.startswith("DF16_", j):
                out.append("_Float16"); j += 5
            elif name.startswith("DF16b", j):
                out.append("__bf16"); j += 5
            else:
                return out                                   # (something this mini-parser does not know: keep what we have)
        return out
    i = name.find(base + "<")
    if i < 0:
        return []
    j, depth, out, cur = i + len(base) + 1, 1, [], ""
    while j < len(name) and depth:
        ch = name[j]
        depth += ch == "<"
        depth -= ch == ">"
        if depth == 1 and ch == ",":
            out.append(cur.strip()); cur = ""
        elif depth:
            cur += ch
        j += 1
    return out + [cur.strip()]


def short(name):
    """Key of a kernel in the summaries.  Network kernels are keyed on their TEMPLATE ARGUMENTS (the FIRST and
    HEADS variants of one kernel were averaged under one name): k_tower<HEADS, c6>, k_resblock_c8<FIRST, C6>, ..."""
    for k in SEARCH + ("k_rules_tpb", "k_movegen_fix", "k_start_selfplay"):
        if k in name:
            return k
    for k in NN:
        if k + "<" in name or k + "(" in name or f"{len(k)}{k}I" in name or f"{len(k)}{k}E" in name:
            a = _targs(name, k)
            t = lambda v: v in ("true", "1")
            if k == "k_tower" and len(a) >= 2:
                return f"k_tower<{'HEADS' if t(a[0]) else 'image'}, {'c6' if a[1] == '1' else 'c8'}{', FIRST' if len(a) > 2 and t(a[2]) else ''}>"
            if k == "k_resblock_ip4_c8" and len(a) == 3:     # <C, XF, YF>: the chain on four matrix waves (exits are run-time arguments)
                return f"k_resblock_ip4_c8<{a[0]}, {'c6' if a[1] == '1' else ('c8' if a[2] == '0' else 'c8 -> c6')}>"
            if k == "k_tower_pairs" and len(a) == 2:
                return f"k_tower_pairs<{'bf16' if 'bf16' in a[0] or 'DF16b' in a[0] else 'f16'}, {'HEADS' if t(a[1]) else 'pairs'}>"
            if k == "k_resblock_c8" and len(a) >= 2:
                tags = [n for n, v in zip(("FIRST", "HEADS", "C6"), a + ["false"] * 3) if t(v)]
                return "k_resblock_c8<" + ", ".join(tags or ["inner"]) + ">"
            if k == "k_resblock_pipe" and a:
                return "k_resblock_pipe<" + ("bf16" if "bf16" in a[0] else "f16") + (", FIRST" if len(a) > 1 and t(a[1]) else "") + ">"
            if k == "k_fc_tile" and a:
                return f"k_fc_tile<{a[0]}>"
            return k if not a else k + "<" + ", ".join(a)[:40] + ">"
    return name[:70]


def read_counters(src, sub):
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def per_kernel(rows, skip_first=2):
    """mean per launch of each counter for each kernel, excluding the first `skip_first` launches of that kernel"""
    seen = collections.Counter()
    first = {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for did, k, c, v in rows:
        key = (k, c)
        seen[key] += 1
        if seen[key] > skip_first:
            acc[k][c].append(v)
    # full-size launches only: since round 4 the load-time guard and numerics_check launch the network kernels on a few
    # hundred positions as well; a counter value below half of the kernel's largest marks such a launch
    def full(v):
        top = max(v)
        return [x for x in v if x > 0.5 * top] if top > 0 else v
    return {k: {c: sum(full(v)) / len(full(v)) for c, v in d.items()} | {"launches": max(len(full(v)) for v in d.values())}
            for k, d in acc.items()}


def search_rounds(rows, counter):
    """sum of `counter` over the 4 launches of each cz_search_round (k_sim, k_advance, k_noise, k_sim)"""
    vals = [v for did, k, c, v in rows if c == counter and k in SEARCH]
    return [sum(vals[i:i + ROUND_LAUNCHES]) for i in range(0, len(vals) - len(vals) % ROUND_LAUNCHES, ROUND_LAUNCHES)]


def search_launch_counters(rows, skip_rounds=3):
    """SQ counters of the launches of a cz_search_round, by position, mean over the steady-state rounds; with the
    fractions of the waves' cycles spent waiting / issuing (SQ_WAIT_ANY etc. over SQ_WAVE_CYCLES)"""
    per = collections.OrderedDict()
    for did, k, c, v in rows:
        if k in SEARCH:
            per.setdefault(did, {"name": k})[c] = v
    seq = list(per.values())
    n = ROUND_LAUNCHES
    rounds = [seq[i:i + n] for i in range(0, len(seq) - len(seq) % n, n)]
    rounds = [r for r in rounds if [x["name"] for x in r] == ROUND_NAMES][skip_rounds:]
    if not rounds:
        return None
    labels = ["k_sim(BACKUP)", "k_advance", "k_noise", "k_sim(SELECT)"]
    out = {"rounds": len(rounds)}
    for j, lab in enumerate(labels):
        acc = collections.defaultdict(float)
        for r in rounds:
            for c, v in r[j].items():
                if c != "name":
                    acc[c] += v / len(rounds)
        d = dict(acc)
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                d[c + "_frac_of_wave_cycles"] = d.get(c, 0.0) / wc
        out[lab] = d
    return out


FILTER_BYTES = {"c6": 2 * 0.52e6, "c8": 2 * 0.59e6, "pair": 2 * 0.59e6}        # one block's two packed filters


def tower_filter_model(e, boards, nb, kind, chained):
    """Adds the FILTER term to a tower launch's traffic entry: every XCD's L2 (4 MB) streams the launch's filters.  A one-block
    launch reads its two filters (1.0-1.2 MB) once per XCD.  A chain of nb blocks cycles nb x 1.0-1.2 MB through each L2 once per
    PAIR of boards of its workgroups (the workgroups of an XCD run nearly in lock-step): 8 XCDs x pairs x nb filters -- more than
    the L2 holds from three blocks on, so every cycle comes from HBM / the memory-side cache again.  (Same instruction stream with
    every block reading ONE block's filters: no faster, profiles/r06_chain_same_filters.log -- the refetches cost nothing.)"""
    fb = nb * FILTER_BYTES.get(kind, 1.1e6)
    pairs = -(-int(boards) // (256 * 2)) if chained else 1
    cycles = pairs if fb > 3.5e6 else 1
    e["filter_bytes_model"] = 8 * cycles * fb
    e["filter_model"] = f"8 XCDs x {cycles} L2 cycle(s) x {nb} block(s) x {FILTER_BYTES.get(kind, 1.1e6) / 1e6:.2f} MB"
    if e.get("measured_bytes"):
        e["modelled_bytes"] = e["algorithmic_activation_bytes"] + e["filter_bytes_model"]
        e["measured_over_modelled"] = e["measured_bytes"] / e["modelled_bytes"]
    return e


def patch_json(path):
    """Recompute the filter-model fields of an existing rNN_pmc_nn.json (the raw counter tables do not travel back from the box)."""
    d = json.load(open(path))
    tt = d.get("tower_traffic") or {}
    boards = (tt.get("per_forward") or {}).get("boards_per_launch")
    tot = 0.0
    for key, e in tt.items():
        if key == "per_forward" or not boards:
            continue
        step, kern = key.split(":", 1)
        kind = "c6" if "c6" in kern.lower() else ("c8" if "c8" in kern else "pair")
        tower_filter_model(e, boards, e.get("blocks", 1), kind, step in ("tower", "pairs", "tower_first"))
        tot += e.get("modelled_bytes", 0.0)
    if tot and tt.get("per_forward", {}).get("measured_bytes"):
        tt["per_forward"]["modelled_bytes"] = tot
        tt["per_forward"]["measured_over_modelled"] = tt["per_forward"]["measured_bytes"] / tot
    json.dump(d, open(path, "w"), indent=1)
    print("patched", path, {k: round(v.get("measured_over_modelled", 0), 3) for k, v in tt.items() if isinstance(v, dict)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=1)
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "prof"))
    ap.add_argument("--patch-json", default=None, help="only recompute the filter-model fields of an existing rNN_pmc_nn.json")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"),
                    help="where to write (on the GPU box: a directory under gpurun_out/, so that the raw counter tables need not travel)")
    a = ap.parse_args()
    if a.patch_json:
        return patch_json(a.patch_json)
    tag = f"r{a.round:02d}"
    prof = a.out
    os.makedirs(prof, exist_ok=True)

    # ---- 1. kernel stats ----
    sf = glob.glob(os.path.join(a.src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if sf:
        rows = list(csv.DictReader(open(sf[0])))
        bench = {}
        try:
            bench = json.loads(open(os.path.join(a.src, "stats.json")).read().strip().splitlines()[-1])
        except Exception:
            pass
        lines = [f"# rocprofv3 --kernel-trace --stats of `bench.py` (normal config, default tower arithmetic), round {a.round}",
                 "",
                 "Command (GPU box): `cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv "
                 "-d gpurun_out/prof/stats -o s -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-micro "
                 "--sustained-rounds 0 --no-other-configs --no-dist`",
                 ""]
        if bench:
            lines += [f"bench.py under the profiler: {bench.get('value', 0):.0f} expansions/s, "
                      f"{bench.get('ms_per_step', 0):.2f} ms per round "
                      f"(search-round kernels {(bench.get('roofline_search') or {}).get('avg_launch_ms', 0):.3f} ms).", ""]
        lines += ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
        for r in rows[:28]:
            lines.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                         f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                         f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
        tf = glob.glob(os.path.join(a.src, "stats", "**", "*kernel_trace.csv"), recursive=True)
        if tf:
            tr = [r for r in csv.DictReader(open(tf[0])) if short(r["Kernel_Name"]) in SEARCH]
            first_sim = next((i for i, r in enumerate(tr) if short(r["Kernel_Name"]) == "k_sim"), 0)
            tr = tr[first_sim:]                                   # (align on a round's first launch)
            seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                    r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("Scratch_Size", ""),
                    r.get("LDS_Block_Size", "")) for r in tr]
            L = ROUND_LAUNCHES
            nr = len(seq) // L
            if nr >= 4:
                lines += ["", f"## One lock-step round = {L} launches on one stream (mean of the last {nr - 3} rounds, us)", "",
                          "| launch | mean us | VGPR | scratch B | LDS B |", "|---|---|---|---|---|"]
                names = ["k_sim(BACKUP)", "k_advance", "k_noise", "k_sim(SELECT)"]
                tot = 0.0
                for j in range(L):
                    v = [seq[i * L + j][1] for i in range(3, nr)]
                    m = sum(v) / len(v)
                    tot += m
                    lines.append(f"| {names[j]} | {m:.1f} | {seq[3 * L + j][2]} | {seq[3 * L + j][3]} | {seq[3 * L + j][4]} |")
                lines.append(f"| **sum** | **{tot:.1f}** | | | |")
        with open(os.path.join(prof, f"{tag}_bench_f32_kernel_stats.md"), "w") as f:
            f.write("\n".join(lines) + "\n")
        print("wrote", f"{tag}_bench_f32_kernel_stats.md")

    # ---- 2. HBM traffic of the search round ----
    fetch = read_counters(a.src, "pmc_fetch")
    write = read_counters(a.src, "pmc_write")
    if fetch and write:
        fr, wr = search_rounds(fetch, "FETCH_SIZE"), search_rounds(write, "WRITE_SIZE")
        n = min(len(fr), len(wr))
        fm = sum(fr[3:n]) / max(1, n - 3)
        wm = sum(wr[3:n]) / max(1, n - 3)
        out = {"kernels": "k_sim(BACKUP) + k_advance + k_noise + k_sim(SELECT) = one cz_search_round",
               "workload": "bench.py normal config (4096 games, K=8, 7x128 split-bf16 network, u8 planes queue), "
                           f"steady-state rounds (first 3 of {n} excluded)",
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (tools/collect_profiles.sh)",
               "unit_note": "FETCH_SIZE / WRITE_SIZE are in KiB. Per MI355X_MICROARCH.md the gfx950 FETCH_SIZE reports a wide "
                            "coalesced read at half its bytes (these reads are narrow gathers: uncalibrated); "
                            "traffic_bytes uses 2*FETCH + WRITE as the conservative figure",
               "FETCH_SIZE_KB_per_round": fr[:n], "FETCH_SIZE_KB_steady_mean": fm,
               "WRITE_SIZE_KB_per_round": wr[:n], "WRITE_SIZE_KB_steady_mean": wm,
               "traffic_bytes_per_launch": (2 * fm + wm) * 1024.0, "traffic_bytes_per_launch_raw": (fm + wm) * 1024.0}
        sqs = search_launch_counters(read_counters(a.src, "pmc_sq"))
        if sqs:
            out["sq_counters_per_launch"] = sqs
            out["sq_note"] = ("SQ pass of tools/collect_profiles.sh (opening-phase rounds of the bench); SQ_WAIT_ANY = wave cycles "
                              "waiting on anything (memory, LDS, dependencies), SQ_WAIT_INST_ANY = waiting to issue, "
                              "SQ_ACTIVE_INST_ANY = executing; four waves share a SIMD in k_sim")
        with open(os.path.join(prof, f"{tag}_pmc_search_round.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", f"{tag}_pmc_search_round.json", out["traffic_bytes_per_launch"] / 1e6, "MB per round")

    # ---- 3. network kernels: MFMA utilisation and HBM bytes ----
    sq = per_kernel(read_counters(a.src, "pmc_sq"))
    grbm = per_kernel(read_counters(a.src, "pmc_grbm"))
    fk, wk = per_kernel(fetch), per_kernel(write)
    nn = {}
    for k in sorted(sq):
        if not any(k == b or k.startswith(b + "<") for b in NN):
            continue
        d = dict(sq[k])
        gui = grbm.get(k, {}).get("GRBM_GUI_ACTIVE")
        d["GRBM_GUI_ACTIVE"] = gui
        if gui:
            # MfmaUtil as rocprof defines it: matrix-pipe busy cycles / (kernel cycles x SIMDs on the chip).
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (62.2 M per 4.0 ms launch = 8 x 1.94 GHz x 4.0 ms).
            d["kernel_cycles"] = gui / 8.0
            d["mfma_util"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 256 * 4)
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                d[c + "_frac_of_wave_cycles"] = d.get(c, 0.0) / wc
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        if k in fk and k in wk:
            d["FETCH_SIZE_KB"] = fk[k]["FETCH_SIZE"]
            d["WRITE_SIZE_KB"] = wk[k]["WRITE_SIZE"]
            # FETCH_SIZE reports half of a wide read, WRITE_SIZE is exact for dense stores and counts 32-byte granules for the
            # 24-byte bf6 pieces (profiles/r06_hbm_counter_calibration.json)
            d["hbm_read_bytes"] = 2 * fk[k]["FETCH_SIZE"] * 1024.0
            d["hbm_write_bytes"] = wk[k]["WRITE_SIZE"] * 1024.0
            d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
        nn[k] = d
    # what a forward's tower launches are expected to move (an operand pair is 46 080 B per board) next to what the
    # counters saw; the launch plan and the boards per launch come from the bench line of the stats run
    bench = {}
    try:
        bench = json.loads(open(os.path.join(a.src, "stats.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    boards = (bench.get("roofline") or {}).get("boards_per_launch")
    plan = (bench.get("roofline") or {}).get("launch_plan")
    tower = {}
    if boards and plan:
        PAIR, HEADF, MASK = 46080.0, 6 * 90 * 4.0, 384.0
        tot_alg = tot_meas = tot_model = 0.0
        for st in plan:
            key, rd, wr, nb = st.get("kernel"), 0.0, 0.0, st.get("blocks", 1)
            rd = MASK if st["step"] in ("first", "tower_first") else PAIR
            wr = HEADF if st.get("exit") == "heads" else PAIR
            alg = boards * (rd + wr)
            e = {"blocks": nb, "algorithmic_activation_bytes": alg}
            m = nn.get(key)
            if m and m.get("hbm_bytes_per_launch"):
                e["measured_bytes"] = m["hbm_bytes_per_launch"]
                e["measured_read_bytes"], e["measured_write_bytes"] = m["hbm_read_bytes"], m["hbm_write_bytes"]
                # the rest: every workgroup streams its blocks' filters through its XCD's L2 (8 L2s: at least 8 first reads of
                # each filter; refetches when a chain's filters exceed 4 MB) + the input-layer table
                e["excess_over_activations"] = m["hbm_bytes_per_launch"] - alg
                e["write_ratio"] = m["hbm_write_bytes"] / (boards * wr)
                e["read_ratio"] = m["hbm_read_bytes"] / (boards * rd)
                tot_meas += m["hbm_bytes_per_launch"]
            tower_filter_model(e, boards, nb, st.get("kind"), st["step"] in ("tower", "pairs", "tower_first"))
            tot_model += e.get("modelled_bytes", 0.0)
            tot_alg += alg
            tower[f"{st['step']}:{key}"] = e
        tower["per_forward"] = {"algorithmic_activation_bytes": tot_alg, "measured_bytes": tot_meas or None,
                                "modelled_bytes": tot_model or None,
                                "measured_over_modelled": (tot_meas / tot_model) if tot_meas and tot_model else None,
                                "blocks": sum(st.get("blocks", 1) for st in plan), "boards_per_launch": boards,
                                "tower_arithmetic": (bench.get("roofline") or {}).get("tower_arithmetic")}
    if nn:
        out = {"workload": "bench.py normal config: the compact queue's boards per launch (tower_traffic.per_forward), 7x128 network, "
                           "default tower arithmetic; kernels keyed on their template arguments; means over the full-size launches "
                           "(the load-time guard's 256-board calibration launches are dropped)",
               "method": "rocprofv3 --pmc passes of tools/collect_profiles.sh (SQ set, GRBM_GUI_ACTIVE, FETCH_SIZE, "
                         "WRITE_SIZE: one run each); means per launch, first 2 launches of each kernel excluded",
               "mfma_util_definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)",
               "counter_calibration": "profiles/r06_hbm_counter_calibration.json (hbm_read_bytes = 2 x FETCH_SIZE, hbm_write_bytes = WRITE_SIZE)",
               "tower_traffic": tower, "kernels": nn}
        with open(os.path.join(prof, f"{tag}_pmc_nn.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", f"{tag}_pmc_nn.json", {k: round(v.get("mfma_util", 0), 3) for k, v in nn.items()})


if __name__ == "__main__":
    main()
