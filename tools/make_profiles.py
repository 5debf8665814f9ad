#!/usr/bin/env python3
"""Turn the rocprofv3 outputs merged back under gpurun_out/ into the tracked summaries under profiles/.

    gpurun_out/prof_bench/bench_results.db      rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline
    gpurun_out/pmc_FETCH_SIZE/r_results.db      rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/prof_one.py corr 2
    gpurun_out/pmc_WRITE_SIZE/r_results.db      rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python tools/prof_one.py corr 2
usage: make_profiles.py <tag>      (e.g. r01b)"""
import json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")

db = os.path.join(G, "prof_bench", "bench_results.db")
if os.path.exists(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(os.path.join(P, "%s_bench_kernel_stats.md" % tag), "w") as f:
        f.write("# %s -- `rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-side-configs --no-e2e`\n\n" % tag)
        f.write("MI355X (gfx950), ROCm 7.2.  Source: gpurun_out/prof_bench/bench_results.db (top_kernels view); durations in "
                "microseconds.\nOne hot-path pass (drop-in mode, weights packed once) = 5 correlation calls (levels 6 / 5 / 4: "
                "`corr_gramk_kernel`, the Gram band on the bf16 matrix cores with a wave per 32 channels; levels 3 and 2: `corr_gram_kernel`, the Gram band with cooperative "
                "full-line stores -- last template argument 2: level 3's two-chunk K loop (64 channels), 1: level 2; fifth template argument = TERMS: 5 the round-6 form (operand split on the matrix cores, results one step behind the chains)), 4 x (`offsets_from_flow_v4_kernel`: the reference's separate offset tensor, + deformable conv `dc_mma_kernel<MT, PT, KW, RING>`: bf16 x 3 on the matrix cores, MT filter tiles x PT pixel tiles x KW K slices per block), 1 warp.  bench.py also runs the pass on two more "
                "streams for its informational `pipelined` figure, the level-2 correlation 200 more times back to back, 200 eager "
                "passes with every kernel timed (`roofline`, `kernels`), 18 launches on rotated buffers, the rough-flow batch and the pass with re-packed weights (`repack`: + `dcm_pack_weights_kernel`); "
                "`profiles/%s_corr_l2_durations_by_context.txt` cuts the level-2 correlation's launches by context.\n" % tag +
                "Only `mfn::` kernels belong to the pass; the `at::native` rows are bench.py's checksum.\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("void ", "").replace("mfn::", "")
            if len(short) > 90:
                short = short[:87] + "..."
            f.write("| `%s` | %d | %.1f | %.3f | %.2f |\n" % (short, calls, tot, avg, pct))
    print("wrote", "%s_bench_kernel_stats.md" % tag)


def pmc(counter):
    db = os.path.join(G, "pmc_%s" % counter, "r_results.db")
    if not os.path.exists(db):
        return None
    cur = sqlite3.connect(db).cursor()
    vals, dur, kname = [], [], None
    for name, cnt, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if "corr_" in name and cnt == counter:
            vals.append(val); kname = name
    for name, d in cur.execute("select name, end-start from kernels"):
        if "corr_" in name:
            dur.append(d)
    return kname, sum(vals) / len(vals), len(vals), (sum(dur) / len(dur) if dur else None)


fe, wr = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
if fe and wr:
    traffic = fe[1] * 1024 * 2 + wr[1] * 1024
    rec = {"kernel": fe[0].split("(")[0] + " on level 2 (N=8, C=32, 96x128, md=4)",
           "FETCH_SIZE_KB": fe[1], "WRITE_SIZE_KB": wr[1],
           "fetch_correction": "x2 (gfx950 rocprofv3 FETCH_SIZE reports half of a wide coalesced stream, MI355X_MICROARCH.md section HBM)",
           "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 4 * 8 * 96 * 128 * (64 + 81),
           "launches": fe[2], "avg_kernel_ns_in_pmc_pass": fe[3],
           "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/prof_one.py corr 2",
                        "rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python tools/prof_one.py corr 2"]}
    json.dump(rec, open(os.path.join(P, "%s_corr_l2_hbm_traffic.json" % tag), "w"), indent=1)
    print("wrote", "%s_corr_l2_hbm_traffic.json" % tag, "traffic %.2f MB" % (traffic / 1e6))
for name in ("bench", "bench_fused", "bench_cfg3", "bench_cfg4", "bench_repack", "bench_streams3"):
    src = os.path.join(G, name + ".log")
    if os.path.exists(src):
        line = open(src).read().strip().splitlines()[-1]
        try:
            json.loads(line)
        except Exception:
            continue
        open(os.path.join(P, "%s_%s.json.log" % (tag, name)), "w").write(line + "\n")
        print("wrote", "%s_%s.json.log" % (tag, name))

# training-step pass (cfg5): gpurun_out/prof_cfg5/cfg5_results.db from
#   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg5 -o cfg5 -- python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-e2e
db5 = os.path.join(G, "prof_cfg5", "cfg5_results.db")
if os.path.exists(db5):
    cur = sqlite3.connect(db5).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(os.path.join(P, "%s_bench_cfg5_kernel_stats.md" % tag), "w") as f:
        f.write("# %s -- `rocprofv3 --kernel-trace --stats -- python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline --no-epe --no-e2e`\n\n" % tag)
        f.write("MI355X (gfx950), ROCm 7.2.  Source: rocprofv3 rocpd database (top_kernels view); durations in microseconds.\n"
                "Training-step pass (BASELINE configs[4], 8 pairs of 384x512 per GPU): the S forward pass, then corr_bwd -> deform_bwd per "
                "level.  `dc_bwd_input_pix_kernel` = input + offset gradient of the deformable conv in the forward's orientation "
                "(kernels/dc_backward.h); `dc_bwd_weight_pc_kernel<MTOT>` + "
                "`dc_bwd_weight_reduce_kernel` = weight + bias gradient (columns as the forward produces them, per-block slabs, "
                "fixed-order sum; its blocks also clear gx / goffset on their way in); `corr_bwd_lds_kernel` computes g1 and g2 in separate blocks (the other feature map's rows through LDS); `fill_zero4_kernel` zeroes the "
                "write-mode gradients of a call where no slab launch does it (level 5).  bench.py also runs the pass on two more streams (`pipelined`) and eager passes for "
                "`roofline` / `kernels`.\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("void ", "").replace("mfn::", "")
            if len(short) > 90:
                short = short[:87] + "..."
            f.write("| `%s` | %d | %.1f | %.3f | %.2f |\n" % (short, calls, tot, avg, pct))
    print("wrote", "%s_bench_cfg5_kernel_stats.md" % tag)
for name in ("bench_cfg5", "bench_cfg5_fused"):
    src = os.path.join(G, name + ".log")
    if os.path.exists(src):
        line = open(src).read().strip().splitlines()[-1]
        try:
            json.loads(line)
        except Exception:
            continue
        open(os.path.join(P, "%s_%s.json.log" % (tag, name)), "w").write(line + "\n")
        print("wrote", "%s_%s.json.log" % (tag, name))
for name in ("bwd_levels", "corr_bwd_levels"):
    src = os.path.join(G, name + ".txt")
    if os.path.exists(src):
        open(os.path.join(P, "%s_%s.txt" % (tag, name)), "w").write(open(src).read())
        print("wrote", "%s_%s.txt" % (tag, name))


# SQ counters of dc_mma_kernel: gpurun_out/r05p/dc_pmc.txt (tools/r05_profiles.sh: one rocprofv3 --pmc pass per counter set, 20 launches each)
src = os.path.join(G, "r05p", "dc_pmc.txt")
if os.path.exists(src) and tag == "r05":
    import re
    data, durs, kern, lvl = {}, {}, {}, None
    for line in open(src):
        m = re.match(r"level (\d) set", line)
        if m:
            lvl = int(m.group(1)); continue
        m = re.match(r"\s+(\S+)\s+mean\s+([0-9.]+)", line)
        if m:
            data.setdefault(lvl, {})[m.group(1)] = float(m.group(2)); continue
        m = re.search(r"trace: void mfn::(dc_mma_kernel<[^>]+>).*avg_ns=(\d+)", line)
        if m:
            kern[lvl] = m.group(1); durs.setdefault(lvl, []).append(int(m.group(2)))
    lv = sorted(data)
    with open(os.path.join(P, "r05_dc_pmc.md"), "w") as f:
        f.write("# r05 -- SQ counters of the shipped deformable convolution (`dc_mma_kernel`, bf16 x 3 on the matrix cores), cfg2 levels 2..5\n\n"
                "MI355X (gfx950), ROCm 7.2.  `tools/r05_profiles.sh`: per level and counter set one `rocprofv3 --pmc <set> --kernel-trace -- python "
                "tools/prof_one.py deform <level>` (20 stand-alone launches, smooth bench flow); raw output gpurun_out/r05p/dc_pmc.txt.  Counters are "
                "sums over the chip (1024 SIMDs); `avg us` is rocprofv3's kernel duration in those passes (stand-alone launches: ~3 us above the "
                "same kernel back to back inside the bench's graph, see r05_bench_kernel_stats.md and the bench line's `ops_in_graph_us`).\n\n")
        f.write("| | " + " | ".join("L%d `%s`" % (l, kern.get(l, "?").replace("dc_mma_kernel", "")) for l in lv) + " |\n|---|" + "---|" * len(lv) + "\n")
        def row(name, fn, fmt="%.1f"):
            f.write("| %s | " % name + " | ".join((fmt % fn(l)) if fn(l) is not None else "-" for l in lv) + " |\n")
        avg = {l: sum(durs[l]) / len(durs[l]) / 1000.0 for l in lv}
        g = lambda l, k: data[l].get(k)
        row("avg us (rocprofv3, stand-alone)", lambda l: avg[l], "%.2f")
        row("waves", lambda l: g(l, "SQ_WAVES"), "%d")
        row("VALU instructions / wave", lambda l: g(l, "SQ_INSTS_VALU") / g(l, "SQ_WAVES"), "%.0f")
        row("SALU instructions / wave", lambda l: g(l, "SQ_INSTS_SALU") / g(l, "SQ_WAVES"), "%.0f")
        row("LDS instructions / wave", lambda l: g(l, "SQ_INSTS_LDS") / g(l, "SQ_WAVES"), "%.0f")
        row("VMEM reads / wave (incl. LDS-DMA)", lambda l: g(l, "SQ_INSTS_VMEM_RD") / g(l, "SQ_WAVES"), "%.0f")
        row("MFMA instructions / wave", lambda l: g(l, "SQ_INSTS_MFMA") / g(l, "SQ_WAVES"), "%.0f")
        row("MFMA busy, SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES)", lambda l: g(l, "SQ_VALU_MFMA_BUSY_CYCLES"), "%.0f")
        row("MFMA busy / (1024 SIMDs x avg x 2.4 GHz)", lambda l: g(l, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * avg[l] * 2400.0), "%.3f")
        row("VALU issue, SIMD-cycles (4 x SQ_ACTIVE_INST_VALU)", lambda l: 4 * g(l, "SQ_ACTIVE_INST_VALU"), "%.0f")
        row("VALU issue / (1024 SIMDs x avg x 2.4 GHz)", lambda l: 4 * g(l, "SQ_ACTIVE_INST_VALU") / (1024 * avg[l] * 2400.0), "%.3f")
        row("wave-cycles waiting for any instruction / wave-cycles", lambda l: g(l, "SQ_WAIT_INST_ANY") / g(l, "SQ_WAVE_CYCLES"), "%.3f")
        row("wave-cycles waiting for LDS / wave-cycles", lambda l: g(l, "SQ_WAIT_INST_LDS") / g(l, "SQ_WAVE_CYCLES"), "%.3f")
        row("LDS bank-conflict cycles / LDS active cycles", lambda l: g(l, "SQ_LDS_BANK_CONFLICT") / g(l, "SQ_LDS_IDX_ACTIVE"), "%.3f")
        row("FETCH_SIZE KB (x2 on gfx950: MI355X_MICROARCH.md)", lambda l: g(l, "FETCH_SIZE"), "%.0f")
        row("WRITE_SIZE KB", lambda l: g(l, "WRITE_SIZE"), "%.0f")
        f.write("\nReading.  The matrix cores are busy 18-21 % of the launch at levels 2 / 3 and the VALU issues for 33-44 % of it: the kernel is bound by the "
                "VALU work that forms the B operand -- per K step of a wave ~42 instructions of separable interpolation and ~46 of the three-term split "
                "for 6 x MT matrix instructions --, and on this chip bf16 matrix instructions and VALU instructions of one SIMD do not overlap to any "
                "useful degree (tools/ubench/mfma_valu_bf16.hip), so the two add.  Levels 4 / 5 have one wave per SIMD (192 blocks): latency of the "
                "weight / window transfers and the K-slice reduction show as wait cycles.  The bank conflicts are the 4x4 neighbourhood reads "
                "(rows 20 or 24 floats apart); LDS waits are 6-7 % of the wave cycles.  HBM traffic is the algorithmic minimum (input once, output "
                "once): WRITE_SIZE = N x Cout x H x W x 4 exactly.\n")
    print("wrote r05_dc_pmc.md")
