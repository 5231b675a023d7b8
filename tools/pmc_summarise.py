"""Summarise separate rocprofv3 --pmc passes over `python3 bench.py ...` into per-walker-step figures for the bench's
dominant kernel and per-launch HBM traffic for the target force kernel.

usage: python tools/pmc_summarise.py <config> <walkers> <steps_per_launch> <out.json> <pass_dir> [<pass_dir> ...]
                                     [--second <steps_per_launch_2> <pass_dir> ...]
Each pass_dir holds one rocprofv3 output tree (…_counter_collection.csv).  With --second (FETCH_SIZE / WRITE_SIZE passes of
the same command at another launch size) the sampler kernel's HBM bytes are fitted as
fixed_bytes_per_launch + bytes_per_walker_step x walkers x steps, which bench.py evaluates at ITS launch size.
Counters used when present:
  SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_INSTS_VALU_TRANS_F32, SQ_ACTIVE_INST_VALU (quad-cycles), SQ_VALU_MFMA_BUSY_CYCLES,
  SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE (sum over the 8 XCDs), FETCH_SIZE, WRITE_SIZE (KB).
Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE are in KB;
FETCH_SIZE counts a coalesced streaming read at half its bytes (128-B requests tallied at 64 B): x2 applied to every
kernel here -- all their reads are coalesced streams (walkers, scratch reloads; >= 256 contiguous bytes per wave
instruction).  Check: the LJ55 sampler reads its 21.6 MB of walkers once per launch and reports FETCH_SIZE 11.3 MB.
valu_issue_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): share of SIMD cycles in which a vector
(non-matrix) instruction holds the issue port."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    config, walkers, chunk, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rest = sys.argv[5:]
    second = None
    if "--second" in rest:
        i = rest.index("--second")
        second = (int(rest[i + 1]), load(rest[i + 2:]))
        rest = rest[:i]
    acc = load(rest)
    mean = lambda k, c: (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else None
    samp = [k for k in acc if "egnn_kernel" in k and "true" in k.split("(")[0]]
    # the sampler launches: the main kernel is the one with the most MFMA work
    samp.sort(key=lambda k: -(mean(k, "SQ_INSTS_MFMA") or mean(k, "WRITE_SIZE") or 0))
    res = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            res = json.load(f)
    if samp:
        k = samp[0]
        ws = walkers * chunk
        e = {"kernel": k.split("(")[0].replace("void pita::", ""), "walkers": walkers, "steps_per_launch": chunk,
             "launches_averaged": max(len(v) for v in acc[k].values()),
             "source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py (separate passes), tools/pmc_summarise.py"}
        g = mean(k, "GRBM_GUI_ACTIVE")
        av = mean(k, "SQ_ACTIVE_INST_VALU")
        if g and av:
            e["valu_issue_frac"] = 4.0 * av / (1024.0 * g / 8.0)
            e["kernel_cycles_per_xcd"] = g / 8.0
        for c, name in (("SQ_INSTS_VALU", "valu_insts_per_walker_step"), ("SQ_INSTS_MFMA", "mfma_insts_per_walker_step"),
                        ("SQ_INSTS_VALU_TRANS_F32", "trans_insts_per_walker_step"),
                        ("SQ_INSTS_LDS", "lds_insts_per_walker_step")):
            v = mean(k, c)
            if v is not None:
                e[name] = v / ws
        mb = mean(k, "SQ_VALU_MFMA_BUSY_CYCLES")
        if mb and g:
            e["mfma_pipe_busy_frac"] = mb / (1024.0 * g / 8.0)
        wc, wa_, wi = mean(k, "SQ_WAVE_CYCLES"), mean(k, "SQ_WAIT_ANY"), mean(k, "SQ_WAIT_INST_ANY")
        if wc:
            e["wave_cycles_per_walker_step"] = 4.0 * wc / ws
            if wa_ is not None:
                e["wave_wait_any_frac"] = wa_ / wc
            if wi is not None:
                e["wave_wait_inst_frac"] = wi / wc
        fe, wr = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
        if fe is not None and wr is not None:
            e["fetch_bytes_per_launch_raw"] = fe * 1024
            e["write_bytes_per_launch"] = wr * 1024
            e["fetch_size_correction"] = 2.0
            t1 = (2.0 * fe + wr) * 1024
            e["traffic_bytes_per_launch"] = t1
            if second is not None and second[1].get(k, {}).get("FETCH_SIZE") and second[1][k].get("WRITE_SIZE"):
                c2, a2 = second
                m2 = lambda c: sum(a2[k][c]) / len(a2[k][c])
                t2 = (2.0 * m2("FETCH_SIZE") + m2("WRITE_SIZE")) * 1024
                per = (t1 - t2) / (walkers * (chunk - c2))
                if per < 0.0:  # no growth with the launch length within the counters' noise: all of it is per launch
                    per = 0.0
                e["bytes_per_walker_step"] = per
                e["fixed_bytes_per_launch"] = max(t1, t2) if per == 0.0 else t1 - per * ws
                e["fit_from"] = {f"{chunk}_steps_per_launch": t1, f"{c2}_steps_per_launch": t2}
        res[f"sampler_{config}"] = e
    force = [k for k in acc if any(t in k for t in ("lj13_kernel", "pair_energy_kernel", "pair_energy_n3l_kernel", "ring_energy_kernel",
                                                    "ff_kernel"))]
    for k in force:
        fe, wr = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
        if fe is None or wr is None:
            continue
        # several batch sizes may share a kernel name: keep the launches whose write size matches this config's batch
        res.setdefault(f"force_{config}_kernels", {})[k.split("(")[0].replace("void pita::", "")] = {
            "launches": len(acc[k]["FETCH_SIZE"]), "fetch_bytes_x2_streaming_correction": fe * 2048,
            "write_bytes": wr * 1024}
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
