#!/usr/bin/env python3
"""Merge the summary of one tools/profile_bench.sh run into profiles/pmc_traffic.json.

usage: pmc_to_json.py gpurun_out/prof_TAG NX NY ROUND_TAG
Every lbm::k_fused<T, V, K, PUSH, LID, AOS, ARITH[, NARROW]> kernel of the summary becomes one entry keyed the way
bench.py looks profiles up: scene-scheme-precision-NXxNY[-aos][-rolled|-contracted][-fusedK]."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    prof, nx, ny, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    summary = json.load(open(os.path.join(prof, "summary.json")))
    command = open(os.path.join(prof, "command.txt")).read().strip()
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = json.load(open(path))
    for kernel, c in summary.items():
        m = re.search(r"k_fused<(float|double), (\d+), (\d+), (true|false), (true|false), (true|false), (\d+)(?:, (true|false))?>", kernel)
        if not m or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        t, v, k, push, lid, aos, arith, narrow = m.groups()
        if narrow == "true":
            continue   # halos of whole lanes: what small lattices run, not what bench.py looks up
        key = f"{'lid' if lid == 'true' else 'shear'}-{'push' if push == 'true' else 'pull'}-{'fp32' if t == 'float' else 'fp64'}-{nx}x{ny}"
        key += ("-aos" if aos == "true" else "") + {"0": "", "1": "-rolled", "2": "-contracted"}[arith]
        if int(k) != 1:
            key += f"-fused{k}"
        bpn = 72 if t == "float" else 144
        fetch_kb, write_kb = c["FETCH_SIZE"]["mean"], c["WRITE_SIZE"]["mean"]
        entry = {
            "source": f"rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*, GRBM_GUI_ACTIVE: one pass each) and --kernel-trace --stats of "
                      f"`{command}`, kernel {kernel}, mean over {c['FETCH_SIZE']['launches']} launch(es), {tag}; raw files under profiles/{tag}/",
            "FETCH_SIZE_KB": fetch_kb,
            "WRITE_SIZE_KB": write_kb,
            "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of coalesced streaming reads -> doubled; WRITE_SIZE exact",
            "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
            "algorithmic_bytes_per_launch": nx * ny * bpn * int(k),
            "min_bytes_per_launch": nx * ny * bpn,
        }
        if "SQ_INSTS_VALU" in c:
            entry["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]["mean"]
            entry["valu_active_cycles_x4_per_launch"] = c.get("SQ_ACTIVE_INST_VALU", {}).get("mean")
        if "GRBM_GUI_ACTIVE" in c and "kernel_trace" in c:
            # one count per XCD, summed over the 8 XCDs
            entry["gfx_clock_ghz_during_launch"] = c["GRBM_GUI_ACTIVE"]["mean"] / 8 / c["kernel_trace"]["average_ns"]
        if "kernel_trace" in c:
            entry["kernel_trace_average_ms"] = c["kernel_trace"]["average_ns"] / 1e6
        table[key] = entry
        print(key, {k_: (round(v_, 4) if isinstance(v_, float) else v_) for k_, v_ in entry.items() if k_ not in ("source", "correction")})
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
