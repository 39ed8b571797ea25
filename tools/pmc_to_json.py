#!/usr/bin/env python3
"""Folds one tools/prof.sh output directory into profiles/pmc_per_launch.json.

    python3 tools/pmc_to_json.py gpurun_out/<dir> <config> [label]

Reads the per-kernel counter CSVs of the PMC passes (the plain, non-counting k_render dispatch of each pass), the
kernel-trace statistics and the JSON line bench.py printed in the kernel-trace pass (frame size, samples, rays of one
launch), and writes the per-launch record bench.py's `roofline` block scales per ray.  FETCH_SIZE / WRITE_SIZE come in
KB; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide reads
as 64 bytes, so the factor 2 is an upper estimate of the read side.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def plain_render(name: str) -> bool:
    """k_render<MODE, COUNT, PIPELINED> with COUNT == false."""
    if "k_render" not in name:
        return False
    args = name.split("k_render")[1].replace(" ", "")
    parts = args.strip("<").split(">")[0].split(",")
    return len(parts) >= 2 and parts[1] in ("false", "0")


def main():
    out, config = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(out.rstrip("/"))
    counters = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per_dispatch = {}
            for r in csv.DictReader(open(f)):
                if plain_render(r["Kernel_Name"]):
                    per_dispatch.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
            if per_dispatch:
                first = per_dispatch[sorted(per_dispatch, key=int)[0]]  # the timed launch (steps 1, warmup 0)
                for k, v in first.items():
                    counters.setdefault(k, v)
    bench = None
    for line in open(os.path.join(out, "kt.log"), errors="ignore"):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            bench = json.loads(line)
    if bench is None:
        raise SystemExit("no bench JSON line in kt.log")
    avg_ns = None
    for f in glob.glob(os.path.join(out, "kt", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if plain_render(r["Name"]):
                avg_ns = float(r["AverageNs"])
    cfg = bench["config"]
    row = {
        "source": f"{label}: rocprofv3 --pmc, one pass per counter group (tools/prof.sh)",
        "width": cfg["width"], "height": cfg["height"], "spp_per_gpu": cfg["spp_per_gpu"],
        "rays_per_launch": bench["roofline"]["rays_per_launch"],
        "kernel_ms_rocprof_avg": avg_ns / 1e6 if avg_ns else None,
        "kernel_ms_hip_events": bench["roofline"]["kernel_ms"],
        "simds": 1024,
    }
    rename = {"SQ_INSTS_VALU": "sq_insts_valu", "SQ_INSTS_SALU": "sq_insts_salu", "SQ_INSTS_LDS": "sq_insts_lds",
              "SQ_INSTS_VMEM_RD": "sq_insts_vmem_rd", "SQ_ACTIVE_INST_VALU": "sq_active_inst_valu",
              "SQ_THREAD_CYCLES_VALU": "sq_thread_cycles_valu", "SQ_WAVE_CYCLES": "sq_wave_cycles", "SQ_BUSY_CYCLES": "sq_busy_cycles",
              "SQ_WAVES": "sq_waves", "GRBM_GUI_ACTIVE": "grbm_gui_active",
              "TCC_HIT_sum": "tcc_hit", "TCC_MISS_sum": "tcc_miss", "TCP_TOTAL_CACHE_ACCESSES_sum": "tcp_total_cache_accesses",
              "TCP_TCC_READ_REQ_sum": "tcp_tcc_read_req", "TCC_EA0_RDREQ_sum": "tcc_ea0_rdreq", "TCC_EA0_WRREQ_sum": "tcc_ea0_wrreq",
              "FETCH_SIZE": "fetch_size_kb", "WRITE_SIZE": "write_size_kb",
              "TA_BUSY_avr": "ta_busy_avr", "TA_TA_BUSY_sum": "ta_busy_sum", "TD_TD_BUSY_sum": "td_busy_sum", "TD_TC_STALL_sum": "td_tc_stall_sum"}
    for k, v in counters.items():
        if k in rename:
            row[rename[k]] = v
    if "fetch_size_kb" in row and "write_size_kb" in row:
        row["hbm_bytes"] = (2.0 * row["fetch_size_kb"] + row["write_size_kb"]) * 1024.0
    if row.get("sq_waves"):
        row["waves_per_simd"] = row["sq_waves"] / 1024.0
    if row.get("sq_active_inst_valu"):
        row["lane_util_hw"] = row["sq_thread_cycles_valu"] / (64.0 * row["sq_active_inst_valu"])
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: the kernel's cycles are an eighth of it
    if row.get("grbm_gui_active") and row.get("ta_busy_avr"):
        row["kernel_cycles"] = row["grbm_gui_active"] / 8.0
        row["ta_busy_frac"] = row["ta_busy_avr"] / row["kernel_cycles"]
    path = os.path.join(ROOT, "profiles", "pmc_per_launch.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        doc = {}
    doc["_comment"] = ("Per-launch PMC figures of k_render from rocprofv3 (tools/prof.sh: one --pmc run per counter group), folded by "
                       "tools/pmc_to_json.py.  FETCH_SIZE / WRITE_SIZE in KB; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 counts "
                       "128-byte read requests as 64 bytes: the factor 2 is an upper estimate).  lane_util_hw = SQ_THREAD_CYCLES_VALU / "
                       "(64 * SQ_ACTIVE_INST_VALU).  ta_busy_frac = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8): the share of the kernel's cycles the address "
                       "unit of a CU's vector L1 is busy, averaged over the CUs.  bench.py uses these figures per ray (divided by rays_per_launch).")
    doc[str(config)] = row
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(row, indent=1))


if __name__ == "__main__":
    main()
