"""Summarise a rocprofv3 --pmc run of tools/bench_gemm.py: matrix-pipe utilisation and wave-state split per GEMM launch.

    python tools/pmc_gemm.py results.db [kernel-name substring, default gemm_f32_kernel]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
PAT = sys.argv[2] if len(sys.argv) > 2 else "gemm_f32_kernel"
cur = db.cursor()
d = collections.defaultdict(dict)
q = ("select dispatch_id, substr(kernel_name,1,110), grid_size, counter_name, sum(value), max(duration) from counters_collection "
     f"where kernel_name like '%{PAT}%' group by dispatch_id, counter_name")
for did, name, gs, cn, val, dur in cur.execute(q):
    d[did]["name"] = name[name.find(PAT) + (15 if PAT == "gemm_f32_kernel" else 0):][:36]
    d[did]["grid"] = gs
    d[did][cn] = val
    d[did]["dur"] = dur
seen = set()
for did, v in sorted(d.items()):
    key = (v["name"], v["grid"]) if PAT == "gemm_f32_kernel" else (v["name"], v["grid"], round(v["dur"] / 20000.0))
    if key in seen:
        continue
    seen.add(key)
    gui = v.get("GRBM_GUI_ACTIVE", 0) / 8.0          # summed over the 8 XCDs
    simd_cycles = gui * 1024
    wc = v.get("SQ_WAVE_CYCLES", 0) * 4.0            # quad-cycles
    out = [f"{v['name']:38s} blocks {v['grid'] // 256:6d} {v['dur'] / 1000:8.0f} us  clk {gui / v['dur']:.2f} GHz"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and simd_cycles:
        out.append(f"mfma {v['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.3f}")
    if wc and simd_cycles:
        out.append(f"waves/simd {wc / simd_cycles:.2f}")
        for k, lab in (("SQ_WAIT_ANY", "wait"), ("SQ_WAIT_INST_ANY", "issue-stall"), ("SQ_ACTIVE_INST_ANY", "active"),
                       ("SQ_WAIT_INST_LDS", "lds-stall"), ("SQ_ACTIVE_INST_LDS", "lds"), ("SQ_ACTIVE_INST_VALU", "valu"),
                       ("SQ_ACTIVE_INST_VMEM", "vmem"), ("SQ_ACTIVE_INST_SCA", "salu")):
            if k in v:
                out.append(f"{lab} {v[k] * 4.0 / wc:.2f}")
    for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
        if k in v:
            out.append(f"{k[3:]} {v[k]:.3g}")
    print("  ".join(out))
